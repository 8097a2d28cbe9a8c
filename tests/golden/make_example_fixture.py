"""Extracts the reference's bundled example data set (inst/extdata/example_data.rds: observed counts M and the
COSMIC signatures / exposures it was generated from) into tests/golden/reference_example_data.npz.

The .rds file is gzip-compressed R serialisation (XDR, version 3); this is a minimal reader for the object types
that occur in it (lists, integer / double / character vectors, attribute pairlists, symbols).  DATA only: no
reference code is read or executed.  Run once, in the container that has /root/reference:

    python tests/golden/make_example_fixture.py
"""
import gzip
import os
import struct
import sys

import numpy as np

SRC = "/root/reference/inst/extdata/example_data.rds"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_example_data.npz")


class Reader:
    def __init__(self, buf):
        self.b, self.i, self.refs = buf, 0, []

    def int(self):
        v = struct.unpack(">i", self.b[self.i:self.i + 4])[0]; self.i += 4; return v

    def raw(self, n):
        v = self.b[self.i:self.i + n]; self.i += n; return v

    def item(self):
        flags = self.int()
        typ, has_attr, has_tag = flags & 0xFF, bool(flags & (1 << 9)), bool(flags & (1 << 10))
        if typ == 254:                       # NILVALUE
            return None
        if typ == 255:                       # REFSXP
            return self.refs[(flags >> 8) - 1]
        if typ == 1:                         # SYMSXP
            name = self.item(); self.refs.append(name); return name
        if typ == 9:                         # CHARSXP
            n = self.int(); return None if n == -1 else self.raw(n).decode()
        if typ == 2:                         # LISTSXP (pairlist) -> list of (tag, value)
            out = []
            while True:
                attr = self.item() if has_attr else None
                tag = self.item() if has_tag else None
                out.append((tag, self.item()))
                flags = self.int()
                typ, has_attr, has_tag = flags & 0xFF, bool(flags & (1 << 9)), bool(flags & (1 << 10))
                if typ == 254:
                    return out
                if typ != 2:
                    raise ValueError("unexpected pairlist tail type %d" % typ)
        if typ in (10, 13):                  # LGLSXP / INTSXP
            n = self.int(); v = np.frombuffer(self.raw(4 * n), dtype=">i4").astype(np.int32)
        elif typ == 14:                      # REALSXP
            n = self.int(); v = np.frombuffer(self.raw(8 * n), dtype=">f8").astype(np.float64)
        elif typ == 16:                      # STRSXP
            n = self.int(); v = [self.item() for _ in range(n)]
        elif typ == 19:                      # VECSXP
            n = self.int(); v = [self.item() for _ in range(n)]
        else:
            raise ValueError("unsupported SEXP type %d at byte %d" % (typ, self.i))
        attrs = dict(self.item()) if has_attr else {}
        return {"v": v, "attr": attrs}


def main():
    buf = gzip.open(SRC, "rb").read()
    assert buf[:2] == b"X\n", "not an XDR serialisation"
    r = Reader(buf); r.i = 2
    version, _, _ = r.int(), r.int(), r.int()
    if version >= 3:
        r.raw(r.int())                       # native encoding name
    top = r.item()
    names = top["attr"]["names"]["v"]
    out = {}
    for name, el in zip(names, top["v"]):
        dim = el["attr"].get("dim")
        a = np.asarray(el["v"])
        if dim is not None:
            a = a.reshape(tuple(int(x) for x in dim["v"]), order="F")
        out[name] = a
    for k, v in out.items():
        print(k, v.dtype, v.shape, float(v.sum()))
    np.savez_compressed(DST, **out)
    print("wrote", DST, os.path.getsize(DST), "bytes")


if __name__ == "__main__":
    sys.exit(main())
