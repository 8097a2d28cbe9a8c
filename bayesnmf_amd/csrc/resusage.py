"""Developer tool: one line per kernel from `make -C bayesnmf_amd/csrc resource` (VGPRs, scratch, occupancy, LDS).
Usage: make -C bayesnmf_amd/csrc resource > /tmp/resource.txt; python bayesnmf_amd/csrc/resusage.py /tmp/resource.txt [filter ...]"""
import re
import subprocess
import sys

text = open(sys.argv[1]).read()
filters = sys.argv[2:]
rows = []
for b in re.split(r"(?=remark: Function Name)", text):
    m = re.search(r"Function Name: (\S+)", b)
    if not m:
        continue
    g = lambda k: int((re.search(k + r": (\d+)", b) or [0, "0"])[1])  # noqa: E731
    rows.append((m.group(1), g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
for r, n in zip(rows, names):
    n = re.sub(r"\((bnmf::|unsigned|int|double).*", "", n).replace("void bnmf::", "").replace("bnmf::", "")
    if filters and not any(f in n for f in filters) and not (("scratch" in filters) and r[3] > 0):
        continue
    print(f"{n[:70]:70s} vgpr {r[1]:3d} agpr {r[2]:3d} scratch {r[3]:4d} occ {r[4]} lds {r[5]}")
