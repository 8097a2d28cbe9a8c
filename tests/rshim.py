"""Test harness: r/bnmf_shim.c compiled against the stand-in R runtime of tests/r_stub/ and driven through ctypes.

`build()` compiles shim + stand-in runtime into tests/r_stub/librshim_test.so (linked against the real libbnmf.so);
`RShim` is `.Call`: routines are reached BY REGISTERED NAME with SEXP arguments, exactly as R reaches them; every call
asserts that the stand-in saw no API violation (wrong accessor, use of a collected object, PROTECT imbalance) and
that the PROTECT stack is empty afterwards.  Test infrastructure only (see tests/r_stub/README.md)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "r_stub")
SO = os.path.join(STUB, "librshim_test.so")
CFLAGS = ["-std=c11", "-Wall", "-Wextra", "-Werror",
          # (DL_FUNC)&routine in an R_CallMethodDef table is the registration idiom of R's own API
          "-Wno-cast-function-type"]
INCLUDES = ["-I" + STUB, "-I" + os.path.join(ROOT, "include")]
NILSXP, LGLSXP, INTSXP, REALSXP, STRSXP, VECSXP, EXTPTRSXP = 0, 10, 13, 14, 16, 19, 22


def syntax_check():
    """gcc -fsyntax-only with every warning an error: the first compiler that ever saw the shim."""
    return subprocess.run(["gcc", *CFLAGS, "-fsyntax-only", *INCLUDES, os.path.join(ROOT, "r", "bnmf_shim.c")],
                          capture_output=True, text=True)


def build(force=False, shim=None, so=None, link_bnmf=True):
    """shim + stand-in runtime -> one shared object (-Bsymbolic: each such library keeps its own runtime state)"""
    shim = shim or os.path.join(ROOT, "r", "bnmf_shim.c")
    so = so or SO
    srcs = [shim, os.path.join(STUB, "rstub.c")]
    deps = srcs + [os.path.join(STUB, f) for f in ("Rinternals.h", "R.h", "R_ext/Rdynload.h")] + [os.path.join(ROOT, "include", "bnmf.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        libdir = os.path.join(ROOT, "bayesnmf_amd")
        link = ["-L" + libdir, "-lbnmf", "-Wl,-rpath," + libdir] if link_bnmf else []
        subprocess.check_call(["gcc", *CFLAGS, "-fPIC", "-shared", "-O1", "-g", "-Wl,-Bsymbolic", *INCLUDES, "-o", so, *srcs, *link])
    return so


class RError(RuntimeError):
    """the routine called Rf_error()"""


class RViolation(AssertionError):
    """the code under test broke a rule of R's C API (reported by the stand-in runtime)"""


class RShim:
    def __init__(self, so=None, init="R_init_bayesNMFhip"):
        L = C.CDLL(so or build())
        vp = C.c_void_p
        for name, res, args in [
                ("rstub_load", C.c_int, [C.c_void_p]), ("rstub_n_routines", C.c_int, []), ("rstub_routine_name", C.c_char_p, [C.c_int]),
                ("rstub_routine_nargs", C.c_int, [C.c_int]), ("rstub_call", C.c_int, [C.c_char_p, C.c_int, C.POINTER(vp), C.POINTER(vp)]),
                ("rstub_nil", vp, []), ("rstub_mk_int", vp, [C.c_long, vp]), ("rstub_mk_lgl", vp, [C.c_long, vp]), ("rstub_mk_real", vp, [C.c_long, vp]),
                ("rstub_mk_int_matrix", vp, [C.c_int, C.c_int, vp]), ("rstub_mk_real_matrix", vp, [C.c_int, C.c_int, vp]),
                ("rstub_typeof", C.c_int, [vp]), ("rstub_xlength", C.c_long, [vp]), ("rstub_nrow", C.c_int, [vp]), ("rstub_ncol", C.c_int, [vp]),
                ("rstub_dataptr", vp, [vp]), ("rstub_elt", vp, [vp, C.c_long]), ("rstub_string", C.c_char_p, [vp, C.c_long]),
                ("rstub_name", C.c_char_p, [vp, C.c_long]), ("rstub_extptr_addr", vp, [vp]), ("rstub_release", None, [vp]), ("rstub_gc", C.c_int, []),
                ("rstub_live_objects", C.c_int, []), ("rstub_protect_depth", C.c_int, []), ("rstub_violations", C.c_int, []),
                ("rstub_finalizers_run", C.c_int, []), ("rstub_collected_in_last_call", C.c_int, []),
                ("rstub_last_error", C.c_char_p, []), ("rstub_last_violation", C.c_char_p, [])]:
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        self.L = L
        rc = L.rstub_load(C.cast(getattr(L, init), C.c_void_p))      # R: dlsym(handle, "R_init_<package>") after dlopen
        assert rc == 0, f"{init}: rc {rc} (-2: no routines registered / dynamic symbols left on)"
        self.routines = {L.rstub_routine_name(i).decode(): L.rstub_routine_nargs(i) for i in range(L.rstub_n_routines())}

    # ---- R values in
    def nil(self):
        return self.L.rstub_nil()

    def integer(self, v):
        a = np.ascontiguousarray(np.atleast_1d(v), dtype=np.int32)
        return self.L.rstub_mk_int(a.size, a.ctypes.data)

    def logical(self, v):
        a = np.ascontiguousarray(np.atleast_1d(v), dtype=np.int32)
        return self.L.rstub_mk_lgl(a.size, a.ctypes.data)

    def real(self, v):
        a = np.ascontiguousarray(np.atleast_1d(v), dtype=np.float64)
        return self.L.rstub_mk_real(a.size, a.ctypes.data)

    def int_matrix(self, m):
        a = np.asfortranarray(m, dtype=np.int32)
        return self.L.rstub_mk_int_matrix(a.shape[0], a.shape[1], a.ctypes.data)

    def real_matrix(self, m):
        a = np.asfortranarray(m, dtype=np.float64)
        return self.L.rstub_mk_real_matrix(a.shape[0], a.shape[1], a.ctypes.data)

    # ---- .Call
    def call(self, name, *args, keep_args=False):
        L = self.L
        arr = (C.c_void_p * len(args))(*args)
        out = C.c_void_p()
        v0 = L.rstub_violations()
        rc = L.rstub_call(name.encode(), len(args), arr, C.byref(out))
        depth = L.rstub_protect_depth()
        err, vio = L.rstub_last_error().decode(), L.rstub_last_violation().decode()
        if not keep_args:
            for a in args:
                if a != L.rstub_nil() and L.rstub_typeof(a) != EXTPTRSXP:
                    L.rstub_release(a)
        assert depth == 0, f".Call({name}): PROTECT stack depth {depth} after the call"
        if rc in (2, 3) or L.rstub_violations() != v0:
            raise RViolation(f".Call({name}): {vio}")
        if rc == 1:
            raise RError(err)
        return out.value

    # ---- R values out
    def to_py(self, x):
        """numpy / python view of a returned SEXP (copied), lists as dicts by name (or python lists)"""
        L = self.L
        t, n = L.rstub_typeof(x), L.rstub_xlength(x)
        if t == NILSXP:
            return None
        if t in (INTSXP, LGLSXP, REALSXP):
            ct = C.c_double if t == REALSXP else C.c_int32
            a = np.ctypeslib.as_array(C.cast(L.rstub_dataptr(x), C.POINTER(ct)), shape=(n,)).copy() if n else np.empty(0, dtype=ct)
            if t == LGLSXP:
                a = a.astype(bool)
            nr, nc = L.rstub_nrow(x), L.rstub_ncol(x)
            return a.reshape((nr, nc), order="F") if nr >= 0 else a
        if t == STRSXP:
            return [L.rstub_string(x, i).decode() for i in range(n)]
        if t == VECSXP:
            vals = [self.to_py(L.rstub_elt(x, i)) for i in range(n)]
            names = [L.rstub_name(x, i) for i in range(n)]
            return {nm.decode(): v for nm, v in zip(names, vals)} if all(names) else vals
        if t == EXTPTRSXP:
            return ("externalptr", L.rstub_extptr_addr(x))
        raise TypeError(f"SEXP type {t}")

    def take(self, x):
        """to_py + drop the reference"""
        v = self.to_py(x)
        self.L.rstub_release(x)
        return v

    def release(self, x):
        self.L.rstub_release(x)

    def gc(self):
        return self.L.rstub_gc()
