"""Per-factor stamps of k_mh_prow's row 0 (libbnmf_zsprof.so, built with -DZSPROF): where a factor step of the P-side sweep of the
MH / Normal models goes.  CONV=1: with the Metropolis-Hastings step (after convergence)."""
import ctypes as C
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bayesnmf_amd.engine as E  # noqa: E402
E.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "bin", "libbnmf_zsprof.so")
from bayesnmf_amd.setup import apply_hyperprior_params, synth_counts  # noqa: E402
M, _, _ = synth_counts(96, 5000, 8, 20250221)
conv = os.environ.get("CONV", "0") == "1"
e = E.Engine(M, 20, prior="truncnormal", MH=True, seed=1, window=0)
apply_hyperprior_params(e, "truncnormal", M, 20)
e.init()
e.run(30, converged=conv, metrics=False)
L = E.lib()
W = 4096
out = (C.c_ulonglong * (8 * W))()
L.bnmf_debug_draw.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
L.bnmf_debug_draw(e._h, out, 1)
n = 50
e.run(n, converged=conv, metrics=False)
L.bnmf_debug_draw(e._h, out, 1)
a = np.array(list(out), dtype=np.float64).reshape(W, 8)[2048:2048 + 20]
names = ["cells + trees", "barrier 1", "segment sums + draw (lane 0)", "barrier 2", "update (+ MH step)", "whole step"]
print(f"k_mh_prow row 0, converged={conv}: 10 ns ticks per factor step, mean over {n} iterations and the 20 factors")
for i, nm in enumerate(names):
    print(f"  {nm:32s} {a[:, i].sum() / a[:, 7].sum() / 100.0:7.3f} us")
b = np.array(list(out), dtype=np.float64).reshape(W, 8)[2048 + 64:2048 + 64 + 20]
su = np.array(list(out), dtype=np.float64).reshape(W, 8)[2048 + 128]
if b[:, 7].sum() > 0:
    print(f"k_mh_ecol16 workgroup 0, wave 0 (4 columns), converged={conv}: per factor step")
    for i, nm in ((0, "prefetch + cells (before the trees)"), (1, "cells + trees"), (2, "draw"), (3, "MH step + update + fence"), (5, "whole step")):
        print(f"  {nm:36s} {b[:, i].sum() / b[:, 7].sum() / 100.0:7.3f} us")
    print(f"  {'set-up before the first step':36s} {su[0] / max(su[7], 1) / 100.0:7.3f} us")
allr = np.array(list(out), dtype=np.float64).reshape(W, 8)
for i, row in enumerate((0, 50, 95)):
    r = allr[2048 + 129 + i]
    if r[7] > 0: print(f"k_mh_prow row {row}: set-up before the first factor {r[0] / r[7] / 100.0:7.2f} us (of it the row's P, A, flags into LDS + barrier {r[2] / r[7] / 100.0:5.2f}), the {20} factor steps {r[1] / r[7] / 100.0:7.2f} us")
