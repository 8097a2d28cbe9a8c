"""In-kernel section timers of k_zalloc_step (libbnmf_zpprof.so, built with -DZPPROF): share of the waves' time per section.
Build: hipcc <Makefile flags> -DZPPROF -o tools/bin/libbnmf_zpprof.so bayesnmf_amd/csrc/api.hip     CFG=4|5  G5=columns of config 5"""
import ctypes as C
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bayesnmf_amd.engine as E  # noqa: E402
E.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "bin", "libbnmf_zpprof.so")
from bayesnmf_amd.setup import apply_hyperprior_params, synth_counts  # noqa: E402

cfg = os.environ.get("CFG", "4")
if cfg == "4":
    K, G, N, R, seed, kw = 96, 10000, 50, 12, 20250222, dict(learning_rank=True, temperature=np.ones(8000))
else:
    K, G, N, R, seed, kw = 1536, int(os.environ.get("G5", "12800")), 100, 30, 20250223, {}
M, _, _ = synth_counts(K, G, R, seed)
e = E.Engine(M, N, prior="gamma", seed=1, window=0, **kw)
apply_hyperprior_params(e, "gamma", M, N)
e.init()
e.run(30 if cfg == "4" else 3, metrics=False)
L = E.lib()
out = (C.c_ulonglong * 8)()
L.bnmf_debug_zsort.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
L.bnmf_debug_zsort(e._h, out)
n = 10 if cfg == "4" else 3
prof = e.profile(n)
L.bnmf_debug_zsort(e._h, out)
v = np.array(list(out), dtype=np.float64)
waves = v[7] / n
names = ["staging + end of step", "pass A (Mhat)", "pass B (thresholds)", "quad loops", "histogram flush", "barrier wait", "whole", "waves"]
print(f"cfg {cfg} K={K} G={G} N={N}: k_zalloc {prof['k_zalloc'] * 1e3:.1f} us per launch; {waves:.0f} waves per launch; ticks are s_memtime shader cycles")
for i in range(7):
    print(f"  {names[i]:24s} {v[i] / v[7]:10.0f} cycles per wave  {100 * v[i] / v[6]:5.1f} %")
