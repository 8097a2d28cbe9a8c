"""In-kernel section timers of k_zalloc_sort (libbnmf_zsprof.so, built with -DZSPROF): share of the waves' time per section.
Build: hipcc <Makefile flags> -DZSPROF -o tools/bin/libbnmf_zsprof.so bayesnmf_amd/csrc/api.hip"""
import ctypes as C
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bayesnmf_amd.engine as E  # noqa: E402
E.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "bin", "libbnmf_zsprof.so")
from bayesnmf_amd.setup import apply_hyperprior_params, synth_counts  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
M, _, _ = synth_counts(96, G, 8, 20250218)
e = E.Engine(M, 20, prior="gamma", seed=1, window=0)
apply_hyperprior_params(e, "gamma", M, 20)
e.init()
e.run(50, metrics=False)
L = E.lib()
out = (C.c_ulonglong * 8)()
L.bnmf_debug_zsort.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
L.bnmf_debug_zsort(e._h, out)
n = 20
prof = e.profile(n)
L.bnmf_debug_zsort(e._h, out)
v = np.array(list(out), dtype=np.float64)
waves = v[7] / n
names = ["setup", "thresholds", "quad loops", "flush", "metric tasks", "end barrier+epilogue", "whole", "waves"]
print(f"G={G}: k_zalloc {prof['k_zalloc'] * 1e3:.1f} us per launch; {waves:.0f} waves per launch; ticks are s_memtime shader cycles")
for i in range(7):
    print(f"  {names[i]:24s} {v[i] / v[7]:10.0f} cycles per wave  {100 * v[i] / v[6]:5.1f} %")
