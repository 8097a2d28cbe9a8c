"""In-kernel section timers of k_zalloc_reg (libbnmf_zprof.so, built with -DZPROF): prints the share of
wave-cycles per section at the metric config."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bayesnmf_amd.engine as E
E.LIB_PATH = os.path.join(os.path.dirname(E.LIB_PATH), "libbnmf_zprof.so")   # build: hipcc ... -DZPROF -o bayesnmf_amd/libbnmf_zprof.so bayesnmf_amd/csrc/api.hip
from bayesnmf_amd import Engine
from bayesnmf_amd.setup import synth_counts, apply_hyperprior_params
K, G, N = 96, 10000, 20
M, _, _ = synth_counts(K, G, 8, 20250218)
names = {1: "phase 1", 2: "range set-up", 3: "quad loop", 4: "hist flush", 5: "phase 3", 6: "whole column", 7: "prologue", 8: "column @100MHz"}
base = int(os.environ.get("BNMF_ABLATE", "0"))
res = {}
for sel in range(1, 9):
    os.environ["BNMF_ABLATE"] = str(base | (sel << 12))
    e = Engine(M, N, prior="gamma", seed=1); apply_hyperprior_params(e, "gamma", M, N); e.init()
    m = e.run(30)
    cyc = (m[10:, 1] ** 2 * K * G).mean()
    kt = e.profile(20)["k_zalloc"] * 1e3
    res[sel] = cyc
    print(f"{names[sel]:14s} {cyc:14.0f} wave-cycles/launch  kernel {kt:6.1f} us", flush=True)
    e.close()
tot = res[6]
for sel in range(1, 6): print(f"  {names[sel]:14s} {100 * res[sel] / tot:5.1f} % of column time")
print("shader clock during the kernel: %.0f MHz" % (res[6] / res[8] * 100.0))
