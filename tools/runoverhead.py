"""Fixed cost of one bnmf_run call (pipeline fill, final reduce / compose, copies, synchronisation): time of runs of 5 .. 2000
iterations at the metric configuration, least-squares line through them."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayesnmf_amd import Engine
from bayesnmf_amd.setup import synth_counts, apply_hyperprior_params
M, _, _ = synth_counts(96, 10000, 8, 20250218)
e = Engine(M, 20, prior="gamma", seed=1, window=1000); apply_hyperprior_params(e, "gamma", M, 20); e.init(); e.run(300, metrics=False)
ns, ts = [5, 10, 20, 50, 100, 400, 2000], []
for n in ns:
    reps = max(3, 400 // n)
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); e.run(n, metrics=True); best = min(best, time.perf_counter() - t0)
    ts.append(best * 1e6)
    print(f"run({n:5d}): {best * 1e6:9.1f} us = {best * 1e6 / n:7.1f} us per iteration", flush=True)
a, b = np.polyfit(ns, ts, 1)
print(f"per iteration {a:.1f} us, fixed per call {b:.1f} us")
