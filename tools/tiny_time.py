"""µs per iteration and the kernels' device times at tutorial-scale sizes (K = 96, G = 64 .. 1,000)."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayesnmf_amd import Engine
from bayesnmf_amd.setup import synth_counts, apply_hyperprior_params
for G, N in ((64, 5), (64, 20), (500, 10), (1000, 20)):
    M, _, _ = synth_counts(96, G, 4, 3)
    e = Engine(M, N, prior="gamma", seed=1, window=100); apply_hyperprior_params(e, "gamma", M, N); e.init(); e.run(200)
    t0 = time.perf_counter(); e.run(3000); dt = (time.perf_counter() - t0) / 3000 * 1e6
    pr = e.profile(30)
    print(f"G={G} N={N}: {dt:.1f} us/iteration ({1e6/dt:.0f} it/s); kernels alone: " + " ".join(f"{k}={v*1e3:.1f}" for k, v in pr.items() if v > 0))
    e.close()
