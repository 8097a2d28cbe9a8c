"""Config 4 (Poisson-Gamma SBFI learned rank 1:50, K=96, G=10,000): µs per iteration over a few calls of 300 iterations."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayesnmf_amd import Engine
from bayesnmf_amd.setup import synth_counts, apply_hyperprior_params
M, _, _ = synth_counts(96, 10000, 8, 20250221)
e = Engine(M, 50, prior="gamma", seed=1, learning_rank=True, rank_method="SBFI", temperature=np.ones(4000), window=1000)
apply_hyperprior_params(e, "gamma", M, 50); e.init(); e.run(100, metrics=False)
v = []
for _ in range(5):
    t0 = time.perf_counter(); e.run(300, metrics=True); v.append((time.perf_counter() - t0) / 300 * 1e6)
print("config 4: " + " ".join("%.1f" % x for x in v) + f" us/iteration; median {np.median(v):.1f} = {1e6/np.median(v):.0f} it/s; rank now {int(e.get('A').sum())}")
