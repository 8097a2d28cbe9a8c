"""Long runs of the three sweep types: no time-out of the in-kernel hand-offs, finite metrics (diagnostics)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayesnmf_amd import Engine
from bayesnmf_amd.setup import synth_counts, apply_hyperprior_params
def soak(name, K, G, N, iters, block, **kw):
    M, _, _ = synth_counts(K, G, 8, 424242)
    e = Engine(M, N, seed=3, window=kw.pop("window", 100), **kw); apply_hyperprior_params(e, kw["prior"], M, N); e.init()
    t0 = time.perf_counter(); done = 0
    while done < iters:
        m = e.run(block, converged=done > iters // 2); done += block
        assert np.all(np.isfinite(m[:, 1:5])), (name, done)
    print("%-28s %7d iterations ok, %.0f it/s" % (name, done, done / (time.perf_counter() - t0)), flush=True)
    e.close()
soak("gamma fixed rank (metric)", 96, 10000, 20, 150000, 5000, prior="gamma")
soak("gamma SBFI rank 1:50", 96, 10000, 50, 15000, 1000, prior="gamma", learning_rank=True, temperature=np.linspace(0.001, 1, 4000))
soak("truncnormal + MH", 96, 5000, 20, 30000, 2000, prior="truncnormal", MH=True)
soak("normal likelihood + rank", 96, 2000, 12, 10000, 1000, prior="exponential", likelihood="normal", learning_rank=True, temperature=np.linspace(0.001, 1, 3000))
