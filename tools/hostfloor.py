"""Host floor of one Gibbs iteration: the same sweep at G = 200 (GPU work negligible), 2,000 and 10,000.  The runtime calls
per iteration are the same in all three; what the smallest takes per iteration is what the host needs to enqueue one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayesnmf_amd import Engine
from bayesnmf_amd.setup import synth_counts, apply_hyperprior_params
for G in (200, 2000, 10000):
    M, _, _ = synth_counts(96, G, 8, 20250218)
    e = Engine(M, 20, prior="gamma", seed=1, window=int(os.environ.get("WINDOW", "1000"))); apply_hyperprior_params(e, "gamma", M, 20); e.init(); e.run(300, metrics=False)
    ts = []
    for _ in range(4):
        t0 = time.perf_counter(); e.run(2000, metrics=True); ts.append((time.perf_counter() - t0) / 2000 * 1e6)
    print(f"G={G:6d}: {min(ts):6.1f} us per iteration (min of 4 x 2000)", flush=True)
    e.close()
