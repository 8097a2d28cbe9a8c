import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from bayesnmf_amd import Engine
from bayesnmf_amd.setup import synth_counts, apply_hyperprior_params
M, _, _ = synth_counts(96, 10000, 8, 20250218)
for abl in ("0", "2", "32", "33"):
    os.environ["BNMF_ABLATE"] = abl
    e = Engine(M, 20, prior="gamma", seed=1); apply_hyperprior_params(e, "gamma", M, 20); e.init(); e.run(50, metrics=False)
    for rep in range(3):
        p = e.profile(20)
        print(abl, {k: round(v * 1e3, 1) for k, v in p.items()}, flush=True)
    t0 = time.perf_counter(); e.run(500, metrics=False); print(abl, "e2e us/iter", (time.perf_counter() - t0) / 500 * 1e6, flush=True)
    e.close()
