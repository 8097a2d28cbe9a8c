"""Times of the post-processing calls at the metric configuration (window of 1000 samples, rank 20): bnmf_map with and without
credible bounds, and bnmf_assign against the 79 COSMIC references (cosines + one assignment problem per sample on the device, votes on the
host)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayesnmf_amd import Engine
from bayesnmf_amd.setup import synth_counts, apply_hyperprior_params
ref = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "cosmic_v3.3.1_sbs.npz"))["P"]
M, _, _ = synth_counts(96, 10000, 8, 20250218)
e = Engine(M, 20, prior="gamma", seed=1, window=1000); apply_hyperprior_params(e, "gamma", M, 20); e.init(); e.run(1100, metrics=False)
for ci in (0.95, None):
    for rep in range(4):
        t0 = time.perf_counter()
        m = e.map(1000, ci)
        print(f"bnmf_map(1000 samples, credible_interval = {ci}): {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
for rep in range(4):
    t0 = time.perf_counter()
    a = e.assign(1000, ref, used=m["used"].astype(np.int32), keep=np.ones(20, dtype=np.int32), MAP_P=m["P"], credible_interval=0.95)
    print(f"bnmf_assign(1000 samples, 20 x {ref.shape[1]}): {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
