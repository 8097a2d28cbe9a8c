"""Diagnostic: iteration time of config 4 / 5 with k_zalloc_step and with the tile kernel (BNMF_ZSTEP=0).  CFG=4|5, G5=columns of config 5"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesnmf_amd import Engine  # noqa: E402
from bayesnmf_amd.setup import apply_hyperprior_params, synth_counts  # noqa: E402

cfg = os.environ.get("CFG", "4")
if cfg == "4":
    K, G, N, R, seed, iters, kw = 96, 10000, 50, 12, 20250222, 100, dict(learning_rank=True, temperature=np.ones(8000))
else:
    K, G, N, R, seed, iters, kw = 1536, int(os.environ.get("G5", "12800")), 100, 30, 20250223, int(os.environ.get("ITERS", "5")), {}
M, _, _ = synth_counts(K, G, R, seed)
for zstep in os.environ.get("ORDER", "1,0,1,0").split(","):
    os.environ["BNMF_ZSTEP"] = zstep
    t0 = time.perf_counter()
    e = Engine(M, N, prior="gamma", seed=1, window=2, **kw)
    tc = time.perf_counter() - t0
    apply_hyperprior_params(e, "gamma", M, N); e.init()
    e.run(max(3, iters // 3), metrics=False)
    ts = []
    for rep in range(3):
        t0 = time.perf_counter(); e.run(iters, metrics=False); ts.append((time.perf_counter() - t0) / iters)
    prof = e.profile(3)
    print(f"cfg {cfg} zstep={zstep}: create {tc:.2f} s, ms/iter {[round(x * 1e3, 3) for x in ts]}, k_zalloc {prof['k_zalloc'] * 1e3:.1f} us, k_rank {prof.get('k_rank', 0) * 1e3:.1f} us, "
          f"rank {e.get('A').sum():.0f}", flush=True)
    e.close()
