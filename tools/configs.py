"""Throughput of the BASELINE.json configurations 2-5 on one MI355X (config 5 at reduced G by default)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayesnmf_amd import Engine
from bayesnmf_amd.setup import synth_counts, apply_hyperprior_params
from bayesnmf_amd.sampler import get_temp_sched_
def bench(name, K, G, N, prior, iters, **kw):
    M, _, _ = synth_counts(K, G, min(8, N), 20250218)
    e = Engine(M, N, prior=prior, seed=1, **kw); apply_hyperprior_params(e, prior, M, N); e.init()
    conv = kw.get("MH", False)
    e.run(max(2, iters // 5), metrics=False)
    t0 = time.perf_counter(); e.run(iters, metrics=True); dt = time.perf_counter() - t0
    line = f"{name:58s} {iters / dt:9.1f} it/s  ({dt / iters * 1e3:8.3f} ms/iter)"
    if conv:
        t0 = time.perf_counter(); e.run(iters, converged=True, metrics=True); dt = time.perf_counter() - t0
        line += f"   after convergence (true MH): {iters / dt:8.1f} it/s"
    print(line, flush=True); e.close()
G5 = int(os.environ.get("CFG5_G", "5000"))
bench("2: Poisson-Gamma N=20, K=96 x G=2,000", 96, 2000, 20, "gamma", 2000)
bench("metric: Poisson-Gamma N=20, K=96 x G=10,000", 96, 10000, 20, "gamma", 2000)
bench("3: Poisson-TruncNormal+MH N=20, K=96 x G=5,000", 96, 5000, 20, "truncnormal", 100, MH=True)
temp = np.ones(4000)
bench("4: Poisson-Gamma SBFI learned rank (N=50), K=96 x G=10,000", 96, 10000, 50, "gamma", 100, learning_rank=True, rank_method="SBFI", temperature=temp)
bench(f"5: Poisson-Gamma N=100, K=1,536 x G={G5:,} (of 50,000)", 1536, G5, 100, "gamma", 10)
