"""Config 3 (Poisson-TruncNormal + MH, K=96, G=5,000, N=20): iterations per second with k_mh_tail's work hosted by the sweep kernels (default)
and with k_mh_tail between the sweeps (BNMF_MHPIPE=0), before and after convergence; interleaved in one process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayesnmf_amd import Engine
import bayesnmf_amd.engine as _E
if os.environ.get("BNMF_TEST_LIB"): _E.LIB_PATH = os.path.abspath(os.environ["BNMF_TEST_LIB"])
from bayesnmf_amd.setup import synth_counts, apply_hyperprior_params
M, _, _ = synth_counts(96, 5000, 8, 20250220)
def mk(pipe):
    os.environ["BNMF_MHPIPE"] = pipe
    e = Engine(M, 20, prior="truncnormal", MH=True, seed=1); apply_hyperprior_params(e, "truncnormal", M, 20); e.init(); e.run(50)
    return e
eng = {"hosted": mk("1"), "k_mh_tail": mk("0")}
res = {(n, c): [] for n in eng for c in (0, 1)}
for rnd in range(5):
    for conv in (0, 1):
        for n, e in eng.items():
            t = time.perf_counter(); e.run(300, converged=bool(conv)); res[(n, conv)].append((time.perf_counter() - t) / 300 * 1e6)
for (n, c), v in res.items(): print(f"{n:22s} converged={c}: median {np.median(v):7.1f} us/iteration = {1e6/np.median(v):7.0f} it/s   ({' '.join('%.1f' % x for x in v)})")
a, b = eng["hosted"], eng["k_mh_tail"]
print("same chain:", all(np.array_equal(a.get(nm).view(np.uint64), b.get(nm).view(np.uint64)) for nm in ("P", "E")))
