"""Config 3 alone (Poisson-TruncNormal + MH, N = 20, K = 96, G = 5,000): iterations per second before / after convergence, three runs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("BNMF_TOOL_LIB"):                      # another build of the library (A/B)
    import bayesnmf_amd.engine as _E
    _E.LIB_PATH = os.path.abspath(os.environ["BNMF_TOOL_LIB"])
from bayesnmf_amd import Engine
from bayesnmf_amd.setup import synth_counts, apply_hyperprior_params
M, _, _ = synth_counts(96, 5000, 8, 20250218)
e = Engine(M, 20, prior="truncnormal", seed=1, MH=True); apply_hyperprior_params(e, "truncnormal", M, 20); e.init()
e.run(50, metrics=False)
for rep in range(3):
    t0 = time.perf_counter(); e.run(200, metrics=True); d0 = time.perf_counter() - t0
    t0 = time.perf_counter(); e.run(200, converged=True, metrics=True); d1 = time.perf_counter() - t0
    print(f"config 3: {200 / d0:8.1f} it/s ({d0 / 200 * 1e6:6.1f} us)   after convergence {200 / d1:8.1f} it/s ({d1 / 200 * 1e6:6.1f} us)", flush=True)
e.close()
