"""Host-side phases of bnmf_run(20) calls at the metric configuration, or at G columns (BNMF_RUNCLOCK=1 makes the library print them)."""
import os, sys, time
os.environ["BNMF_RUNCLOCK"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bayesnmf_amd.engine as E
if len(sys.argv) > 1: E.LIB_PATH = os.path.abspath(sys.argv[1])
from bayesnmf_amd.setup import synth_counts, apply_hyperprior_params
G = int(os.environ.get("G", "10000"))
M, _, _ = synth_counts(96, G, 8, 20250218)
e = E.Engine(M, 20, prior="gamma", seed=1, window=1000); apply_hyperprior_params(e, "gamma", M, 20); e.init(); e.run(600, metrics=False)
for _ in range(6):
    t0 = time.perf_counter(); e.run(20, metrics=True); print(f"python: {1e6 * (time.perf_counter() - t0):.1f} us", file=sys.stderr)
e.run(2000, metrics=True)
