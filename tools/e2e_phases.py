"""Where the time of one bayesNMF() call goes at the metric configuration (default convergence control): engine creation (static
schedule), init, the sampling loop with its MAP checks, the final MAP with credible bounds, the saved object."""
import cProfile, io, json, os, pstats, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayesnmf_amd import Engine
from bayesnmf_amd.sampler import bayesNMF
from bayesnmf_amd.convergence import new_convergence_control
from bayesnmf_amd.setup import synth_counts, apply_hyperprior_params
M, _, _ = synth_counts(96, int(os.environ.get("G", "10000")), 8, 20250218)
out = {}
t0 = time.perf_counter(); e = Engine(M, 20, prior="gamma", seed=1, window=1000); out["create_ms"] = 1e3 * (time.perf_counter() - t0)
apply_hyperprior_params(e, "gamma", M, 20)
t0 = time.perf_counter(); e.init(); out["init_ms"] = 1e3 * (time.perf_counter() - t0)
t0 = time.perf_counter(); e.run(1000, metrics=True); out["run1000_ms"] = 1e3 * (time.perf_counter() - t0)
t0 = time.perf_counter(); e.run(1000, metrics=True); out["run1000_again_ms"] = 1e3 * (time.perf_counter() - t0)
for ci in (0.95, None):
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); e.map(1000, ci); ts.append(time.perf_counter() - t0)
    out[f"bnmf_map_ms_ci={ci}"] = 1e3 * float(np.median(ts))
e.close()
d = tempfile.mkdtemp()
for rep in range(2):
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    s = bayesNMF(M, 20, prior="gamma", convergence_control=new_convergence_control(), output_dir=os.path.join(d, f"o{rep}"), periodic_save=False, overwrite=True, save_all_samples=False)
    pr.disable()
    dt = time.perf_counter() - t0
    out[f"bayesNMF_{rep}"] = dict(iters=int(s.state["iter"]), seconds=dt, it_per_s=s.state["iter"] / dt)
    s.close()
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("cumulative").print_stats(28)
print(json.dumps(out))
print(st.getvalue()[:6000])
