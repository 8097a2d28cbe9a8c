"""End-to-end throughput of bayesNMF() (the reference's user call, default convergence control) on one MI355X at the
metric configuration, next to the bare kernel loop: what a caller gets, with MAP checks (bnmf_map) every 100 iterations."""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayesnmf_amd import Engine
from bayesnmf_amd.sampler import bayesNMF
from bayesnmf_amd.convergence import new_convergence_control
from bayesnmf_amd.setup import synth_counts, apply_hyperprior_params
G = int(os.environ.get("G", "10000"))
M, _, _ = synth_counts(96, G, 8, 20250218)
out = {}
e = Engine(M, 20, prior="gamma", seed=1, window=1000); apply_hyperprior_params(e, "gamma", M, 20); e.init(); e.run(1100, metrics=False)
for ci in (0.95, None):
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); e.map(1000, ci); ts.append(time.perf_counter() - t0)
    out[f"bnmf_map_ms_ci={ci}"] = 1e3 * float(np.median(ts))
t0 = time.perf_counter(); P = e.window("P", 1000); E = e.window("E", 1000); out["window_copy_PE_ms"] = 1e3 * (time.perf_counter() - t0)
del P, E
e.close()
d = tempfile.mkdtemp()
for mode, kw in (("default cc, save_all_samples=False", dict(save_all_samples=False)),):
    cc = new_convergence_control()
    t0 = time.perf_counter()
    s = bayesNMF(M, 20, prior="gamma", convergence_control=cc, output_dir=os.path.join(d, "o"), periodic_save=False, overwrite=True, **kw)
    dt = time.perf_counter() - t0
    out[mode] = dict(iters=int(s.state["iter"]), seconds=dt, it_per_s=s.state["iter"] / dt, why=s.state.get("why"),
                     MAP_checks=len(s.state["MAP_metrics"]))
    s.close()
print(json.dumps(out))
