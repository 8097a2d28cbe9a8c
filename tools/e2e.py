import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayesnmf_amd import Engine
from bayesnmf_amd.setup import synth_counts, apply_hyperprior_params
M, _, _ = synth_counts(96, 10000, 8, 20250218)
def run(env):
    for k in ("BNMF_ABLATE", "BNMF_ZGRID", "BNMF_ZW"):
        os.environ.pop(k, None)
    os.environ.update(env)
    e = Engine(M, 20, prior="gamma", seed=1); apply_hyperprior_params(e, "gamma", M, 20); e.init(); e.run(200, metrics=False)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); e.run(1000, metrics=True); ts.append((time.perf_counter() - t0) / 1000 * 1e6)
    e.close(); return min(ts)
for name, env in [("zw10", {"BNMF_ZW": "10"}), ("zw8", {"BNMF_ZW": "8"}), ("zw6", {"BNMF_ZW": "6"}), ("zw4", {"BNMF_ZW": "4"}), ("zw8 grid512", {"BNMF_ZW": "8", "BNMF_ZGRID": "512"}), ("zw4 grid512", {"BNMF_ZW": "4", "BNMF_ZGRID": "512"})]:
    print(f"{name:10s} {run(env):8.1f} us/iter", flush=True)
