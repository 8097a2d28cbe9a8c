"""Bit-exact check of a k_zalloc variant (env) against the default variant on the metric shape (short run)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayesnmf_amd import Engine
from bayesnmf_amd.setup import synth_counts, apply_hyperprior_params
M, _, _ = synth_counts(96, 2000, 8, 20250218)
def run(env):
    for k in ("BNMF_ABLATE", "BNMF_ZGRID", "BNMF_ZW"): os.environ.pop(k, None)
    os.environ.update(env)
    e = Engine(M, 20, prior="gamma", seed=1); apply_hyperprior_params(e, "gamma", M, 20); e.init()
    m = e.run(20); r = (m[:, :9].copy(), e.get("P"), e.get("E"), e.get("ZsumK")); e.close(); return r
ref = run({})
for name, env in [("zw4", {"BNMF_ZW": "4"}), ("zw2 grid 512", {"BNMF_ZW": "2", "BNMF_ZGRID": "512"})]:
    cur = run(env)
    print(name, "bit-identical:", all(np.array_equal(a, b) for a, b in zip(ref, cur)), flush=True)
