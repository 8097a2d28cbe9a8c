"""k_zalloc_sort at small G: device time per launch (profile mode: HIP events, one kernel at a time) against G and the waves per workgroup."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayesnmf_amd import Engine
import bayesnmf_amd.engine as _E
if os.environ.get("BNMF_TEST_LIB"): _E.LIB_PATH = os.path.abspath(os.environ["BNMF_TEST_LIB"])
from bayesnmf_amd.setup import synth_counts, apply_hyperprior_params
def run(G, env):
    for k in ("BNMF_ZSW", "BNMF_ZSPK", "BNMF_ZSIT16", "BNMF_ZSLDS", "BNMF_ZSQMAX"): os.environ.pop(k, None)
    os.environ.update(env)
    M, _, _ = synth_counts(96, G, 8, 20250218)
    e = Engine(M, 20, prior="gamma", seed=1); apply_hyperprior_params(e, "gamma", M, 20); e.init(); e.run(30, metrics=False)
    v = [e.profile(40)["k_zalloc"] * 1e3 for _ in range(3)]
    q = int(e.stat(5))
    e.close()
    return min(v) if env else (min(v), q)
variants = (("default", {}),) + tuple((f"q={q}", {"BNMF_ZSQMAX": str(q)}) for q in (64, 32, 16, 8, 4))
if os.environ.get("ZSMALL_DEFAULT_ONLY"): variants = variants[:1]
for G in (250, 500, 1000, 2000, 4000, 10000):
    d, q = run(G, {})
    print(f"G={G:6d}: default (q={q}) {d:6.1f}  " + "  ".join(f"{tag} {run(G, env):6.1f}" for tag, env in variants[1:]), flush=True)
