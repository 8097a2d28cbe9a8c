"""k_zalloc time vs waves per CU (interleaved in one process)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayesnmf_amd import Engine
from bayesnmf_amd.setup import synth_counts, apply_hyperprior_params
M, _, _ = synth_counts(96, 10000, 8, 20250218)
def mk(env):
    for k in ("BNMF_ABLATE", "BNMF_ZGRID", "BNMF_ZW"): os.environ.pop(k, None)
    os.environ.update(env)
    e = Engine(M, 20, prior="gamma", seed=1); apply_hyperprior_params(e, "gamma", M, 20); e.init(); e.run(20, metrics=False)
    return e
variants = [("2 waves/CU (zw2 g256)", {"BNMF_ZW": "2", "BNMF_ZGRID": "256"}), ("4 waves/CU (zw4 g256)", {"BNMF_ZW": "4", "BNMF_ZGRID": "256"}),
            ("8 waves/CU (zw8 g256)", {"BNMF_ZW": "8", "BNMF_ZGRID": "256"}), ("8 waves/CU (zw4 g512)", {"BNMF_ZW": "4", "BNMF_ZGRID": "512"}),
            ("4 waves/CU nophase2", {"BNMF_ZW": "4", "BNMF_ZGRID": "256", "BNMF_ABLATE": "2"}), ("8 waves/CU nophase2", {"BNMF_ZW": "8", "BNMF_ZGRID": "256", "BNMF_ABLATE": "2"})]
eng = [(n, mk(env)) for n, env in variants]
res = {n: [] for n, _ in eng}
for rnd in range(4):
    for n, e in eng: res[n].append(e.profile(30)["k_zalloc"] * 1e3)
for n, _ in eng: print(f"{n:28s} median {np.median(res[n]):7.1f} us", flush=True)
