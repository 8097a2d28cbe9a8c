"""Interleaved A/B timing of k_zalloc variants in ONE process (cdna guide rule 24)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bayesnmf_amd.engine as _E
if os.environ.get('ABL_LIB'): _E.LIB_PATH = os.path.abspath(os.environ['ABL_LIB'])
from bayesnmf_amd import Engine
from bayesnmf_amd.setup import synth_counts, apply_hyperprior_params
M, _, _ = synth_counts(96, 10000, 8, 20250218)
def mk(env):
    for k in ("BNMF_ABLATE", "BNMF_ZGRID", "BNMF_ZW"):
        os.environ.pop(k, None)
    os.environ.update(env)
    e = Engine(M, 20, prior="gamma", seed=1); apply_hyperprior_params(e, "gamma", M, 20); e.init(); e.run(20, metrics=False)
    return e
ZW = os.environ.get("ABL_ZW", "16")
variants = [("base", {}), ("nophase2", {"BNMF_ABLATE": "2"}), ("nosearch", {"BNMF_ABLATE": "8"}),
            ("nophilox", {"BNMF_ABLATE": "16"}), ("noatomic", {"BNMF_ABLATE": "4"}), ("none", {"BNMF_ABLATE": "28"}),
            ("empty", {"BNMF_ABLATE": "32"}), ("empty_noflush", {"BNMF_ABLATE": "33"}),
            ("noflush", {"BNMF_ABLATE": "1"}), ("nophase2_noflush", {"BNMF_ABLATE": "3"})]
variants = [(n, dict(e, BNMF_ZW=ZW)) for n, e in variants]
if len(sys.argv) > 1:
    variants = [v for v in variants if v[0] in sys.argv[1:]] or variants
eng = [(n, mk(env)) for n, env in variants]
res = {n: [] for n, _ in eng}
for rnd in range(5):
    for n, e in eng:
        res[n].append(e.profile(40)["k_zalloc"] * 1e3)
for n, _ in eng:
    v = np.array(res[n]); print(f"{n:28s} median {np.median(v):7.1f}  min {v.min():7.1f} us", flush=True)
