"""Run a few iterations of one BASELINE.json configuration (for rocprofv3 --kernel-trace --stats).  CFG=2|3|3c|4|5|m"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bayesnmf_amd.engine as _E
if os.environ.get('BNMF_TEST_LIB'): _E.LIB_PATH = os.path.abspath(os.environ['BNMF_TEST_LIB'])
from bayesnmf_amd import Engine
from bayesnmf_amd.setup import synth_counts, apply_hyperprior_params
cfg = os.environ.get("CFG", "3")
iters = int(os.environ.get("ITERS", "30"))
if cfg in ("3", "3c"):
    M, _, _ = synth_counts(96, 5000, 8, 20250221); N, prior, kw = 20, "truncnormal", dict(MH=True)
elif cfg == "4":
    M, _, _ = synth_counts(96, 10000, 12, 20250222); N, prior, kw = 50, "gamma", dict(learning_rank=True, temperature=np.ones(8000))
elif cfg == "5":
    M, _, _ = synth_counts(1536, int(os.environ.get("G5", "5000")), 30, 20250223); N, prior, kw = 100, "gamma", dict()
elif cfg == "2":
    M, _, _ = synth_counts(96, 2000, 8, 20250218); N, prior, kw = 20, "gamma", dict()
else:
    M, _, _ = synth_counts(96, 10000, 8, 20250218); N, prior, kw = 20, "gamma", dict()
e = Engine(M, N, prior=prior, seed=1, window=int(os.environ.get("WINDOW", "100")), **kw)
apply_hyperprior_params(e, prior, M, N); e.init()
e.run(iters, converged=(cfg == "3c"), metrics=False)
e.close()
