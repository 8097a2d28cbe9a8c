"""Where the time goes in create -> init -> run -> destroy -> create ... at the metric configuration (the 8-13 ms wait at the first
synchronisation after a bnmf_create that follows a bnmf_destroy): wall time of every C-ABI call of three lives of a
handle in one process; BNMF_TIMING=1 adds bnmf_create's own marks on stderr."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bayesnmf_amd.engine as _E
if os.environ.get("BNMF_TEST_LIB"): _E.LIB_PATH = os.path.abspath(os.environ["BNMF_TEST_LIB"])
from bayesnmf_amd import Engine
from bayesnmf_amd.setup import synth_counts, apply_hyperprior_params
M, _, _ = synth_counts(96, 10000, 8, 20250218)
W = int(os.environ.get("WINDOW", "1000"))
def tm(f):
    t0 = time.perf_counter(); r = f(); return r, 1e3 * (time.perf_counter() - t0)
for life in range(4):
    e, t_create = tm(lambda: Engine(M, 20, prior="gamma", seed=1, window=W))
    _, t_hyper = tm(lambda: apply_hyperprior_params(e, "gamma", M, 20))
    _, t_init = tm(e.init)
    _, t_r1 = tm(lambda: e.run(20))
    _, t_r2 = tm(lambda: e.run(20))
    _, t_r3 = tm(lambda: e.run(200))
    _, t_close = tm(e.close)
    print(f"life {life}: create {t_create:7.2f}  hyper-priors {t_hyper:6.2f}  init {t_init:6.2f}  run(20) {t_r1:6.2f}  run(20) {t_r2:6.2f}  run(200) {t_r3:6.2f}  destroy {t_close:6.2f} ms", flush=True)
