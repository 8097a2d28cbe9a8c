import sys, numpy as np
sys.path.insert(0, "/root/repo")
import oracle as O
import os
import bayesnmf_amd.engine as _E
if os.environ.get("DBG_LIB"): _E.LIB_PATH = os.environ["DBG_LIB"]
from bayesnmf_amd import Engine
from bayesnmf_amd.setup import apply_hyperprior_params
rng = np.random.default_rng(8)
def one(K, G, N, tag):
    M = rng.poisson(rng.gamma(0.5, 20.0, size=(K, G))).astype(np.int32)
    M[:, G // 2] = 0
    M[K // 2, :] = 0
    o = O.Oracle(M, N, prior="gamma", seed=4, save_Z=True, nthreads=4)
    apply_hyperprior_params(o, "gamma", M, N)
    o.init(); mo = o.run(6)
    e = Engine(M, N, prior="gamma", seed=4, save_Z=True)
    apply_hyperprior_params(e, "gamma", M, N)
    e.init()
    me = e.run(6)
    bad = [nm for nm in ["P", "E", "ZsumK", "ZsumG", "Alpha_e", "Beta_e", "Alpha_p", "Beta_p"] if not np.array_equal(np.asarray(o.get(nm), dtype=np.float64), np.asarray(e.get(nm), dtype=np.float64))]
    rows = [i for i in range(6) if not np.array_equal(mo[i, :9], me[i, :9])]
    print(tag, (K, G, N), "bad arrays", bad, "bad metric rows", rows, flush=True)
one(96, 64, 20, "warm")
for rep in range(3):
    one(5, 3, 1, "tiny")
    one(70, 9, 2, "small")
