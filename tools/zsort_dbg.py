import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesnmf_amd import Engine
from bayesnmf_amd.setup import apply_hyperprior_params
def mk(M, N, zs, **kw):
    os.environ["BNMF_ZSORT"] = str(zs)
    e = Engine(M, N, prior="gamma", seed=5, **kw)
    apply_hyperprior_params(e, "gamma", M, N)
    return e
for K, N in [(7, 2), (96, 5), (130, 20)]:
    rng = np.random.default_rng(K)
    M = rng.poisson(rng.gamma(1.0, 40.0, size=(K, 1))).astype(np.int32)
    e0, e1 = mk(M, N, 0, save_Z=True), mk(M, N, 1)
    e0.init(); e1.init()
    Z = e0.get("Z")[:, :, 0]
    ZG = e1.get("ZsumG")
    print("K", K, "N", N)
    for k in range(min(K, 40)):
        print(k, "m", M[k, 0], "nq", (M[k,0]+3)//4, "Z", Z[k].tolist(), "got", ZG[k].tolist(), "ratio", ZG[k].sum() / max(1, M[k, 0]))
