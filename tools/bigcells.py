"""Cost of a few very large cells under the sorted schedule, with and without the export of their fragments (BNMF_ZSSPREAD)."""
import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from bayesnmf_amd import Engine
from bayesnmf_amd.setup import apply_hyperprior_params
rng = np.random.default_rng(5)
K, G, N = 96, 3000, 5
lam = rng.gamma(0.7, 30.0, size=(K, G))
M0 = np.asfortranarray(rng.poisson(lam).astype(np.int32))
for label, edit in (("plain", lambda M: None), ("one 1e6 cell", lambda M: M.__setitem__((5, 7), 1_000_000)),
                    ("one 1e7 cell", lambda M: M.__setitem__((5, 7), 10_000_000)),
                    ("ten 1e5 cells in a column", lambda M: M.__setitem__((slice(0, 10), 9), 100_000))):
    M = M0.copy(order="F"); edit(M)
    for spread in ("1", "0"):
        os.environ["BNMF_ZSSPREAD"] = spread
        try:
            e = Engine(M, N, prior="gamma", seed=3)
        except Exception as ex:
            print(f"{label:28s} spread {spread}: refused ({str(ex)[:60]})"); continue
        apply_hyperprior_params(e, "gamma", M, N)
        e.init(); e.run(200)
        t = time.perf_counter(); e.run(1000); dt = time.perf_counter() - t
        print(f"{label:28s} spread {spread}: {dt*1e3:.1f} us/iteration, sum M {M.sum()}")
        e.close()
