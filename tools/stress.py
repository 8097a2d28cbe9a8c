"""Determinism stress: many short chains on tiny shapes (all kernels launch-bound, maximal stream overlap);
every repetition must reproduce the first one bit for bit."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesnmf_amd import Engine
from bayesnmf_amd.setup import apply_hyperprior_params
rng = np.random.default_rng(8)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
tot_bad = 0
for (K, G, N, prior, kw) in [(5, 3, 1, "gamma", {}), (70, 9, 2, "gamma", {}), (33, 17, 26, "gamma", {}), (96, 300, 8, "gamma", {}),
                             (40, 30, 4, "exponential", {}), (40, 30, 5, "gamma", dict(learning_rank=True)),
                             (40, 700, 4, "truncnormal", dict(MH=True)), (30, 40, 3, "truncnormal", dict(likelihood="normal")),
                             (96, 2000, 20, "gamma", {})]:
    M = rng.poisson(rng.gamma(0.5, 20.0, size=(K, G))).astype(np.int32)
    ref, bad = None, 0
    for r in range(reps):
        e = Engine(M, N, prior=prior, seed=4, **kw)
        apply_hyperprior_params(e, prior, M, N)
        e.init()
        m = e.run(40)
        cur = (m[:, :9].copy(), e.get("P"), e.get("E"))
        e.close()
        if ref is None: ref = cur
        elif not all(np.array_equal(a.view(np.uint64), b.view(np.uint64)) for a, b in zip(ref, cur)):
            bad += 1
            rows = [i for i in range(40) if not np.array_equal(ref[0][i].view(np.uint64), cur[0][i].view(np.uint64))]
            print("  mismatch rep", r, "first bad metric rows", rows[:5], "P equal", np.array_equal(ref[1], cur[1]), flush=True)
    print((K, G, N, prior, kw), "bad", bad, "of", reps - 1, flush=True)
    tot_bad += bad
print("TOTAL BAD", tot_bad)
