"""Diagnostic: k_zalloc_step (default for 25 <= N <= 100, stats mode) against the oracle on small shapes, then its launch time at
the config-4 / config-5 sizes beside the tile kernel (BNMF_ZSTEP=0).  Usage: python tools/zstep_check.py [quick]"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as O  # noqa: E402
from bayesnmf_amd import Engine  # noqa: E402
from bayesnmf_amd.setup import apply_hyperprior_params, synth_counts  # noqa: E402


def cmp(tag, a, b):
    a, b = np.ascontiguousarray(a, dtype=np.float64), np.ascontiguousarray(b, dtype=np.float64)
    if np.array_equal(a.view(np.uint64), b.view(np.uint64)):
        return True
    bad = np.argwhere(a != b)
    print(f"  {tag}: {len(bad)} of {a.size} differ; first {bad[:5].tolist()}  ref {a[tuple(bad[0])]} got {b[tuple(bad[0])]}  sums {a.sum()} {b.sum()}", flush=True)
    return False


ok = True
shapes = [(33, 17, 26, {}), (96, 300, 50, {}), (130, 12, 40, {}), (200, 9, 30, {}), (70, 45, 75, {}), (45, 23, 76, {}), (64, 700, 100, {}),
          (96, 120, 50, dict(learning_rank=True, temperature=np.ones(100)))]
for K, G, N, kw in shapes:
    rng = np.random.default_rng(K + N)
    M = rng.poisson(rng.gamma(0.5, 30.0, size=(K, G))).astype(np.int32)
    M[:, G // 2] = 0
    M[K // 3, :] = 0
    M[1, 1] = 2000                                          # fragments
    o = O.Oracle(M, N, prior="gamma", seed=9, save_Z=True, nthreads=8, **kw)
    e = Engine(M, N, prior="gamma", seed=9, **kw)
    apply_hyperprior_params(o, "gamma", M, N); apply_hyperprior_params(e, "gamma", M, N)
    ro, re = o.init(), e.init()
    good = cmp("init ZsumK", o.get("ZsumK"), e.get("ZsumK")) & cmp("init ZsumG", o.get("ZsumG"), e.get("ZsumG")) & cmp("init row", ro[:9], re[:9])
    for it in range(3):
        mo, me = o.run(1), e.run(1)
        good &= cmp(f"it{it} ZsumK", o.get("ZsumK"), e.get("ZsumK")) & cmp(f"it{it} ZsumG", o.get("ZsumG"), e.get("ZsumG"))
        good &= cmp(f"it{it} row", mo[:, :9], me[:, :9]) & cmp(f"it{it} P", o.get("P"), e.get("P")) & cmp(f"it{it} E", o.get("E"), e.get("E"))
        if not good:
            break
    print(f"K={K} G={G} N={N} {list(kw)}: {'OK' if good else 'MISMATCH'}", flush=True)
    ok &= good
    e.close(); o.close()
print("ALL OK" if ok else "FAILED", flush=True)
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    sys.exit(0 if ok else 1)


def timing(name, K, G, N, R, seed, iters, **kw):
    M, _, _ = synth_counts(K, G, R, seed)
    for zstep in ("1", "0"):
        os.environ["BNMF_ZSTEP"] = zstep
        t0 = time.perf_counter()
        e = Engine(M, N, prior="gamma", seed=1, window=2, **kw)
        apply_hyperprior_params(e, "gamma", M, N); e.init()
        tc = time.perf_counter() - t0
        e.run(3, metrics=False)
        t0 = time.perf_counter(); e.run(iters, metrics=False); dt = (time.perf_counter() - t0) / iters
        prof = e.profile(3)
        print(f"{name} zstep={zstep}: create+init {tc:.1f} s, {dt * 1e3:.3f} ms/iter, k_zalloc {prof['k_zalloc'] * 1e3:.1f} us, k_rank {prof.get('k_rank', 0) * 1e3:.1f} us", flush=True)
        e.close()
    os.environ.pop("BNMF_ZSTEP")


timing("config 4 (K=96, G=10000, N=50, rank learning)", 96, 10000, 50, 12, 20250222, 30, learning_rank=True, temperature=np.ones(8000))
timing("config 5 at G=10000 (K=1536, N=100)", 1536, 10000, 100, 30, 20250223, 4)
sys.exit(0 if ok else 1)
