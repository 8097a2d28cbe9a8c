"""Diagnostic: k_zalloc_sort against k_zalloc_reg (BNMF_ZSORT=0) on the same chains: ZsumK, ZsumG, metric rows, P, E."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesnmf_amd import Engine  # noqa: E402
from bayesnmf_amd.setup import apply_hyperprior_params, synth_counts  # noqa: E402


def mk(M, N, zs, **kw):
    os.environ["BNMF_ZSORT"] = str(zs)
    os.environ["BNMF_ZSPK"] = os.environ.get("ZSPK", "1")
    e = Engine(M, N, prior="gamma", seed=5, **kw)
    apply_hyperprior_params(e, "gamma", M, N)
    return e


def cmp(tag, a, b):
    if np.array_equal(a, b):
        return True
    bad = np.argwhere(a != b)
    print(f"  {tag}: {len(bad)} of {a.size} differ; first {bad[:5].tolist()}  ref {a[tuple(bad[0])]} got {b[tuple(bad[0])]}  sums {a.sum()} {b.sum()}")
    return False


shapes = [(96, 64, 5, {}), (200, 30, 6, {}), (96, 500, 20, {}), (50, 300, 12, {}), (7, 3, 2, {}), (96, 2000, 20, {}),
          (200, 30, 6, dict(learning_rank=True, temperature=np.ones(100)))]
ok = True
for K, G, N, kw in shapes:
    rng = np.random.default_rng(K + G)
    M = rng.poisson(rng.gamma(1.0, 15.0, size=(K, G))).astype(np.int32) if K != 96 else synth_counts(K, G, 4, 3)[0]
    e0, e1 = mk(M, N, 0, **kw), mk(M, N, 1, **kw)
    r0, r1 = e0.init(), e1.init()
    print(f"K={K} G={G} N={N} {list(kw)}")
    good = cmp("init ZsumK", e0.get("ZsumK"), e1.get("ZsumK")) & cmp("init ZsumG", e0.get("ZsumG"), e1.get("ZsumG")) & cmp("init row", r0[:9], r1[:9])
    for it in range(3):
        m0, m1 = e0.run(1), e1.run(1)
        good &= cmp(f"it{it} ZsumK", e0.get("ZsumK"), e1.get("ZsumK")) & cmp(f"it{it} ZsumG", e0.get("ZsumG"), e1.get("ZsumG"))
        good &= cmp(f"it{it} row", m0[:, :9], m1[:, :9]) & cmp(f"it{it} P", e0.get("P"), e1.get("P")) & cmp(f"it{it} E", e0.get("E"), e1.get("E"))
        if not good:
            break
    print("  OK" if good else "  MISMATCH")
    ok &= good
    e0.close(); e1.close()
print("ALL OK" if ok else "FAILED")
