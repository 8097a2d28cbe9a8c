"""Randomised bit-parity sweep: random shapes and models, engine against the oracle (diagnostics; the fixed cases are in tests/)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle as O
from bayesnmf_amd import Engine
from bayesnmf_amd.setup import apply_hyperprior_params
def same_bits(a, b):
    # bit-exact, except that NaN (degenerate all-zero data: log 0 - log 0) is NaN whatever its sign / payload
    a_, b_ = np.array(a, dtype=np.float64, copy=True), np.array(b, dtype=np.float64, copy=True)
    if a_.shape != b_.shape: return False
    nn = np.isnan(a_) & np.isnan(b_)
    a_[nn] = 0.0; b_[nn] = 0.0
    return np.array_equal(a_.view(np.uint64), b_.view(np.uint64))
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "1")))
rng_big = np.random.default_rng(7919 + int(os.environ.get("FUZZ_SEED", "1")))    # (its own stream: the cases of a seed stay what they were)
n_cases = int(os.environ.get("FUZZ_N", "60"))
bad = 0
t0 = time.time()
for case in range(n_cases):
    model = rng.choice(["gamma", "exponential", "tn_mh", "exp_mh", "normal_tn", "normal_exp"])
    K = int(rng.choice([3, 12, 31, 64, 65, 96, 97, 128, 130, 200]))
    G = int(rng.integers(1, int(os.environ.get("FUZZ_GMAX", "90"))))
    N = int(rng.choice([1, 2, 5, 9, 17, 21, 25, 26, 33, 50, 70]))
    lr = bool(rng.random() < 0.4) and N > 1
    if os.environ.get("FUZZ_LR_ONLY"):                     # the persistent rank sweep only (register forms: K <= 96)
        lr = True; N = max(N, 2); K = int(rng.choice([3, 12, 31, 64, 65, 96]))
    window = int(rng.choice([0, 3]))
    M = rng.poisson(rng.gamma(0.7, 10.0, size=(K, G))).astype(np.int32)
    if rng.random() < 0.3: M[:, rng.integers(0, G)] = 0
    if rng.random() < 0.3: M[rng.integers(0, K), :] = 0
    if rng_big.random() < 0.25:                           # a few large cells: the sorted schedule exports their fragments to other blocks
        for _ in range(int(rng_big.integers(1, 5))):
            M[rng_big.integers(0, K), rng_big.integers(0, G)] = int(rng_big.integers(8_193, 150_000))
    kw = dict(seed=int(rng.integers(1, 1000)), learning_rank=lr)
    save_Z = bool(rng.random() < 0.4) and model in ("gamma", "exponential")      # full mode of the Gibbs sweep: Z itself is compared
    if save_Z: kw["save_Z"] = True
    only = os.environ.get("FUZZ_ONLY")
    if only is not None and case != int(only): continue
    if lr: kw["temperature"] = np.linspace(0.1, 1.0, 12)
    if model in ("gamma", "exponential"): kw.update(prior=model)
    elif model == "tn_mh": kw.update(prior="truncnormal", MH=True)
    elif model == "exp_mh": kw.update(prior="exponential", MH=True)
    elif model == "normal_tn": kw.update(prior="truncnormal", likelihood="normal")
    else: kw.update(prior="exponential", likelihood="normal")
    try:
        o = O.Oracle(M, N, nthreads=4, **kw)
        e = Engine(M, N, window=window, **kw)
        apply_hyperprior_params(o, kw["prior"], M, N); apply_hyperprior_params(e, kw["prior"], M, N)
        r0, r1 = o.init(), e.init()
        ok = same_bits(r0[:9], r1[:9])
        for conv in (False, True):
            mo, me = o.run(7, converged=conv), e.run(7, converged=conv)
            same = same_bits(mo[:, :9], me[:, :9])
            if not same and os.environ.get("FUZZ_ONLY") is not None:
                dd = mo[:, :9].copy().view(np.uint64) != me[:, :9].copy().view(np.uint64)
                print("conv", conv, "rows", np.where(dd.any(1))[0], "cols", np.where(dd.any(0))[0]); i = np.where(dd.any(1))[0][0]; print(mo[i, :9]); print(me[i, :9])
            ok = ok and same
            for nm in ("P", "E", "A") + (("ZsumK", "ZsumG") if model in ("gamma", "exponential") else ()) + (("Z",) if save_Z else ()):
                ok = ok and same_bits(o.get(nm), e.get(nm))
        e.close()
    except Exception as ex:
        ok = False; print("EXC", ex)
    if not ok:
        bad += 1
        print("MISMATCH", case, model, K, G, N, lr, window, kw["seed"], "save_Z" if save_Z else "", flush=True)
print("cases %d, mismatches %d, %.0f s" % (n_cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
