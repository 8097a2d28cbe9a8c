"""GPU parity tests proper: the HIP engine (through the C ABI) against the CPU oracle on the
same seeded inputs.  Integers bit-exact; fp64 values bit-exact where the op sequences are
identical (they are, by the stream spec) with a stated fallback tolerance."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ULP_TOL = 0          # spec: identical op order + -ffp-contract=off on both sides => 0 ulp
METRIC_RTOL = 1e-12  # written tolerance for the fp64-accumulated metrics rows


def _hyper(chain, prior, M, N):
    from bayesnmf_amd.setup import apply_hyperprior_params
    apply_hyperprior_params(chain, prior, M, N)


def _pair(M, N, prior, seed=1, chain_id=0, save_Z=True):
    import oracle as O
    from bayesnmf_amd import Engine
    o = O.Oracle(M, N, prior=prior, seed=seed, chain_id=chain_id, save_Z=save_Z, nthreads=8)
    e = Engine(M, N, prior=prior, seed=seed, chain_id=chain_id, save_Z=save_Z)
    _hyper(o, prior, M, N)
    _hyper(e, prior, M, N)
    return o, e


def test_philox_kat_on_device():
    from bayesnmf_amd import engine as E
    assert E.test_philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert E.test_philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert E.test_philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


@pytest.mark.parametrize("fn", ["log", "exp", "lgamma", "digamma", "qnorm", "log_pnorm", "sqrt", "recip"])
def test_math_bitexact_vs_oracle(fn, oracle_lib):
    from bayesnmf_amd import engine as E
    rng = np.random.default_rng(7)
    if fn in ("log", "sqrt", "recip"):
        x = np.concatenate([10 ** rng.uniform(-300, 300, 20000), rng.uniform(0.5, 2, 20000), [1.0, 2.0, 1e-320]])
    elif fn == "exp":
        x = np.concatenate([rng.uniform(-745, 709, 30000), rng.uniform(-1, 1, 10000), [0.0, 709.7, -745.1]])
    elif fn in ("lgamma", "digamma"):
        x = np.concatenate([10 ** rng.uniform(-3, 7, 20000), rng.uniform(0.001, 20, 20000)])
    elif fn == "qnorm":
        x = np.concatenate([rng.uniform(0, 1, 30000), 10 ** rng.uniform(-300, -1, 10000)])
        x = x[(x > 0) & (x < 1)]
    else:
        x = rng.uniform(-38, 10, 40000)
    got = E.test_math(fn, x)
    if fn == "sqrt":
        ref = np.sqrt(x)
    elif fn == "recip":
        ref = 1.0 / x
    else:
        ref = oracle_lib.vec(fn, x)
    assert np.array_equal(got.view(np.uint64), ref.view(np.uint64)), \
        f"{fn}: {np.sum(got.view(np.uint64) != ref.view(np.uint64))} of {x.size} differ"


def test_samplers_bitexact_vs_oracle(oracle_lib):
    from bayesnmf_amd import engine as E
    O = oracle_lib
    rng = np.random.default_rng(3)
    n = 50000
    shape = 10 ** rng.uniform(-2, 4, n)
    rate = 10 ** rng.uniform(-2, 2, n)
    a = E.test_sampler("rgamma", shape, rate, var=2, it=5)
    b = O.rgamma(shape, rate, var=2, it=5)
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64))
    mu = rng.uniform(-20, 20, n)
    sd = 10 ** rng.uniform(-1, 1, n)
    a = E.test_sampler("rtnorm0", mu, sd, var=3, it=6)
    b = O.rtnorm0(mu, sd, var=3, it=6)
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64))
    c = 10 ** rng.uniform(-2, 3.5, n)
    tau = rng.uniform(-9, 50, n)
    xp = 10 ** rng.uniform(-3, 4, n)
    a = E.test_sampler("ralpha", c, tau, xp, var=5, it=7)
    b, _ = O.ralpha(c, tau, xp, var=5, it=7)
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64))


@pytest.mark.parametrize("prior", ["gamma", "exponential"])
def test_chain_bitexact_config1(prior):
    """Config 1 (K=96, G=100, N=5): 30 whole iterations, every integer and fp64 array."""
    from bayesnmf_amd.setup import synth_counts
    M, _, _ = synth_counts(96, 100, 5, 20250219)
    o, e = _pair(M, 5, prior)
    r0, r1 = o.init(), e.init()
    assert np.allclose(r0[:9], r1[:9], rtol=METRIC_RTOL, atol=0)
    names = ["P", "E", "ZsumK", "ZsumG", "Z"] + (["Alpha_p", "Beta_p", "Alpha_e", "Beta_e"] if prior == "gamma"
                                                  else ["Lambda_p", "Lambda_e"])
    for step in range(3):
        mo, me = o.run(10), e.run(10)
        for nm in names:
            a, b = o.get(nm), e.get(nm)
            if nm in ("ZsumK", "ZsumG", "Z"):
                assert np.array_equal(a.astype(np.int64), b.astype(np.int64)), f"{nm} differs at block {step}"
            else:
                assert np.array_equal(a.view(np.uint64), b.view(np.uint64)), \
                    f"{nm}: {np.sum(a != b)} of {a.size} values differ at block {step}"
        assert np.allclose(mo[:, :9], me[:, :9], rtol=METRIC_RTOL, atol=0)
        assert np.array_equal(mo[:, :9].view(np.uint64), me[:, :9].view(np.uint64)), "metrics rows not bit-identical"
    Z = e.get("Z")
    assert (Z.sum(1) == M).all()


def test_invariants_config2_full_size():
    """Config 2 size (K=96, G=2000, N=20): exact integer invariants of the allocation
     + oracle parity on one iteration."""
    from bayesnmf_amd.setup import synth_counts
    M, _, _ = synth_counts(96, 2000, 8, 20250220)
    o, e = _pair(M, 20, "gamma")
    o.init(); e.init()
    e.run(3); o.run(3)
    Z = e.get("Z")
    assert (Z >= 0).all()
    assert (Z.sum(1) == M).all()
    assert np.array_equal(e.get("ZsumK"), Z.sum(0))
    assert np.array_equal(e.get("ZsumG"), Z.sum(2))
    assert np.array_equal(e.get("ZsumK"), o.get("ZsumK").astype(np.int32))
    assert np.array_equal(e.get("P").view(np.uint64), o.get("P").view(np.uint64))


def test_headline_size_invariants_stats_mode():
    """Metric config (K=96, G=10000, N=20), stats mode: marginals must reproduce row/column sums."""
    from bayesnmf_amd import Engine
    from bayesnmf_amd.setup import synth_counts, apply_hyperprior_params
    M, _, _ = synth_counts(96, 10000, 8, 20250218)
    e = Engine(M, 20, prior="gamma", seed=1)
    apply_hyperprior_params(e, "gamma", M, 20)
    e.init()
    met = e.run(20)
    assert np.array_equal(e.get("ZsumK").sum(0), M.sum(0))
    assert np.array_equal(e.get("ZsumG").sum(1), M.sum(1))
    assert np.all(np.isfinite(met[:, :9]))
    assert met[-1, 1] < met[0, 1]   # RMSE goes down


def _temp_schedule(n):
    return np.concatenate([np.zeros(3), 10.0 ** np.linspace(-6, 0, 40), np.ones(max(0, n - 43))])


@pytest.mark.parametrize("method", ["SBFI", "BFI"])
def test_learned_rank_chain_bitexact(method):
    """sample_R / sample_An (R/sample_params.R:101-241): A, R, and everything downstream bit-exact
    against the oracle over a tempered run (config-4 model at a size the oracle finishes in seconds)."""
    import oracle as O
    from bayesnmf_amd import Engine
    from bayesnmf_amd.setup import synth_counts, apply_hyperprior_params
    M, _, _ = synth_counts(96, 120, 3, 20250221)
    N = 8
    temp = _temp_schedule(200)
    o = O.Oracle(M, N, prior="gamma", learning_rank=True, rank_method=method, seed=5, temperature=temp, save_Z=True, nthreads=8)
    e = Engine(M, N, prior="gamma", learning_rank=True, rank_method=method, seed=5, temperature=temp, save_Z=True)
    apply_hyperprior_params(o, "gamma", M, N)
    apply_hyperprior_params(e, "gamma", M, N)
    r0, r1 = o.init(), e.init()
    assert np.array_equal(o.get("A"), e.get("A")) and o.get("R")[0] == e.get("R")[0]
    assert np.array_equal(r0[:9].view(np.uint64), r1[:9].view(np.uint64))
    ranks = []
    for step in range(4):
        mo, me = o.run(15), e.run(15)
        assert np.array_equal(o.get("A"), e.get("A")), f"A differs at block {step}"
        assert o.get("R")[0] == e.get("R")[0]
        assert np.array_equal(o.get("Z").astype(np.int32), e.get("Z"))
        for nm in ("P", "E", "Alpha_e", "Beta_p"):
            assert np.array_equal(o.get(nm).view(np.uint64), e.get(nm).view(np.uint64)), nm
        assert np.array_equal(mo[:, :9].view(np.uint64), me[:, :9].view(np.uint64))
        ranks.append(me[-1, 7])
    Z, A = e.get("Z"), e.get("A")[0]
    assert (Z[:, A == 0, :] == 0).all()          # excluded factors receive no counts
    assert (Z.sum(1) == M).all() or A.sum() == 0
