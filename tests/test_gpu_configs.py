"""GPU parity tests of the BASELINE.json configurations' real kernel paths (through the C ABI):
config 4's learned rank at N = 50 (general allocation kernel with excluded factors), full-size property
runs of configs 3, 4 and 5, the large-K fallback of the register kernel, record_sample against the oracle,
the vignette's initial-value checks and the ABI's error codes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _temp_schedule(n):
    return np.concatenate([np.zeros(3), 10.0 ** np.linspace(-6, 0, 40), np.ones(max(0, n - 43))])


def _mk(cls, M, N, prior, **kw):
    from bayesnmf_amd.setup import apply_hyperprior_params
    c = cls(M, N, prior=prior, **kw)
    apply_hyperprior_params(c, prior, M, N)
    return c


@pytest.mark.parametrize("half_blocks", ["1", "0"])
def test_config4_learned_rank_N50_bitexact(half_blocks, monkeypatch):
    """Config 4's model on its real kernel path: rank = 1:50 => N = 50 > 25, so the Z allocation runs on the
    general kernel k_zalloc with A containing zeros (R/sample_params.R:101-241, :253-265).  Bit-exact against
    the oracle at a reduced number of columns.  The rank sweep in both register forms: half a block of columns per
    compute wave (round 4, the default where its grid fits the device) and a whole block (BNMF_RANKHALF=0)."""
    import oracle as O
    from bayesnmf_amd import Engine
    from bayesnmf_amd.setup import synth_counts
    monkeypatch.setenv("BNMF_RANKHALF", half_blocks)
    M, _, _ = synth_counts(96, 72, 5, 20250230)
    N = 50
    temp = _temp_schedule(200)
    kw = dict(learning_rank=True, rank_method="SBFI", seed=13, temperature=temp, save_Z=True)
    o = _mk(O.Oracle, M, N, "gamma", nthreads=8, **kw)
    e = _mk(Engine, M, N, "gamma", **kw)
    r0, r1 = o.init(), e.init()
    assert np.array_equal(o.get("A"), e.get("A")) and o.get("R")[0] == e.get("R")[0]
    assert np.array_equal(r0[:9].view(np.uint64), r1[:9].view(np.uint64))
    saw_zero = False
    for step in range(4):
        mo, me = o.run(12), e.run(12)
        A = e.get("A")[0]
        saw_zero = saw_zero or (A == 0).any()
        assert np.array_equal(o.get("A"), e.get("A")), f"A differs at block {step}"
        assert o.get("R")[0] == e.get("R")[0]
        Z = e.get("Z")
        assert np.array_equal(o.get("Z").astype(np.int32), Z)
        assert (Z[:, A == 0, :] == 0).all()
        for nm in ("P", "E", "Alpha_e", "Beta_p"):
            assert np.array_equal(o.get(nm).view(np.uint64), e.get(nm).view(np.uint64)), nm
        assert np.array_equal(mo[:, :9].view(np.uint64), me[:, :9].view(np.uint64))
    assert saw_zero, "the run never excluded a factor: the zero-probability path was not exercised"


@pytest.mark.parametrize("K,G,N,lik,prior,MH", [(200, 30, 6, "poisson", "gamma", False), (130, 700, 5, "poisson", "truncnormal", True),
                                                 (70, 40, 4, "normal", "exponential", False)])
def test_rank_sweep_general_path_bitexact(K, G, N, lik, prior, MH):
    """The persistent rank sweep outside its register-resident fast path (K > 128: Mhat in global scratch), with the
    MH and Normal models (Normal log-likelihood in sample_An), and G spanning several 320-column segments."""
    import oracle as O
    from bayesnmf_amd import Engine
    rng = np.random.default_rng(K)
    M = rng.poisson(rng.gamma(1.0, 15.0, size=(K, G))).astype(np.int32)
    temp = _temp_schedule(100)
    kw = dict(likelihood=lik, MH=MH, learning_rank=True, seed=17, temperature=temp)
    o = _mk(O.Oracle, M, N, prior, nthreads=8, **kw)
    e = _mk(Engine, M, N, prior, **kw)
    o.init(); e.init()
    for step in range(3):
        mo, me = o.run(15), e.run(15)
        assert np.array_equal(o.get("A"), e.get("A")), step
        for nm in ("P", "E"):
            assert np.array_equal(o.get(nm).view(np.uint64), e.get(nm).view(np.uint64)), (nm, step)
        assert np.array_equal(mo[:, :9].view(np.uint64), me[:, :9].view(np.uint64)), step
    e.close()


@pytest.mark.parametrize("K,G,N", [(8, 13000, 4), (100, 12500, 3), (8, 17000, 3)])
def test_rank_sweep_wide_G_bitexact(K, G, N):
    """Wide problems: G = 13,000 / 12,500: more column blocks than the half-block and whole-block register grids hold on 256 CUs
    (K = 8: whole blocks in registers; K = 100: Mhat in global scratch); G = 17,000: 2,125 blocks, i.e. the decision wave gathers
    the block sums in two rounds of 2,048.  The sums of factor n+1 are published a step ahead of the decision for factor n (again
    after a flip)."""
    import oracle as O
    from bayesnmf_amd import Engine
    rng = np.random.default_rng(K + G)
    M = rng.poisson(rng.gamma(1.0, 6.0, size=(K, G))).astype(np.int32)
    kw = dict(learning_rank=True, seed=23, temperature=_temp_schedule(40))
    o = _mk(O.Oracle, M, N, "gamma", nthreads=8, **kw)
    e = _mk(Engine, M, N, "gamma", **kw)
    o.init(); e.init()
    flips = 0
    prev = o.get("A").copy()
    for step in range(12):
        n_it = 1 if step < 8 else 4                               # single iterations while the temperature is ~0 (A flips freely)
        mo, me = o.run(n_it), e.run(n_it)
        assert np.array_equal(o.get("A"), e.get("A")), step
        flips += int((o.get("A") != prev).sum()); prev = o.get("A").copy()
        for nm in ("P", "E"):
            assert np.array_equal(o.get(nm).view(np.uint64), e.get(nm).view(np.uint64)), (nm, step)
        assert np.array_equal(mo[:, :9].view(np.uint64), me[:, :9].view(np.uint64)), step
    assert flips > 0, "no factor ever flipped: the redo path was not exercised"
    e.close()


def test_config4_full_size_properties():
    """Config 4 at full size (K = 96, G = 10,000, N = 50, SBFI): sum_n Z = M for every cell, Z = 0 wherever
    A[n] = 0, marginals consistent."""
    from bayesnmf_amd import Engine
    from bayesnmf_amd.setup import synth_counts
    M, _, _ = synth_counts(96, 10000, 12, 20250222)
    N = 50
    A0 = np.ones((1, N)); A0[0, 5::3] = 0.0                       # start with a third of the factors excluded
    e = _mk(Engine, M, N, "gamma", learning_rank=True, rank_method="SBFI", seed=1, temperature=np.ones(100), save_Z=True)
    e.set("A", A0)
    e.init()
    met = e.run(3)
    A = e.get("A")[0]
    Z = e.get("Z")
    assert (Z >= 0).all()
    assert (Z[:, A == 0, :] == 0).all()
    if A.sum() > 0:
        assert (Z.sum(1) == M).all()
    assert np.array_equal(e.get("ZsumK"), Z.sum(0)) and np.array_equal(e.get("ZsumG"), Z.sum(2))
    assert np.all(np.isfinite(met[:, :9]))
    assert met[-1, 7] == A.sum()
    e.close()


def test_config3_N20_bitexact_reduced_G():
    """Config 3's model at its full rank N = 20 (K = 96), G reduced to two 512-column segments so that the
    oracle finishes in seconds: both phases (accept-all, true MH) bit-exact."""
    import oracle as O
    from bayesnmf_amd import Engine
    from bayesnmf_amd.setup import synth_counts
    M, _, _ = synth_counts(96, 600, 8, 20250223)
    N = 20
    o = _mk(O.Oracle, M, N, "truncnormal", MH=True, seed=4, nthreads=16)
    e = _mk(Engine, M, N, "truncnormal", MH=True, seed=4)
    o.init(); e.init()
    for conv in (False, True):
        mo, me = o.run(3, converged=conv), e.run(3, converged=conv)
        for nm in ("P", "E", "P_acceptance_rate", "E_acceptance_rate", "Mu_p", "Sigmasq_e"):
            a, b = o.get(nm), e.get(nm)
            assert np.array_equal(a.view(np.uint64), b.view(np.uint64)), f"{nm}: {np.sum(a != b)} differ (converged={conv})"
        assert np.array_equal(mo.view(np.uint64), me.view(np.uint64))


def test_config3_full_size_properties():
    """Config 3 at full size (Poisson-TruncNormal + MH, N = 20, K = 96, G = 5,000), both phases: state stays
    finite and non-negative, acceptance rates are 1 before convergence and in [0, 1] after, the fit improves."""
    from bayesnmf_amd import Engine
    from bayesnmf_amd.setup import synth_counts
    M, _, _ = synth_counts(96, 5000, 8, 20250221)
    e = _mk(Engine, M, 20, "truncnormal", MH=True, seed=1)
    r0 = e.init()
    met = e.run(30, converged=False)
    assert np.all(np.isfinite(met))
    assert (e.get("P_acceptance_rate") == 1.0).all() and (e.get("E_acceptance_rate") == 1.0).all()
    assert (met[:, 9] == 1.0).all() and (met[:, 10] == 1.0).all()
    assert met[-1, 1] < r0[1]                                      # RMSE below the prior draw's
    met2 = e.run(10, converged=True)
    P, E = e.get("P"), e.get("E")
    assert np.isfinite(P).all() and np.isfinite(E).all() and (P >= 0).all() and (E >= 0).all()
    for nm in ("P_acceptance_rate", "E_acceptance_rate"):
        a = e.get(nm)
        assert (a >= 0).all() and (a <= 1).all()
    assert np.all(np.isfinite(met2)) and (met2[:, 9] <= 1).all() and (met2[:, 9] > 0).all()
    e.close()


def test_config5_full_size_properties():
    """Config 5 at FULL size (K = 1,536, G = 50,000, N = 100): the row-chunked allocation kernel; the marginals
    of Z reproduce the row and column sums of M exactly, metrics finite."""
    from bayesnmf_amd import Engine
    from bayesnmf_amd.setup import synth_counts
    M, _, _ = synth_counts(1536, 50000, 30, 20250223)
    e = _mk(Engine, M, 100, "gamma", seed=1)
    e.init()
    met = e.run(2)
    zk, zg = e.get("ZsumK"), e.get("ZsumG")
    assert (zk >= 0).all() and (zg >= 0).all()
    assert np.array_equal(zk.sum(0, dtype=np.int64), M.sum(0, dtype=np.int64))
    assert np.array_equal(zg.sum(1, dtype=np.int64), M.sum(1, dtype=np.int64))
    assert np.all(np.isfinite(met[:, :9]))
    e.close()


@pytest.mark.parametrize("K,G,N", [(1536, 12, 5), (700, 9, 20)])
def test_large_K_small_N_falls_back_to_general_kernel(K, G, N):
    """N <= 24 normally takes the register kernel, whose per-wave LDS slab grows with K; when it does not fit
    (K = 1,536: SBS-1536 catalogues) create falls back to the general kernel instead of failing.  Bit-exact."""
    import oracle as O
    from bayesnmf_amd import Engine
    rng = np.random.default_rng(K + N)
    M = rng.poisson(rng.gamma(0.5, 12.0, size=(K, G))).astype(np.int32)
    o = _mk(O.Oracle, M, N, "gamma", seed=5, save_Z=True, nthreads=8)
    e = _mk(Engine, M, N, "gamma", seed=5, save_Z=True)
    o.init(); e.init()
    mo, me = o.run(4), e.run(4)
    assert np.array_equal(o.get("Z").astype(np.int32), e.get("Z"))
    assert np.array_equal(o.get("E").view(np.uint64), e.get("E").view(np.uint64))
    assert np.array_equal(mo[:, :9].view(np.uint64), me[:, :9].view(np.uint64))
    e.close()


@pytest.mark.parametrize("model", ["gamma", "truncnormal_mh", "normal"])
def test_record_sample_against_oracle(model):
    """record_sample (R/bayesNMF_sampler.R:651-672): every recorded array of every iteration in the device
    ring equals what the ORACLE held at that iteration (incl. iteration 1 and, for the Normal likelihood,
    sigmasq drawn in the same iteration), across a wrap-around of the ring."""
    import oracle as O
    from bayesnmf_amd import Engine
    from bayesnmf_amd.setup import synth_counts
    M, _, _ = synth_counts(96, 40, 3, 77)
    N = 4
    if model == "gamma":
        kw, prior, names = dict(), "gamma", ["P", "E", "A", "R", "Alpha_p", "Beta_p", "Alpha_e", "Beta_e"]
    elif model == "truncnormal_mh":
        kw, prior = dict(MH=True), "truncnormal"
        names = ["P", "E", "Mu_p", "Sigmasq_p", "Mu_e", "Sigmasq_e", "P_acceptance_rate", "E_acceptance_rate"]
    else:
        kw, prior, names = dict(likelihood="normal"), "exponential", ["P", "E", "Lambda_p", "Lambda_e", "sigmasq"]
    W = 5
    o = _mk(O.Oracle, M, N, prior, seed=21, nthreads=4, **kw)
    e = _mk(Engine, M, N, prior, seed=21, window=W, **kw)
    o.init(); e.init()
    hist = {1: {nm: o.get(nm).copy() for nm in names}}
    for nm in names:                                                # samples[[nm]][[1]]
        assert np.array_equal(e.window(nm, 1)[0].view(np.uint64), hist[1][nm].view(np.uint64)), (nm, 1)
    for it in range(2, 13):
        o.run(1); e.run(1)
        hist[it] = {nm: o.get(nm).copy() for nm in names}
        if it in (3, 7, 12):
            n = min(W, it)
            for nm in names:
                win = e.window(nm, n)
                for j, i2 in enumerate(range(it - n + 1, it + 1)):
                    assert np.array_equal(win[j].view(np.uint64), hist[i2][nm].view(np.uint64)), (nm, i2, it)
    e.close()


def test_Z_history_window():
    """samples$Z (R/bayesNMF_sampler.R:245-252): with save_Z and a window the last Z arrays are kept per sample."""
    import oracle as O
    from bayesnmf_amd import Engine
    from bayesnmf_amd.setup import synth_counts
    M, _, _ = synth_counts(96, 30, 3, 5)
    o = _mk(O.Oracle, M, 4, "gamma", seed=3, save_Z=True, nthreads=4)
    e = _mk(Engine, M, 4, "gamma", seed=3, save_Z=True, window=3)
    o.init(); e.init()
    hist = {1: o.get("Z").copy()}
    for it in range(2, 7):
        o.run(1); e.run(1); hist[it] = o.get("Z").copy()
    win = e.window("Z", 3)
    for j, it in enumerate((4, 5, 6)):
        assert np.array_equal(win[j], hist[it]), it
    e.close()


@pytest.mark.parametrize("prior", ["truncnormal", "exponential", "gamma"])
def test_vignette_initial_value_checks(prior, tmp_path):
    """The reference's own executable checks (vignettes/advanced.qmd:181-185, :245-249, :315-319):
    samples$P[[1]] == init_params$P, samples$<prior param>[[1]] == init_prior_params$<...>, and the supplied
    hyper-prior scalar is kept — through the bayesNMF() mirror with rank = 1:10 as in the vignette."""
    import os
    from bayesnmf_amd.sampler import bayesNMF
    from bayesnmf_amd.convergence import new_convergence_control
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_example_data.npz"))
    M = d["M"]
    K, G = M.shape
    N = 10
    init_params = dict(R=[N], A=np.ones((1, N)), P=np.ones((K, N)), E=np.ones((N, G)))
    if prior == "truncnormal":
        ipp = dict(Mu_p=np.zeros((K, N)), Sigmasq_p=np.ones((K, N)), Mu_e=np.zeros((N, G)), Sigmasq_e=np.ones((N, G)))
        hp = dict(m_p=0.0, s_p=np.sqrt(M.mean() / N), a_p=N + 1.0, b_p=np.sqrt(N), m_e=0.0, s_e=np.sqrt(M.mean() / N),
                  a_e=N + 1.0, b_e=np.sqrt(N))
        key, hkey = "Mu_p", "m_p"
    elif prior == "exponential":
        ipp = dict(Lambda_p=np.ones((K, N)), Lambda_e=np.ones((N, G)))
        hp = dict(a_p=10 * np.sqrt(N), b_p=10 * np.sqrt(M.mean()), a_e=10 * np.sqrt(N), b_e=10 * np.sqrt(M.mean()))
        key, hkey = "Lambda_p", "a_p"
    else:
        ipp = dict(Alpha_p=np.ones((K, N)), Beta_p=np.ones((K, N)), Alpha_e=np.ones((N, G)), Beta_e=np.ones((N, G)))
        hp = dict(a_p=10 * np.sqrt(N), b_p=10.0, c_p=10 * np.sqrt(M.mean()), d_p=10.0, a_e=10 * np.sqrt(N), b_e=10.0,
                  c_e=10 * np.sqrt(M.mean()), d_e=10.0)
        key, hkey = "Alpha_p", "a_p"
    cc = new_convergence_control(MAP_over=20, MAP_every=10, maxiters=40, tol=0.01, miniters=0)   # "short", shortened further
    s = bayesNMF(M, range(1, N + 1), prior=prior, hyperprior_params=hp, init_prior_params=ipp, init_params=init_params,
                 output_dir=str(tmp_path / "o"), save_all_samples=True, overwrite=True, convergence_control=cc,
                 periodic_save=False, post_warmup=10)
    smp = s.samples
    assert np.array_equal(smp["P"][0], init_params["P"])
    assert np.array_equal(smp[key][0], ipp[key])
    assert s.hyperprior_params[hkey] == hp[hkey]
    assert np.array_equal(smp["A"][0], init_params["A"]) and smp["R"][0][0] == N
    s.close()


def test_abi_error_codes():
    """Error behaviour of the boundary: negative codes + bnmf_last_error(), nothing thrown across the ABI."""
    from bayesnmf_amd import Engine
    from bayesnmf_amd.engine import BnmfError
    from bayesnmf_amd.setup import apply_hyperprior_params
    M = np.random.default_rng(0).poisson(5.0, size=(12, 9)).astype(np.int32)
    with pytest.raises(BnmfError) as ei:
        Engine(M, 3, prior="gamma", MH=True)
    assert ei.value.code == -6                                      # BNMF_EMODEL (check_model)
    with pytest.raises(BnmfError) as ei:
        Engine(M, 3, prior="truncnormal", MH=False)
    assert ei.value.code == -6
    with pytest.raises(BnmfError) as ei:
        Engine(-M - 1, 3, prior="gamma")
    assert ei.value.code == -1                                      # BNMF_EINVAL: negative counts
    e = Engine(M, 3, prior="gamma", window=4)
    with pytest.raises(BnmfError) as ei:
        e.run(1)
    assert ei.value.code == -7                                      # BNMF_ESTATE: run before init
    with pytest.raises(BnmfError) as ei:
        e.init()
    assert ei.value.code == -3                                      # BNMF_EUNSET: hyper-prior arrays missing
    with pytest.raises(BnmfError) as ei:
        e.set("P", np.ones((5, 5)))
    assert ei.value.code == -2                                      # BNMF_ESIZE
    apply_hyperprior_params(e, "gamma", M, 3)
    e.init()
    with pytest.raises(BnmfError) as ei:
        e.get("Z")
    assert ei.value.code == -3                                      # Z not materialised (save_Z = 0)
    with pytest.raises(BnmfError) as ei:
        e.window("P", 3)
    assert ei.value.code == -2                                      # only one sample recorded so far
    with pytest.raises(BnmfError) as ei:
        e.window("sigmasq", 1)
    assert ei.value.code == -3                                      # not recorded for this model
    e.run(2)
    assert len(e.window("P", 3)) == 3
    e.close()


def test_randomised_parity_sweep():
    """tools/fuzz_parity.py: 60 random (model, K, G, N, rank learning, window) cases, engine against the oracle, bit-exact
    (NaN = NaN); tiny shapes, where the asynchronous side-stream work is most likely to be caught out of order."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FUZZ_N="60", FUZZ_SEED="77")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def _solo_metrics(M, rank, cc, out, **kw):
    from bayesnmf_amd.sampler import bayesNMF
    s = bayesNMF(M, rank, convergence_control=cc, output_dir=out, periodic_save=False, save_all_samples=False, **kw)
    sm = s.state["sample_metrics"].to_numpy().copy()
    s.close()
    return sm


def test_bic_sweep_runs_concurrent_handles_on_one_device(tmp_path):
    """bayesNMF(rank_method = "BIC") (R/bayesNMF.R:66-126): one fixed-rank chain per rank, here as concurrent handles (one
    host thread each) on ONE device.  Every chain is bit-identical to the same chain run alone."""
    from bayesnmf_amd.sampler import bayesNMF
    from bayesnmf_amd.convergence import new_convergence_control
    from bayesnmf_amd.setup import synth_counts
    M, _, _ = synth_counts(96, 1500, 3, 33)
    cc = new_convergence_control(MAP_over=40, MAP_every=20, miniters=40, maxiters=120)
    res = bayesNMF(M, [2, 3, 4, 5], rank_method="BIC", prior="gamma", convergence_control=cc, output_dir=str(tmp_path / "bic"),
                   periodic_save=False, save_all_samples=False, seed=9, devices=[0])
    assert sorted(res["results"]["rank"].tolist()) == [2, 3, 4, 5] and res["best_rank"] in (2, 3, 4, 5)
    import glob
    import pickle
    for k in (2, 3, 4, 5):
        solo = _solo_metrics(M, k, cc, str(tmp_path / f"solo{k}"), prior="gamma", seed=9)
        with open(glob.glob(str(tmp_path / "bic" / f"rank_{k}" / "sampler.pkl"))[0], "rb") as f:
            conc = pickle.load(f)["state"]["sample_metrics"].to_numpy()
        assert conc.shape == solo.shape and np.array_equal(np.nan_to_num(conc), np.nan_to_num(solo)), k


def test_run_chains_with_rank_learning_chains_sharing_a_device(tmp_path):
    """run_chains with n_chains = 3 on one device where every chain learns the rank (G >= 8,000: each persistent rank sweep
    wants a workgroup on most CUs, so two of them may not be co-resident — their calls take turns inside the library).
    No time-out, every chain bit-identical to its solo run."""
    from bayesnmf_amd.multichain import run_chains
    from bayesnmf_amd.sampler import bayesNMF_sampler
    from bayesnmf_amd.convergence import new_convergence_control
    from bayesnmf_amd.setup import synth_counts
    M, _, _ = synth_counts(96, 8200, 4, 41)
    cc = new_convergence_control(MAP_over=20, MAP_every=10, miniters=20, maxiters=50)
    kw = dict(prior="gamma", convergence_control=cc, periodic_save=False, save_all_samples=False, seed=4, prop_temp=0.4)
    chains = run_chains(M, range(1, 9), n_chains=3, devices=[0], output_dir=str(tmp_path / "mc"), **kw)
    for c, s in enumerate(chains):
        solo = bayesNMF_sampler(M, range(1, 9), chain_id=c, device=0, output_dir=str(tmp_path / f"solo{c}"), **kw)
        solo.run_gibbs_sampler()
        a, b = s.state["sample_metrics"].to_numpy(), solo.state["sample_metrics"].to_numpy()
        assert a.shape == b.shape and np.array_equal(np.nan_to_num(a), np.nan_to_num(b)), c
        assert np.array_equal(s.params["A"], solo.params["A"])
        solo.close(); s.close()


def test_four_different_chains_at_once_match_their_solo_runs():
    """Four handles of four different sweep types (gated fixed rank, rank learning, MH, exponential prior with K > 128) driven
    from four host threads on one device: same bits as each chain alone (tools/concurrent_check.py in the suite)."""
    import threading
    from bayesnmf_amd import Engine
    from bayesnmf_amd.setup import synth_counts, apply_hyperprior_params
    cases = [dict(K=96, G=8000, N=20, kw=dict(prior="gamma"), window=30),
             dict(K=96, G=900, N=8, kw=dict(prior="gamma", learning_rank=True, temperature=np.linspace(0.2, 1, 40)), window=0),
             dict(K=96, G=1200, N=6, kw=dict(prior="truncnormal", MH=True), window=10),
             dict(K=200, G=700, N=24, kw=dict(prior="exponential"), window=5)]

    def make(c, cid):
        M, _, _ = synth_counts(c["K"], c["G"], 4, 77 + cid)
        e = Engine(M, c["N"], seed=5, chain_id=cid, window=c["window"], **c["kw"])
        apply_hyperprior_params(e, c["kw"]["prior"], M, c["N"])
        e.init()
        return e
    n = 60
    alone = []
    for cid, c in enumerate(cases):
        e = make(c, cid)
        m = np.concatenate([e.run(n // 2, converged=True), e.run(n - n // 2, converged=True)])
        alone.append((m, e.get("P").copy(), e.get("E").copy()))
        e.close()
    es = [make(c, cid) for cid, c in enumerate(cases)]
    out, err = [None] * len(es), [None] * len(es)

    def work(i):
        try:
            out[i] = np.concatenate([es[i].run(n // 2, converged=True), es[i].run(n - n // 2, converged=True)])
        except BaseException as ex:  # noqa: BLE001
            err[i] = ex
    ths = [threading.Thread(target=work, args=(i,)) for i in range(len(es))]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert all(x is None for x in err), err
    for i, e in enumerate(es):
        assert np.array_equal(out[i][:, :9].view(np.uint64), alone[i][0][:, :9].view(np.uint64)), i
        assert np.array_equal(e.get("P").view(np.uint64), alone[i][1].view(np.uint64)), i
        assert np.array_equal(e.get("E").view(np.uint64), alone[i][2].view(np.uint64)), i
        e.close()


# repetition is weak evidence for a race and costs GPU time on every run of the suite — what guards the orderings are the
# deterministic hooks (test_every_side_stream_kernel_held_back: every side- / main-stream kernel held back, test_probe_*, the gate's host
# test).  The loops below run a few repetitions by default; BNMF_SOAK=1 restores the long ones (tools/soak_r5.sh runs them longer still:
# profiles/r05_soak.txt).
def _reps(default, soak):
    import os
    return str(soak if os.environ.get("BNMF_SOAK") == "1" else default)


def test_six_chains_at_once_ten_times_over():
    """tools/concurrent_check.py: six chains of six sweep types (MH before and after convergence and the Normal likelihood among them) from six host
    threads on one device, ten times with fresh handles: every run the bits of the chain alone, metric rows included.  (Round 4: with the MH /
    Normal hyper sweep on the main stream the reduce of the iteration before had lost its ordering against the next writers of its slot; this
    loop showed it in three of four repetitions.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "concurrent_check.py"), _reps(3, 10)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_six_full_size_chains_at_once():
    """The same with full-size chains (CONC_BIG=1: two gated fixed-rank chains at G = 10,000, config 3 twice, config 4 — whose persistent
    rank sweep wants every CU —, N = 100): before rank-learning calls took the device's lock exclusively, a gate-waiting allocation workgroup,
    the rank sweep's resident workgroups and the side-stream kernel between them waited for each other until a time-out poisoned a handle."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "concurrent_check.py"), _reps(1, 3)], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, CONC_BIG="1"))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_chains_mixing_every_entry_point_at_once():
    """tools/api_mix_check.py: five chains from five host threads on one device, each mixing bnmf_map (with bounds), bnmf_window, bnmf_get_array,
    a bnmf_set_array of P mid-chain and bnmf_assign into its run blocks: the bits of the same call sequence made alone, five times over."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "api_mix_check.py"), _reps(2, 5)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_two_processes_share_the_device():
    """tools/two_process_check.sh: two PROCESSES at once on the device, six full-size chains each, a rank-learning chain in both: the device's lock
    file (/tmp/bnmf_dev_<PCI bus id>.lock: exclusive for a rank-learning call, shared for every other) keeps the two persistent rank sweeps — and a
    rank sweep and the other process's gated chains — apart; without it (BNMF_DEVLOCK=0) both processes end in the rank sweep's time-out."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["bash", os.path.join(root, "tools", "two_process_check.sh"), _reps(1, 2)], capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_recreated_handle_does_not_wait_for_remapped_memory():
    """the first stream synchronisation after a bnmf_create that followed a bnmf_destroy waited 8-13 ms (27 ms with
    round 5's Mhat buffers) — freed device memory made again is mapped lazily, at the first submission that touches it
    (tools/recreate.py).  A destroyed handle's blocks now go to a per-device pool and the next handle of the same shape takes them from
    there: the calls between bnmf_create and the first iteration of a re-created handle take well under a millisecond each again.
    (Bound: 5 ms for the eight scalar bnmf_set_array calls together; they took 27 ms.)  bnmf_trim gives the pool back."""
    import time
    from bayesnmf_amd import Engine
    from bayesnmf_amd.engine import trim
    from bayesnmf_amd.setup import synth_counts, apply_hyperprior_params
    M, _, _ = synth_counts(96, 10000, 8, 20250218)
    waits = []
    for life in range(3):
        e = Engine(M, 20, prior="gamma", seed=1, window=50)
        t0 = time.perf_counter(); apply_hyperprior_params(e, "gamma", M, 20); waits.append(1e3 * (time.perf_counter() - t0))
        e.init(); e.run(5)
        e.close()
    assert max(waits[1:]) < 5.0, waits
    assert trim(0) > 20e6                       # the pooled blocks (Mhat alone is 23 MB) and the cached rings go back to the device
    assert trim(0) == 0
