"""CPU tests: the C-ABI library loads and exports every symbol include/bnmf.h declares (no compute
calls without a GPU), it refuses to run without a device (no CPU fallback), and the host-side mirror
of the R interface (convergence control, tempering schedule, hyper-prior defaults, model check)."""
import ctypes
import os
import re
import warnings

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _built():
    from bayesnmf_amd import engine
    if not os.path.exists(engine.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return engine


def test_header_symbols_exported():
    engine = _built()
    hdr = open(os.path.join(ROOT, "include", "bnmf.h")).read()
    decl = set(re.findall(r"\b(bnmf_[a-z0-9_]+)\s*\(", hdr))
    decl -= {"bnmf_handle"}
    lib = ctypes.CDLL(engine.LIB_PATH)
    missing = [s for s in sorted(decl) if not hasattr(lib, s)]
    assert not missing, f"libbnmf.so lacks {missing}"
    assert set(engine.ABI_SYMBOLS) <= decl
    assert engine.lib().bnmf_version() == 100
    # ... and nothing beyond it: every bnmf_* symbol the product library exports is declared in the header
    import shutil
    import subprocess
    if shutil.which("nm"):
        out = subprocess.check_output(["nm", "-D", "--defined-only", engine.LIB_PATH], text=True)
        exported = {ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("bnmf_")}
        assert exported <= decl, f"exported but not declared in include/bnmf.h: {sorted(exported - decl)}"


def test_no_cpu_fallback():
    """Without a visible GPU the product must fail loudly, never route through a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    engine = _built()
    with pytest.raises(engine.BnmfError) as ei:
        engine.Engine(np.ones((4, 3), dtype=np.int32), 2, prior="gamma")
    assert ei.value.code == -4   # BNMF_ENODEVICE


def test_product_never_imports_oracle():
    for root, _, files in os.walk(os.path.join(ROOT, "bayesnmf_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip")):
                txt = open(os.path.join(root, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "orc_" not in txt, f


def test_new_convergence_control_defaults():
    from bayesnmf_amd.convergence import new_convergence_control
    cc = new_convergence_control()   # R/convergence.R:16-45
    assert cc == dict(MAP_over=1000, MAP_every=100, tol=0.001, Ninarow_nochange=5, Ninarow_nobest=10,
                      miniters=1000, maxiters=5000, minA=0, metric="logposterior")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        cc = new_convergence_control(miniters=500, maxiters=400)
        assert cc["miniters"] == 0 and "miniters >= maxiters" in str(w[0].message)


def test_check_convergence_sequence():
    from bayesnmf_amd.convergence import new_convergence_control, check_convergence
    cc = new_convergence_control(MAP_over=10, MAP_every=10, miniters=0, maxiters=1000, Ninarow_nochange=3)
    temps = np.ones(2000)
    state = dict(iter=10, converged=False)
    msg = check_convergence(state, cc, -1000.0, temps)      # first call: prev = best = m + 1
    assert "no change" in msg and state["inarow_no_change"] == 1 and not state["converged"]
    for it, v in ((20, -1000.1), (30, -1000.05)):
        state["iter"] = it
        check_convergence(state, cc, v, temps)
    assert state["converged"] and state["why"] == "no change"
    # tempering blocks qualification
    state = dict(iter=10, converged=False)
    temps2 = np.concatenate([np.full(15, 0.5), np.ones(100)])
    for it in (10, 20):
        state["iter"] = it
        check_convergence(state, cc, -5.0, temps2)
    assert not state["converged"]


def test_temp_schedule_shape():
    from bayesnmf_amd.sampler import get_temp_sched_
    s = get_temp_sched_(5000, 1000)     # R/utils.R:307-332, n_temp = round(0.2 * 5000)
    assert len(s) == 5000 and s[0] == 0 and (np.diff(s) >= 0).all() and s[-1] == 1
    assert (s[:1000] < 1).all() and (s[1000:] == 1).all()
    s = get_temp_sched_(300, 100)       # longer than n_temp: sub-sampled in order
    assert len(s) == 300 and (np.diff(s) >= 0).all() and (s[100:] == 1).all()


def test_default_hyperprior_params():
    from bayesnmf_amd.setup import default_hyperprior_params
    M = np.full((96, 10), 42.0)
    g = default_hyperprior_params("gamma", M, 20)      # R/setup.R:167-181
    assert np.isclose(g["a_p"], 10 * np.sqrt(20)) and g["b_p"] == 10 and np.isclose(g["c_p"], 10 * np.sqrt(42)) and g["d_e"] == 10
    t = default_hyperprior_params("truncnormal", M, 20)  # :123-138
    assert t["m_p"] == 0 and np.isclose(t["s_p"], np.sqrt(42 / 20)) and t["a_e"] == 21 and np.isclose(t["b_e"], np.sqrt(20))
    e = default_hyperprior_params("exponential", M, 20)  # :148-157
    assert np.isclose(e["a_p"], 10 * np.sqrt(20)) and np.isclose(e["b_e"], 10 * np.sqrt(42))


class _OracleChain:
    """Test adapter: the CPU oracle behind the Engine method names, with a host-side sample window."""

    def __init__(self, M, N, **kw):
        import oracle as O
        self._o = O.Oracle(M, N, nthreads=4, **kw)
        self._hist = {}

    def set(self, name, v):
        self._o.set(name, v)

    def get(self, name):
        return self._o.get(name)

    def _rec(self):
        self._hist[self._o.iter] = {n: self._o.get(n) for n in ("P", "E", "A", "R", "Alpha_p", "Beta_p", "Alpha_e", "Beta_e")}

    def init(self):
        r = self._o.init(); self._rec(); return r

    def run(self, n, converged=False):
        rows = []
        for _ in range(n):
            rows.append(self._o.run(1, converged)[0]); self._rec()
        return np.array(rows)

    def window(self, name, last_n):
        it = self._o.iter
        return [self._hist[i][name] for i in range(it - last_n + 1, it + 1)]

    def close(self):
        self._o.close()


def test_sampler_mirror_end_to_end_on_oracle(tmp_path):
    """bayesNMF() control flow (blocks, MAP, convergence, files) exercised on CPU with the oracle
    standing in for the engine: fixed-rank Poisson-Gamma on example-like data recovers the signatures."""
    from bayesnmf_amd.sampler import bayesNMF
    from bayesnmf_amd.convergence import new_convergence_control
    from bayesnmf_amd.setup import synth_counts
    M, Pt, _ = synth_counts(24, 30, 2, 3, mean_total=1500)
    cc = new_convergence_control(MAP_over=40, MAP_every=20, miniters=60, maxiters=200)
    s = bayesNMF(M, 2, likelihood="poisson", prior="gamma", convergence_control=cc, output_dir=str(tmp_path / "out"),
                 periodic_save=False, save_all_samples=False, engine_factory=_OracleChain)
    assert s.state["iter"] <= 200 and len(s.state["sample_metrics"]) == s.state["iter"]
    assert list(s.state["sample_metrics"].columns) == ["iter", "RMSE", "KL", "loglikelihood", "logposterior", "n_params", "BIC", "rank", "temp"]
    assert {"P", "A", "E", "idx", "A_counts", "keep_sigs"} <= set(s.MAP)
    assert np.allclose(s.MAP["P"].sum(0), 1.0)
    P = s.MAP["P"] / np.linalg.norm(s.MAP["P"], axis=0)
    cos = (P.T @ (Pt / np.linalg.norm(Pt, axis=0))).max(0)
    assert (cos > 0.95).all()
    assert os.path.exists(os.path.join(s.specs["output_dir"], "log.txt")) and os.path.exists(os.path.join(s.specs["output_dir"], "sampler.pkl"))
    assert {"total", "per_iter"} <= set(s.time)
    s.close()


def test_model_check_errors(tmp_path):
    from bayesnmf_amd.sampler import bayesNMF_sampler
    M = np.ones((4, 3), dtype=np.int32)
    with pytest.raises(ValueError, match="gamma prior cannot be used in a MH-within-gibbs sampler"):
        bayesNMF_sampler(M, 2, prior="gamma", MH=True, output_dir=str(tmp_path / "a"), engine_factory=_OracleChain)
    with pytest.raises(ValueError, match="truncnormal prior can only be used in a MH-within-gibbs sampler"):
        bayesNMF_sampler(M, 2, prior="truncnormal", MH=False, output_dir=str(tmp_path / "b"), engine_factory=_OracleChain)
    with pytest.raises(ValueError, match="with `likelihood = 'normal'`"):
        bayesNMF_sampler(M, 2, likelihood="normal", prior="gamma", MH=False, output_dir=str(tmp_path / "c"), engine_factory=_OracleChain)


def test_BIC_sweep_concurrent_fixed_rank_chains(tmp_path):
    """rank_method = "BIC" (R/bayesNMF.R:66-126): one fixed-rank sampler per rank, run concurrently; the best BIC wins."""
    from bayesnmf_amd.sampler import bayesNMF
    from bayesnmf_amd.convergence import new_convergence_control
    from bayesnmf_amd.setup import synth_counts
    M, _, _ = synth_counts(24, 40, 2, 5, mean_total=2500)
    cc = new_convergence_control(MAP_over=30, MAP_every=15, miniters=45, maxiters=90)
    r = bayesNMF(M, range(1, 4), likelihood="poisson", prior="gamma", rank_method="BIC", convergence_control=cc,
                 output_dir=str(tmp_path / "bic"), periodic_save=False, save_all_samples=False, engine_factory=_OracleChain)
    assert set(r) == {"results", "best_rank", "sampler"} and list(r["results"]["rank"].sort_values()) == [1, 2, 3]
    assert r["best_rank"] == 2 and r["sampler"].dims["N"] == 2
    assert r["results"].iloc[0]["BIC"] == r["results"]["BIC"].min()


def _shim_functions():
    """name -> (number of SEXP parameters, body) of every `SEXP C_bnmf_*(...)` definition in r/bnmf_shim.c"""
    import re
    src = open(os.path.join(ROOT, "r", "bnmf_shim.c")).read()
    out = {}
    for m in re.finditer(r"^SEXP (C_bnmf_\w+)\(([^)]*)\)\s*\{", src, re.M):
        depth, i = 1, m.end()
        while depth:
            depth += {"{": 1, "}": -1}.get(src[i], 0)
            i += 1
        params = [p for p in m.group(2).split(",") if p.strip()]
        assert all(p.strip().startswith("SEXP ") for p in params), m.group(1)
        out[m.group(1)] = (len(params), src[m.end():i])
    return src, out


def test_r_shim_binds_every_entry_point_of_the_header():
    """Every chain-level entry point of include/bnmf.h has a C_bnmf_* binding in r/bnmf_shim.c that calls it.  (What the
    compiler and an executed call can check — argument counts and types, PROTECT balance, registration, the `.Call`s of
    r/bayesNMF_hip.R — is checked by compiling and RUNNING the shim: tests/test_rshim.py.)"""
    import re
    hdr = open(os.path.join(ROOT, "include", "bnmf.h")).read()
    declared = set(re.findall(r"^(?:int|const char\*|void)\s+(bnmf_\w+)\(", hdr, re.M))
    # not bound: unit probes of the parity tests, the profiler hook, library-level queries R has no use for
    not_bound = {"bnmf_ubench", "bnmf_test_math", "bnmf_test_sampler", "bnmf_test_philox", "bnmf_test_philox7", "bnmf_profile", "bnmf_kernel_name", "bnmf_version",
                 "bnmf_device_count", "bnmf_last_error", "bnmf_get_array_i32", "bnmf_debug_rank", "bnmf_debug_zsort", "bnmf_debug_set_timeout", "bnmf_trim",
                 "bnmf_probe_overlap", "bnmf_test_gate", "bnmf_test_devlock", "bnmf_get_stat"}
    src, fns = _shim_functions()
    for name in sorted(declared - not_bound):
        b = "C_" + name
        assert b in fns, f"{name} has no .Call binding in r/bnmf_shim.c"
        assert re.search(r"\b%s\(" % name, fns[b][1]) or name == "bnmf_destroy", f"{b} does not call {name}"
    assert "bnmf_last_error()" in src                                       # errors surface as Rf_error(bnmf_last_error())


# ---- who may use a device at the same time (api.hip DeviceGate, devlock_open): host-only, deterministic ----
def _gate_lib():
    import ctypes as C
    from bayesnmf_amd.engine import lib
    L = lib()
    L.bnmf_test_gate.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_long), C.POINTER(C.c_long)]
    L.bnmf_test_devlock.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    return L


def test_device_gate_lets_the_exclusive_caller_in_under_overlapping_sharers():
    """Six sharers keep the gate busy with overlapping calls (each 2 ms, staggered: the gate never falls empty by itself for seconds).
    A reader-preferring lock — glibc's std::shared_mutex, which rounds 3-4 used — would not admit the exclusive caller before they stop
    (6 x 400 calls x 2 ms = 0.8 s each); the gate admits no sharer while it waits, so it is in after the calls that were running."""
    import ctypes as C
    L = _gate_lib()
    waited, admitted = C.c_long(-1), C.c_long(-1)
    assert L.bnmf_test_gate(6, 400, 2000, C.byref(waited), C.byref(admitted)) == 0
    assert admitted.value == 0
    assert 0 <= waited.value < 100_000, waited.value            # the running calls end within ~2.2 ms; 100 ms allows for a loaded host


def test_device_lock_files(tmp_path, monkeypatch, capfd):
    """The two lock files of a device: created world-accessible under BNMF_LOCKDIR, an existing read-only file of somebody else is
    still good for flock (no O_CREAT on it), the lock order of bnmf_run works on them; a directory that cannot be written is said on
    stderr, once, and reported (bnmf_run then refuses rank learning instead of running it unprotected)."""
    import ctypes as C
    import stat
    L = _gate_lib()
    ok1, ok2 = C.c_int(-1), C.c_int(-1)
    monkeypatch.setenv("BNMF_LOCKDIR", str(tmp_path))
    assert L.bnmf_test_devlock(b"0000_e5_00_0", C.byref(ok1), C.byref(ok2)) == 0 and ok1.value == 1 and ok2.value == 1
    for ext in ("lock", "gate"):
        f = tmp_path / f"bnmf_dev_0000_e5_00_0.{ext}"
        assert f.exists() and stat.S_IMODE(f.stat().st_mode) == 0o666
        os.chmod(f, 0o444)                                       # as another user's file under umask 022 looks to us
    assert L.bnmf_test_devlock(b"0000_e5_00_0", C.byref(ok1), C.byref(ok2)) == 0 and ok1.value == 1 and ok2.value == 1
    capfd.readouterr()
    monkeypatch.setenv("BNMF_LOCKDIR", str(tmp_path / "no" / "such" / "dir"))
    assert L.bnmf_test_devlock(b"0000_e5_00_0", C.byref(ok1), C.byref(ok2)) == 0 and ok1.value == 0 and ok2.value == 0
    assert "cannot open the device lock files" in capfd.readouterr().err
    L.bnmf_test_devlock(b"0000_e5_00_0", C.byref(ok1), C.byref(ok2))
    assert "cannot open" not in capfd.readouterr().err           # said once per process
