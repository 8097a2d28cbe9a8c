"""bnmf_map (on-device MAP window statistics) against a numpy restatement of get_MAP_
(R/utils.R:194-288), get_mode and renormalize (R/helpers.R:35-79) evaluated on the same window of samples."""
from collections import Counter

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-12      # stated tolerance: fp64 sums in a different association order than numpy's


def _np_map(e, n, ci, M):
    A = e.window("A", n); P = e.window("P", n); E = e.window("E", n)
    keys = ["".join(str(int(v)) for v in a.ravel()) for a in A]
    tab = sorted(Counter(keys).items(), key=lambda kv: (-kv[1], kv[0]))      # sort(table(.), decreasing = TRUE)
    mode = tab[0][0]
    idx = [i for i, k in enumerate(keys) if k == mode]
    Pn = np.stack([P[i] / P[i].sum(0)[None, :] for i in idx], axis=2)
    En = np.stack([E[i] * P[i].sum(0)[:, None] for i in idx], axis=2)
    Am = np.array([float(c) for c in mode])
    out = dict(P=Pn.mean(2), E=En.mean(2), A=Am, idx=idx, tab=tab)
    if ci:
        pr = [0.5 - ci / 2, 0.5 + ci / 2]
        out.update(P_lower=np.quantile(Pn, pr[0], axis=2), P_upper=np.quantile(Pn, pr[1], axis=2),
                   E_lower=np.quantile(En, pr[0], axis=2), E_upper=np.quantile(En, pr[1], axis=2))
    Mh = (out["P"] * Am[None, :]) @ out["E"]
    out["rmse"] = np.sqrt(((Mh - M) ** 2).mean())
    Mt = np.maximum(M, 1e-6)
    out["kl"] = (Mt * np.log(Mt / np.maximum(Mh, 1e-6))).sum()
    return out


def _check(e, n, ci, M):
    ref = _np_map(e, n, ci, M)
    got = e.map(n, ci)
    assert got["n_used"] == len(ref["idx"])
    assert np.array_equal(np.where(got["used"])[0], ref["idx"])
    assert np.array_equal(got["A"].ravel(), ref["A"])
    assert got["top_counts"] == [c for _, c in ref["tab"][:5]]
    assert ["".join(str(int(v)) for v in r) for r in got["top_A"]] == [k for k, _ in ref["tab"][:5]]
    names = ["P", "E"] + (["P_lower", "P_upper", "E_lower", "E_upper"] if ci else [])
    for nm in names:
        assert np.allclose(got[nm], ref[nm], rtol=RTOL, atol=1e-300), (nm, np.abs(got[nm] / ref[nm] - 1).max())
    assert np.isclose(got["rmse"], ref["rmse"], rtol=1e-11) and np.isclose(got["kl"], ref["kl"], rtol=1e-10)


@pytest.mark.parametrize("n,ci", [(200, 0.95), (37, 0.95), (5, 0.95), (2, 0.95), (1, 0.95), (200, 0.5), (200, None)])
def test_map_fixed_rank(n, ci):
    from bayesnmf_amd import Engine
    from bayesnmf_amd.setup import synth_counts, apply_hyperprior_params
    M, _, _ = synth_counts(96, 150, 4, 5)
    e = Engine(M, 5, prior="gamma", seed=3, window=200)
    apply_hyperprior_params(e, "gamma", M, 5)
    e.init()
    e.run(260)                                       # the ring has wrapped
    _check(e, n, ci, M)
    e.close()


def test_map_window_above_2048_samples():
    """More than 2,048 samples in the window: the credible bounds then come from the per-lane running sets of the kt smallest / largest
    values (k_map_stats; up to 2,048 samples a wave sorts an element's samples: k_map_quant, round 4) — same values either way."""
    from bayesnmf_amd import Engine
    from bayesnmf_amd.setup import synth_counts, apply_hyperprior_params
    M, _, _ = synth_counts(96, 40, 3, 6)
    e = Engine(M, 3, prior="gamma", seed=4, window=2100)
    apply_hyperprior_params(e, "gamma", M, 3)
    e.init()
    e.run(2150, metrics=False)
    _check(e, 2100, 0.95, M)
    _check(e, 2048, 0.95, M)
    e.close()


def test_map_learned_rank_mode_of_A():
    """Several A patterns in the window: only the samples at the mode enter the means."""
    from bayesnmf_amd import Engine
    from bayesnmf_amd.setup import synth_counts, apply_hyperprior_params
    M, _, _ = synth_counts(96, 64, 3, 9)
    N = 8
    temp = np.concatenate([np.zeros(3), 10.0 ** np.linspace(-6, -1, 120), np.ones(100)])
    e = Engine(M, N, prior="gamma", learning_rank=True, seed=5, temperature=temp, window=100)
    apply_hyperprior_params(e, "gamma", M, N)
    e.init()
    e.run(110)                                       # window = iterations 12..111 of the tempered phase: A still moves
    pats = {tuple(a.ravel()) for a in e.window("A", 100)}
    assert len(pats) > 1, "test needs more than one A pattern in the window"
    _check(e, 100, 0.95, M)
    _check(e, 40, 0.9, M)
    e.close()


def test_map_metric_config_size_matches_host_path():
    """K = 96, G = 10,000, N = 20 with MAP_over = 1000 samples (the reference's default check): finite, normalised,
    and equal to the host evaluation on a strided subset of elements."""
    from bayesnmf_amd import Engine
    from bayesnmf_amd.setup import synth_counts, apply_hyperprior_params
    M, _, _ = synth_counts(96, 10000, 8, 20250218)
    e = Engine(M, 20, prior="gamma", seed=1, window=1000)
    apply_hyperprior_params(e, "gamma", M, 20)
    e.init()
    e.run(1100, metrics=False)
    r = e.map(1000, 0.95)
    assert r["n_used"] == 1000
    assert np.allclose(r["P"].sum(0), 1.0, rtol=1e-12)
    assert (r["E_lower"] <= r["E"]).all() and (r["E"] <= r["E_upper"]).all()
    assert (r["P_lower"] <= r["P_upper"]).all()
    P = np.stack(e.window("P", 1000), axis=2)
    cs = P.sum(0)                                                  # N x samples
    Pn = P / cs[None, :, :]
    assert np.allclose(r["P"], Pn.mean(2), rtol=RTOL)
    assert np.allclose(r["P_lower"], np.quantile(Pn, 0.025, axis=2), rtol=RTOL)
    assert np.allclose(r["P_upper"], np.quantile(Pn, 0.975, axis=2), rtol=RTOL)
    e.close()


def test_bayesNMF_uses_device_map(tmp_path, monkeypatch):
    """The bayesNMF() mirror computes its MAP / credible intervals through bnmf_map and agrees with the host path."""
    from bayesnmf_amd.engine import Engine
    from bayesnmf_amd.sampler import bayesNMF
    from bayesnmf_amd.convergence import new_convergence_control
    from bayesnmf_amd.setup import synth_counts
    M, Pt, _ = synth_counts(96, 64, 4, 12)
    cc = new_convergence_control(MAP_over=100, MAP_every=50, miniters=100, maxiters=300)
    s = bayesNMF(M, 4, prior="gamma", convergence_control=cc, output_dir=str(tmp_path / "o"), periodic_save=False,
                 save_all_samples=False)
    assert "RMSE" in s.MAP                                          # came from the device
    dev = dict(P=s.MAP["P"].copy(), E=s.MAP["E"].copy(), lo=s.credible_intervals["P"]["lower"].copy(),
               hi=s.credible_intervals["E"]["upper"].copy())
    monkeypatch.delattr(Engine, "map")
    s.get_MAP(final=True)                                           # host path: window copy + numpy
    assert "RMSE" not in s.MAP
    assert np.allclose(dev["P"], s.MAP["P"], rtol=1e-12) and np.allclose(dev["E"], s.MAP["E"], rtol=1e-12)
    assert np.allclose(dev["lo"], s.credible_intervals["P"]["lower"], rtol=1e-12)
    assert np.allclose(dev["hi"], s.credible_intervals["E"]["upper"], rtol=1e-12)
    s.close()


@pytest.mark.parametrize("model", ["gamma_fixed", "gamma_rank", "truncnormal_mh", "truncnormal_mh_rank"])
def test_run_until_matches_blockwise_loop(model, tmp_path):
    """f2: the warm-up loop on the engine side (bnmf_run_until: blocks, MAP, MAP metrics, check_convergence_) gives the
    same chain, the same MAP-metrics table and the same stopping point as the block-by-block host loop."""
    from bayesnmf_amd.sampler import bayesNMF
    from bayesnmf_amd.convergence import new_convergence_control
    from bayesnmf_amd.setup import synth_counts
    M, _, _ = synth_counts(96, 64, 3, 21)
    if model == "gamma_fixed":
        kw, rank = dict(prior="gamma"), 3
        cc = new_convergence_control(MAP_over=100, MAP_every=50, miniters=100, maxiters=600)
    elif model == "gamma_rank":
        kw, rank = dict(prior="gamma", prop_temp=0.3), range(1, 7)
        cc = new_convergence_control(MAP_over=60, MAP_every=30, miniters=90, maxiters=420, tol=0.01)
    elif model == "truncnormal_mh_rank":
        # the default model WITH excluded signatures at the end: the last check (get_MAP(final = TRUE) + check_convergence on the kept
        # signatures, R/bayesNMF_sampler.R:364-375) must leave the same MAP-metrics row AND the same bookkeeping on both paths
        kw, rank = dict(prior="truncnormal", post_warmup=50, prop_temp=0.3), range(1, 7)
        cc = new_convergence_control(MAP_over=60, MAP_every=30, miniters=90, maxiters=420, tol=0.01)
    else:
        kw, rank = dict(prior="truncnormal", post_warmup=40), 3
        cc = new_convergence_control(MAP_over=60, MAP_every=30, miniters=90, maxiters=300, tol=0.01)
    res = []
    for eng in (True, False):
        s = bayesNMF(M, rank, convergence_control=cc, output_dir=str(tmp_path / f"o{int(eng)}"), periodic_save=False,
                     save_all_samples=False, seed=5, engine_side_convergence=eng, **kw)
        res.append(dict(iter=s.state["iter"], why=s.state.get("why"), sm=s.state["sample_metrics"].to_numpy(),
                        mm=s.state["MAP_metrics"].to_numpy(), cols=list(s.state["MAP_metrics"].columns), P=s.MAP["P"].copy(),
                        st={k: s.state.get(k) for k in ("inarow_no_change", "inarow_no_best", "inarow_na", "best_iter", "converged_iter")},
                        pm=(s.state.get("prev_MAP_metric"), s.state.get("best_MAP_metric")), keep=list(np.asarray(s.MAP["keep_sigs"]).ravel())))
        s.close()
    a, b = res
    assert a["iter"] == b["iter"] and a["why"] == b["why"] and a["st"] == b["st"]
    assert np.array_equal(np.nan_to_num(a["sm"]), np.nan_to_num(b["sm"]))          # the chains are identical
    assert a["cols"] == b["cols"] and a["mm"].shape == b["mm"].shape
    assert np.allclose(a["mm"], b["mm"], rtol=1e-10, equal_nan=True)
    assert np.allclose(a["P"], b["P"], rtol=1e-12)
    assert a["keep"] == b["keep"] and np.allclose(a["pm"], b["pm"], rtol=1e-10, equal_nan=True)
    if model == "truncnormal_mh_rank":
        assert len(a["keep"]) < 6                                                   # the case is there for excluded signatures
