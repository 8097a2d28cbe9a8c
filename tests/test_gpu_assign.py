"""bnmf_assign (posterior reference assignment) against numpy / scipy: per-sample cosine matrices,
one Hungarian assignment per sample (scipy.optimize.linear_sum_assignment), cosine-weighted votes, credible bounds."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _np_assign(Ps, ref, keep, MAP_P, ci):
    from scipy.optimize import linear_sum_assignment
    K, N = Ps[0].shape
    R = ref.shape[1]
    sig = np.where(keep)[0]
    rn = ref / np.linalg.norm(ref, axis=0)
    votes = np.zeros((N, R))
    cos_all = []
    for P in Ps:
        Pk = P[:, sig]
        cos = (Pk / np.linalg.norm(Pk, axis=0)).T @ rn                  # pairwise_sim, R/helpers.R:218-268
        r, c = linear_sum_assignment(-cos)                              # RcppHungarian::HungarianSolver(-sim)
        votes[sig[r], c] += cos[r, c]
        cos_all.append(cos)
    cos_all = np.stack(cos_all)
    asg = np.full(N, -1)
    out = dict(votes=votes, assigned=asg, MAP_cosine=np.full(N, np.nan), lower=np.full(N, np.nan), upper=np.full(N, np.nan))
    for i, n in enumerate(sig):
        if not (votes[n] > 0).any():                                    # never assigned (more signatures than references):
            continue                                                    # the reference labels it "None"
        j = int(np.argmax(votes[n]))
        asg[n] = j
        out["MAP_cosine"][n] = MAP_P[:, n] @ ref[:, j] / np.sqrt((MAP_P[:, n] ** 2).sum() * (ref[:, j] ** 2).sum())
        out["lower"][n], out["upper"][n] = np.quantile(cos_all[:, i, j], [(1 - ci) / 2, 1 - (1 - ci) / 2])
    return out


@pytest.mark.parametrize("R,lr,N", [(79, False, 5), (79, True, 5), (3, False, 5), (150, False, 5), (8, False, 12), (200, False, 20)])
def test_assign_matches_scipy(R, lr, N):
    """(150 / 200 references: several columns per lane of the assignment kernel; 3 and 8: more signatures than references)"""
    from bayesnmf_amd import Engine
    from bayesnmf_amd.setup import synth_counts, apply_hyperprior_params
    rng = np.random.default_rng(R)
    cosmic = np.load(os.path.join(GOLD, "cosmic_v3.3.1_sbs.npz"))["P"]
    ref = cosmic[:, :R] if R <= cosmic.shape[1] else rng.dirichlet(np.ones(96), size=R).T
    M, _, _ = synth_counts(96, 64, 4, 33)
    temp = np.concatenate([np.zeros(3), 10.0 ** np.linspace(-6, 0, 60), np.ones(200)]) if lr else None
    e = Engine(M, N, prior="gamma", learning_rank=lr, seed=4, window=80, temperature=temp)
    apply_hyperprior_params(e, "gamma", M, N)
    e.init(); e.run(150)
    n = 80
    m = e.map(n, None)
    keep = (m["A"].ravel() == 1).astype(np.int32) if lr else np.ones(N, dtype=np.int32)
    if keep.sum() == 0:
        keep[:] = 1
    Ps = [P for P, u in zip(e.window("P", n), m["used"]) if u]
    got = e.assign(n, ref, used=m["used"].astype(np.int32), keep=keep, MAP_P=m["P"], credible_interval=0.9)
    want = _np_assign(Ps, ref, keep.astype(bool), m["P"], 0.9)
    assert np.allclose(got["votes"], want["votes"], rtol=1e-11, atol=1e-13)
    assert np.array_equal(got["assigned"], want["assigned"])
    k = keep.astype(bool) & (want["assigned"] >= 0)
    assert np.allclose(got["MAP_cosine"][k], want["MAP_cosine"][k], rtol=1e-12)
    assert np.allclose(got["lower_cosine"][k], want["lower"][k], rtol=1e-12) and np.allclose(got["upper_cosine"][k], want["upper"][k], rtol=1e-12)
    assert np.isnan(got["MAP_cosine"][~k]).all() and (got["assigned"][~k] == -1).all()
    e.close()


def test_reference_example_assigned_to_its_cosmic_signatures(tmp_path):
    """Known answer: the reference's example data were generated from COSMIC SBS signatures (its P columns); the
    ensemble assignment of a fixed-rank fit names exactly those catalogue entries, with high MAP cosine."""
    from bayesnmf_amd.sampler import bayesNMF
    from bayesnmf_amd.convergence import new_convergence_control
    d = np.load(os.path.join(GOLD, "reference_example_data.npz"))
    c = np.load(os.path.join(GOLD, "cosmic_v3.3.1_sbs.npz"))
    cosmic, names = c["P"], list(c["signatures"])
    truth = set()
    for j in range(d["P"].shape[1]):                                        # which catalogue columns generated the data
        cs = (cosmic / np.linalg.norm(cosmic, axis=0)).T @ (d["P"][:, j] / np.linalg.norm(d["P"][:, j]))
        truth.add(names[int(np.argmax(cs))])
    cc = new_convergence_control(MAP_over=200, MAP_every=100, miniters=300, maxiters=800)
    s = bayesNMF(d["M"], 4, prior="gamma", convergence_control=cc, output_dir=str(tmp_path / "o"), periodic_save=False,
                 save_all_samples=False, seed=3)
    res = s.assign_signatures_ensemble(cosmic, reference_names=names)
    a = res["assignments"]
    assert set(a["sig_ref"]) == truth, (set(a["sig_ref"]), truth)
    assert (a["MAP_cosine"] > 0.9).all() and (a["lower_cosine"] <= a["upper_cosine"]).all()
    v = res["votes"]
    assert np.allclose(v.groupby("sig_est")["prop_votes"].sum(), 1.0)
    assert s.reference_comparison["assignments"] is a
    s.close()


def test_vignette_call_reproduces_the_tutorials_assignment_table(tmp_path):
    """The one outcome the reference publishes about its own output (vignettes/bayesNMF_tutorial.pdf p.10-13, text decoded from
    the PDF): the tutorial's call — the DEFAULT model, `bayesNMF(data$M, rank = 1:10)`: Poisson likelihood, truncated-normal
    prior, MH, SBFI — on the bundled example learns rank 4, and its assignments table names SBS26, SBS2, SBS40, SBS58 with
    MAP cosine similarities 0.9992822, 0.9996010, 0.9642135, 0.9992944.  Another RNG, another chain: the names must match exactly,
    the cosines within +-0.03 (the chain-to-chain spread of a 64-sample fit) — on each of six engine seeds."""
    from bayesnmf_amd.sampler import bayesNMF
    d = np.load(os.path.join(GOLD, "reference_example_data.npz"))
    c = np.load(os.path.join(GOLD, "cosmic_v3.3.1_sbs.npz"))
    cosmic, names = c["P"], [str(x) for x in c["signatures"]]
    tutorial = {"SBS26": 0.9992822, "SBS2": 0.9996010, "SBS40": 0.9642135, "SBS58": 0.9992944}
    # one engine seed of a statistical known-answer test says little about its pass rate — six seeds, every one of them
    # must learn rank 4 and name the four signatures
    for seed in (11, 1, 2, 3, 4, 5):
        s = bayesNMF(d["M"], range(1, 11), output_dir=str(tmp_path / f"o{seed}"), periodic_save=False, save_all_samples=False, seed=seed)
        assert s.specs["likelihood"] == "poisson" and s.specs["prior"] == "truncnormal" and s.specs["MH"] and s.specs["rank_method"] == "SBFI"
        assert int(np.sum(s.MAP["A"])) == 4, (seed, s.MAP["A"])
        a = s.assign_signatures_ensemble(cosmic, reference_names=names)["assignments"]
        got = dict(zip([str(x) for x in a["sig_ref"]], [float(x) for x in a["MAP_cosine"]]))
        assert set(got) == set(tutorial), (seed, got, tutorial)
        for nm, cs in tutorial.items():
            assert abs(got[nm] - cs) <= 0.03, (seed, nm, got[nm], cs)
        s.close()
