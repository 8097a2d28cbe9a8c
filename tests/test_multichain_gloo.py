"""world_size-2 `gloo` tests of the multi-chain path: chains are independent replicas (one per rank); the only
collective is the per-block gather of metric rows and convergence flags.  The chains here are real
`bayesNMF_sampler`s (oracle-backed engine factory on the CPU) driven through the launcher."""
import os
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _data():
    from bayesnmf_amd.setup import synth_counts
    return synth_counts(24, 30, 2, 3, mean_total=1500)


def _cc(rank):
    from bayesnmf_amd.convergence import new_convergence_control
    # different stopping points per rank: rank 1 runs longer, so rank 0 has to keep answering the collective
    return new_convergence_control(MAP_over=40, MAP_every=20, miniters=60, maxiters=120 if rank == 0 else 200, tol=1e-9)


def _worker(rank, world, port, out, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from bayesnmf_amd.multichain import run_rank, chain_seed, gather_rows, all_converged, gather_window
    from test_abi_host import _OracleChain
    dist.init_process_group("gloo", rank=rank, world_size=world)
    M, _, _ = _data()
    s, sync = run_rank(M, 2, dist, likelihood="poisson", prior="gamma", convergence_control=_cc(rank),
                       output_dir=os.path.join(tmp, "o"), periodic_save=False, save_all_samples=False,
                       engine_factory=_OracleChain)
    own = s.state["sample_metrics"].to_numpy()[1:, :]                       # rows of iterations 2.. (9 columns)
    flag = all_converged(s.state["converged"], dist)
    g = gather_rows(np.full((1, 2), float(rank)), dist)
    win = gather_window(s, dist, what=("P", "E"), last_n=5)                   # the chains' last samples on every rank
    own_win = np.stack([np.asarray(x, dtype=np.float64) for x in s._chain.window("P", 5)])

    def full(mp):                                                            # a MAP with excluded signatures as zero columns / rows
        keep = np.asarray(mp["keep_sigs"], dtype=int)
        P = np.zeros((s.dims["K"], s.dims["N"])); P[:, keep] = mp["P"]
        E = np.zeros((s.dims["N"], s.dims["G"])); E[keep, :] = mp["E"]
        return P, E
    n_checks = len(s.state["MAP_metrics"])
    if rank == 0:
        out.put(dict(own=own, mine=sync.metrics(0), other=sync.metrics(1), iters=[s.state["iter"]], conv=list(sync.converged),
                     done=list(sync.done), n_coll=sync.n_collectives, flag=flag, g=g, seeds=(chain_seed(7, 0), chain_seed(7, 1)),
                     n_checks0=n_checks, map_iters=[[m["iter"] for m in sync.maps[c]] for c in range(2)], own_final=full(s.MAP),
                     held_final=[(sync.maps[c][-1]["P"], sync.maps[c][-1]["E"], sync.maps[c][-1]["A"]) for c in range(2)],
                     win_P=win["P"], own_win=own_win))
    else:
        out.put(dict(rank1_own=own, n_coll=sync.n_collectives, iters=[s.state["iter"]], n_checks1=n_checks, rank1_final=full(s.MAP),
                     rank1_win=own_win, rank1_held0=sync.maps[0][-1]["P"]))
    dist.barrier()
    dist.destroy_process_group()
    s.close()


def test_two_chains_through_the_launcher(tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q, str(tmp_path))) for r in range(2)]
    [p.start() for p in ps]
    res = [q.get(timeout=300), q.get(timeout=300)]
    [p.join(60) for p in ps]
    r0 = next(r for r in res if "mine" in r)
    r1 = next(r for r in res if "rank1_own" in r)
    # rank 0 received, block by block, exactly the rows each chain produced
    assert np.array_equal(r0["mine"][:, :9], r0["own"][:, :9])
    assert np.array_equal(r0["other"][:, :9], r1["rank1_own"][:, :9])
    assert r0["iters"] == [120] and r1["iters"] == [200]                   # the chains stopped independently ...
    assert r0["n_coll"] == r1["n_coll"]                                    # ... with the same number of collectives
    assert all(r0["done"]) and r0["conv"] == [True, True] and r0["flag"] is True
    assert not np.array_equal(r0["mine"][:5, 1], r0["other"][:5, 1])       # different chain_id -> different chains
    assert r0["g"].shape == (2, 1, 2) and r0["g"][1, 0, 0] == 1.0
    assert r0["seeds"][0] != r0["seeds"][1]
    # rank 0 holds BOTH chains' MAPs (mode of A, window means of P and E) — one per check of each chain plus the
    # final one that travels with its `done` message — and they are the chains' own
    assert len(r0["map_iters"][0]) == r0["n_checks0"] + 1 and len(r0["map_iters"][1]) == r1["n_checks1"] + 1
    assert r0["map_iters"][0][-1] == 120 and r0["map_iters"][1][-1] == 200
    for c, (P, E) in ((0, r0["own_final"]), (1, r1["rank1_final"])):
        assert np.array_equal(r0["held_final"][c][0], P) and np.array_equal(r0["held_final"][c][1], E)
        assert r0["held_final"][c][2].shape == (2,) and set(np.unique(r0["held_final"][c][2])) <= {0.0, 1.0}
    assert np.array_equal(r1["rank1_held0"], r0["own_final"][0])            # ... and so does every other rank
    # the final samples of both chains on rank 0
    assert r0["win_P"].shape == (2, 5, 24, 2)
    assert np.array_equal(r0["win_P"][0], r0["own_win"]) and np.array_equal(r0["win_P"][1], r1["rank1_win"])


def test_chains_in_one_process_threads(tmp_path):
    """bayesNMF(..., n_chains = 2): two samplers from one process (one host thread per chain), different chain ids."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from bayesnmf_amd.sampler import bayesNMF
    from bayesnmf_amd.convergence import new_convergence_control
    from test_abi_host import _OracleChain
    M, Pt, _ = _data()
    cc = new_convergence_control(MAP_over=40, MAP_every=20, miniters=60, maxiters=120)
    ss = bayesNMF(M, 2, likelihood="poisson", prior="gamma", convergence_control=cc, output_dir=str(tmp_path / "mc"),
                  periodic_save=False, save_all_samples=False, engine_factory=_OracleChain, n_chains=2, devices=[0])
    assert len(ss) == 2 and all(s.state["iter"] <= 120 for s in ss)
    a, b = ss[0].state["sample_metrics"], ss[1].state["sample_metrics"]
    assert not np.array_equal(a["RMSE"].to_numpy()[:10], b["RMSE"].to_numpy()[:10])
    for s in ss:
        P = s.MAP["P"] / np.linalg.norm(s.MAP["P"], axis=0)
        assert ((P.T @ (Pt / np.linalg.norm(Pt, axis=0))).max(0) > 0.95).all()
        assert os.path.exists(os.path.join(s.specs["output_dir"], "sampler.pkl"))
        s.close()
