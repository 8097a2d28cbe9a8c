"""The RCCL path on hardware: `torch.distributed` backend "nccl" IS RCCL on ROCm.  A one-GPU box can only form a
world of one rank, which still executes the real collectives (ncclAllGather / ncclAllReduce on the device, through the RCCL
communicator) that the multi-GPU launcher uses at block boundaries.  Run in a child process: the process group is global."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_run_rank_over_a_world_of_one_nccl_rank(tmp_path):
    code = textwrap.dedent(f"""
        import os, sys
        sys.path.insert(0, {ROOT!r})
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + os.getpid() % 1000), RANK="0", WORLD_SIZE="1",
                          HSA_ENABLE_IPC_MODE_LEGACY="0")
        import numpy as np, torch, torch.distributed as dist
        from bayesnmf_amd.multichain import run_rank, gather_rows, all_converged, gather_window
        from bayesnmf_amd.convergence import new_convergence_control
        from bayesnmf_amd.setup import synth_counts
        torch.cuda.set_device(0)
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        assert dist.get_backend() == "nccl"
        M, _, _ = synth_counts(96, 400, 3, 8)
        cc = new_convergence_control(MAP_over=40, MAP_every=20, miniters=40, maxiters=100)
        s, sync = run_rank(M, 3, dist, device=0, tensor_device="cuda", prior="gamma", convergence_control=cc,
                           output_dir={str(tmp_path / 'o')!r}, periodic_save=False, save_all_samples=False, seed=2)
        own = s.state["sample_metrics"].to_numpy()[1:, :9]
        got = sync.metrics(0)[:, :9]
        assert got.shape == own.shape and np.array_equal(np.nan_to_num(got), np.nan_to_num(own)), "rows gathered over RCCL differ from the chain's own"
        assert sync.n_collectives >= 5 and sync.done == [True]
        # SURVEY 8e (ii) / (iii) over RCCL: the chain's MAP at every check and at its end, and its last samples
        assert len(sync.maps[0]) == len(s.state["MAP_metrics"]) + 1 and sync.maps[0][-1]["iter"] == s.state["iter"]
        keep = np.asarray(s.MAP["keep_sigs"], dtype=int)
        assert np.array_equal(sync.maps[0][-1]["P"][:, keep], np.asarray(s.MAP["P"])) and np.array_equal(sync.maps[0][-1]["E"][keep, :], np.asarray(s.MAP["E"]))
        win = gather_window(s, dist, what=("P", "E"), last_n=4, device="cuda")
        assert win["P"].shape == (1, 4, 96, 3) and win["E"].shape == (1, 4, 3, 400)
        assert np.array_equal(win["E"][0, -1], np.asarray(s._chain.window("E", 1)[0]))
        g = gather_rows(np.arange(6.0).reshape(2, 3), dist, device="cuda")
        assert g.shape == (1, 2, 3) and np.array_equal(g[0], np.arange(6.0).reshape(2, 3))
        t = torch.tensor([3.5], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t.item()) == 3.5 and all_converged(True, dist, device="cuda")
        dist.barrier()
        dist.destroy_process_group()
        s.close()
        print("RCCL_OK", sync.n_collectives)
    """)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_bench_two_ranks_rehearsal():
    """SURVEY.md 8(e), VERDICT r4 item 8: `bench.py --gpus 2` must not run for the first time in a SCALE measurement.  A one-GPU box
    cannot measure two GPUs, so BNMF_BENCH_REHEARSE=1 puts both ranks on GPU 0 and gathers over gloo: the launcher path (spawn_ranks),
    the process group set-up with stdout kept clean, the barriers and the all-reduce(MAX) of the timed region, the gathers of the
    chains' last rows and MAP statistics all execute.  Exactly ONE JSON line on stdout, marked as a rehearsal, two different chains."""
    import json
    env = dict(os.environ, BNMF_BENCH_REHEARSE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--no-secondary",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["warmup"] == 5 and "REHEARSAL" in d
    assert d["scaling"] == "weak" and d["config"]["chains"] == 2 and len(d["rep_values"]) == d["reps"]
    lp = d["chains_final_logposterior"]
    assert len(lp) == 2 and lp[0] != lp[1] and all(np.isfinite(lp))          # chain_id = rank: two different chains
    assert d["collectives"]["world"] == 2 and len(d["collectives"]["gathered"]) == 2
    assert d["value"] > 0 and abs(d["value"] - 2 * 20 / (d["ms_per_step"] * 1e-3 * 20)) / d["value"] < 1e-9
