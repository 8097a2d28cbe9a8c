#!/usr/bin/env python3
"""bench.py — Gibbs iterations/sec of the HIP engine at BASELINE.json's metric config.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one whole Gibbs iteration (hyper sweep -> P -> E -> Z allocation -> metrics row) of
the Poisson-Gamma model at K=96, G=10,000, N=20 on synthetic counts already resident in HBM.
One independent chain per GPU (chain_id = rank); no data-path collective; RCCL only gathers the
chains' final metrics rows.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

K_, G_, N_, R_TRUE, DATA_SEED = 96, 10000, 20, 8, 20250218
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def z_bytes(K, G, N, save_Z):
    """Algorithmic bytes one k_zalloc launch must move (state is fp64):
    M int32 + E fp64 + P fp64 read; ZsumK int32 + ZsumG int32 (+ Z int32) written."""
    b = 4 * K * G + 8 * N * G + 8 * K * N + 4 * N * G + 4 * K * N
    if save_Z:
        b += 4 * K * N * G
    return b


def pmc_traffic(save_Z):
    """HBM bytes per k_zalloc launch from the PMC counters (FETCH_SIZE / WRITE_SIZE collected in separate
    rocprofv3 --pmc passes by tools/pmc2.sh, gfx950-corrected; committed under profiles/).  None if absent."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
        return d["k_zalloc_full" if save_Z else "k_zalloc_stats"]["hbm_bytes_per_launch"]
    except Exception:
        return None


def roofline_of(chain, K, G, N, save_Z, total_counts, n_iter):
    prof = chain.profile(n_iter)
    zb = z_bytes(K, G, N, save_Z)
    z_ms = prof["k_zalloc"]
    achieved = zb / (z_ms * 1e-3) / 1e9 if z_ms > 0 else 0.0
    return prof, {"bound": "hbm", "kernel": "k_zalloc_reg" + ("<save_Z>" if save_Z else ""), "achieved": achieved,
                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(save_Z),
                  "algorithmic_bytes_per_launch": zb, "avg_launch_ms": z_ms,
                  "draws_per_s": total_counts / (z_ms * 1e-3) if z_ms > 0 else 0.0,
                  "note": "stats mode moves 6.3 MB per 40 M categorical draws: the kernel is VALU/Philox-bound, not HBM-bound (DESIGN.md 5)"
                          if not save_Z else "full mode: Z (K x N x G int32) written every iteration"}


def make_chain(M, N, seed, chain_id, device, save_Z=False):
    from bayesnmf_amd import Engine
    from bayesnmf_amd.setup import apply_hyperprior_params
    e = Engine(M, N, prior="gamma", seed=seed, chain_id=chain_id, device=device, save_Z=save_Z)
    apply_hyperprior_params(e, "gamma", M, N)
    e.init()
    return e


def cpu_baseline(M, N, budget_s=15.0):
    """The CPU oracle ("port" of the reference sweep) timed on this box's host cores on a
    bounded sample of the same workload."""
    import oracle as O
    from bayesnmf_amd.setup import apply_hyperprior_params
    cores = min(os.cpu_count() or 1, 64)
    o = O.Oracle(M, N, prior="gamma", seed=1, nthreads=cores)
    apply_hyperprior_params(o, "gamma", M, N)
    o.init()
    t0 = time.perf_counter(); o.run(1); dt = time.perf_counter() - t0
    n = int(max(2, min(200, budget_s / max(dt, 1e-3))))
    t0 = time.perf_counter(); o.run(n); dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "Gibbs iterations/s", "cores": cores, "kind": "port",
            "sample": f"{n} iterations of the same K={M.shape[0]}, G={M.shape[1]}, N={N} Poisson-Gamma sweep "
                      f"(CPU oracle, scalar C + OpenMP over columns)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--save-z", action="store_true", help="full mode: materialise Z every iteration")
    ap.add_argument("--G", type=int, default=G_)
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)

    from bayesnmf_amd.setup import synth_counts
    M, _, _ = synth_counts(K_, args.G, R_TRUE, DATA_SEED)
    chain = make_chain(M, N_, seed=1, chain_id=rank, device=local_rank, save_Z=args.save_z)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    chain.run(args.warmup, metrics=False)
    barrier()
    t0 = time.perf_counter()
    met = chain.run(args.steps, metrics=True)       # synchronises the chain's stream at the end
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    barrier()
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    gathered = None
    if dist is not None:
        from bayesnmf_amd.multichain import gather_rows
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        gathered = gather_rows(met[-1:], dist, device="cuda")   # RCCL: gather the chains' last metrics rows
    tmax = float(tmax.item())

    # roofline of the dominant kernel (k_zalloc), HIP events on the chain's own stream
    total_counts = int(M.sum())
    prof, roof = roofline_of(chain, K_, args.G, N_, args.save_z, total_counts, min(200, max(20, args.steps // 10)))
    out = None
    if rank == 0:
        out = {
            "metric": "Gibbs iters/sec at K=96, G=10k, N=20; per-chain scaling at 1/2/4/8 GPUs",
            "value": world * args.steps / tmax,
            "unit": "Gibbs iterations/s (aggregate over chains)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * tmax / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"Poisson-Gamma fixed rank N={N_}, K={K_} x G={args.G} synthetic counts, "
                                   f"one chain per GPU, {'full (save_Z)' if args.save_z else 'stats'} mode, "
                                   f"metrics every iteration",
                       "chains": world, "seed": 1, "sum_M": total_counts},
            "roofline": roof,
            "kernel_ms": prof,
        }
        if world == 1 and not args.save_z:
            # the same kernel in full mode (Z materialised): the only mode in which HBM traffic is substantial
            cz = make_chain(M, N_, seed=1, chain_id=0, device=local_rank, save_Z=True)
            cz.run(50, metrics=False)
            _, out["roofline_save_Z"] = roofline_of(cz, K_, args.G, N_, True, total_counts, 100)
            cz.close()
        if gathered is not None:
            out["chains_final_logposterior"] = [float(g[0][4]) for g in gathered]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(M, N_)
        print(json.dumps(out))
    chain.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
