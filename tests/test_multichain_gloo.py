"""world_size-2 `gloo` test of the multi-chain path: chains are independent replicas (one per rank);
the only collective is the gather of per-chain metric rows at block boundaries."""
import os
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from bayesnmf_amd.multichain import gather_rows, all_converged, chain_seed
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows = np.arange(22, dtype=np.float64).reshape(2, 11) + 100 * rank      # this chain's block of metric rows
    g = gather_rows(rows, dist)
    flag = all_converged(rank == 0, dist)
    if rank == 0:
        out.put((g, flag, chain_seed(7, rank), chain_seed(7, 1)))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_rows_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    g, flag, s0, s1 = q.get(timeout=120)
    [p.join(60) for p in ps]
    assert g.shape == (2, 2, 11)
    assert np.array_equal(g[1] - g[0], np.full((2, 11), 100.0))
    assert flag is False          # rank 1 has not converged
    assert s0 != s1               # chains differ only by chain_id in the Philox key
