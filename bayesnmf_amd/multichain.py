"""Multi-chain helpers: independent chains, one per GPU/rank (replicas only).  The only
communication is a gather of per-chain metric rows / convergence flags at block boundaries; on GPUs the
`dist` backend is "nccl" (= RCCL over xGMI), in the CPU tests it is "gloo"."""
import numpy as np
import torch


def chain_seed(seed, chain_id):
    """Philox key of a chain: (seed_lo, seed_hi ^ chain_id) — only the chain id differs between replicas."""
    return (int(seed) & 0xFFFFFFFF, ((int(seed) >> 32) & 0xFFFFFFFF) ^ int(chain_id))


def gather_rows(rows, dist, device=None):
    """all_gather a (n_rows, n_metric) float64 block from every chain -> (world, n_rows, n_metric)."""
    t = torch.as_tensor(np.ascontiguousarray(rows), dtype=torch.float64, device=device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return torch.stack(out).cpu().numpy()


def all_converged(flag, dist, device=None):
    """True iff every chain reports convergence (all_gather of one int per chain)."""
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return bool(all(int(o.item()) == 1 for o in out))
