"""Multi-chain launcher: independent chains, one per GPU (replicas only).

The reference has no multi-chain concept; chains are independent Markov chains on the same data whose Philox key
differs by `chain_id`.  There is no data-path collective.  Two ways to run them:

* `run_chains(data, rank, n_chains, devices)` in ONE process: one `bayesNMF_sampler` (= one C-ABI handle = one
  device) per chain, each driven by its own host thread (the ABI calls release the GIL); returns the list of samplers.
* one process per GPU (`torchrun` / `bench.py --gpus N`): every rank calls `run_rank(...)`; at every block boundary
  the ranks all-gather the block's metric rows and their convergence flags (`ChainSync`), on GPUs over RCCL
  (backend "nccl"), in the CPU tests over "gloo".  Ranks that finish early keep answering the collective until every
  rank is done, so the number of collectives is the same on every rank.
"""
import os
import threading

import numpy as np

from .engine import NMETRIC


def chain_seed(seed, chain_id):
    """Philox key of a chain: (seed_lo, seed_hi ^ chain_id) — only the chain id differs between replicas."""
    return (int(seed) & 0xFFFFFFFF, ((int(seed) >> 32) & 0xFFFFFFFF) ^ int(chain_id))


def gather_rows(rows, dist, device=None):
    """all_gather a (n_rows, n_metric) float64 block from every chain -> (world, n_rows, n_metric)."""
    import torch
    t = torch.as_tensor(np.ascontiguousarray(rows), dtype=torch.float64, device=device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return torch.stack(out).cpu().numpy()


def all_converged(flag, dist, device=None):
    """True iff every chain reports convergence (all_gather of one int per chain)."""
    import torch
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return bool(all(int(o.item()) == 1 for o in out))


class ChainSync:
    """Per-block exchange between the ranks of a multi-process run: every rank contributes
    [n_rows, converged, done, iter] + its block of metric rows (padded to `max_rows`); every rank receives all of
    them.  `history[c]` accumulates chain c's metric rows on every rank; `converged[c]` / `done[c]` its flags."""

    def __init__(self, dist, max_rows, device=None):
        self.dist, self.max_rows, self.device = dist, int(max_rows), device
        self.world = dist.get_world_size()
        self.history = [[] for _ in range(self.world)]
        self.converged = [False] * self.world
        self.done = [False] * self.world
        self.n_collectives = 0

    def _exchange(self, rows, converged, done, it):
        buf = np.zeros((self.max_rows + 1, NMETRIC))
        n = 0 if rows is None else len(rows)
        buf[0, :4] = [n, float(converged), float(done), it]
        if n:
            buf[1:n + 1] = rows
        g = gather_rows(buf, self.dist, self.device)
        self.n_collectives += 1
        for c in range(self.world):
            k = int(g[c, 0, 0])
            if k:
                self.history[c].append(g[c, 1:k + 1].copy())
            self.converged[c] = bool(g[c, 0, 1])
            self.done[c] = bool(g[c, 0, 2])

    def block(self, sampler, rows):
        """block hook of bayesNMF_sampler.run_gibbs_sampler: called after every engine block."""
        self._exchange(rows, sampler.state["converged"], False, sampler.state["iter"])

    def finish(self, sampler):
        """this rank is done: keep answering until every rank is."""
        self._exchange(None, sampler.state["converged"], True, sampler.state["iter"])
        while not all(self.done):
            self._exchange(None, sampler.state["converged"], True, sampler.state["iter"])

    def metrics(self, c):
        return np.concatenate(self.history[c]) if self.history[c] else np.zeros((0, NMETRIC))


def run_rank(data, rank, dist, device=None, tensor_device=None, **kw):
    """One chain on this rank (chain_id = dist rank).  Returns (sampler, sync): `sync.metrics(c)` holds every chain's
    metric rows (from iteration 2 on), `sync.converged` every chain's convergence flag."""
    from .sampler import bayesNMF_sampler
    cc = kw.get("convergence_control") or {}
    r = dist.get_rank()
    out = kw.pop("output_dir", None) or f"nmf_{kw.get('likelihood', 'poisson')}_{kw.get('prior', 'truncnormal')}"
    s = bayesNMF_sampler(data, rank, chain_id=r, device=0 if device is None else device, output_dir=f"{out}_chain{r}", **kw)
    sync = ChainSync(dist, max_rows=(cc.get("MAP_every", 100) if isinstance(cc, dict) else 100), device=tensor_device)
    s._block_hook = sync.block
    s.run_gibbs_sampler()
    sync.finish(s)
    return s, sync


def run_chains(data, rank, n_chains, devices=None, **kw):
    """n_chains independent chains from ONE process: chain c runs on devices[c % len(devices)] in its own host thread
    (one C-ABI handle per chain; handles are independent and thread-safe against each other).  Returns the samplers."""
    from .sampler import bayesNMF_sampler
    if devices is None:
        try:
            from .engine import device_count
            nd = max(1, device_count())
        except Exception:  # noqa: BLE001  (library not built: the engine_factory path of the CPU tests)
            nd = 1
        devices = list(range(nd))
    out = kw.pop("output_dir", None) or f"nmf_{kw.get('likelihood', 'poisson')}_{kw.get('prior', 'truncnormal')}"
    kw.pop("chain_id", None); kw.pop("device", None)
    samplers = [None] * n_chains
    errors = [None] * n_chains

    def work(c):
        try:
            s = bayesNMF_sampler(data, rank, chain_id=c, device=devices[c % len(devices)], output_dir=os.path.join(out, f"chain_{c}"), **kw)
            s.run_gibbs_sampler()
            samplers[c] = s
        except BaseException as ex:  # noqa: BLE001
            errors[c] = ex

    threads = [threading.Thread(target=work, args=(c,), name=f"chain{c}") for c in range(n_chains)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    for ex in errors:
        if ex is not None:
            raise ex
    return samplers
