"""Multi-chain launcher: independent chains, one per GPU (replicas only).

The reference has no multi-chain concept; chains are independent Markov chains on the same data whose Philox key
differs by `chain_id`.  There is no data-path collective.  Two ways to run them:

* `run_chains(data, rank, n_chains, devices)` in ONE process: one `bayesNMF_sampler` (= one C-ABI handle = one
  device) per chain, each driven by its own host thread (the ABI calls release the GIL); returns the list of samplers.
* one process per GPU (`torchrun` / `bench.py --gpus N`): every rank calls `run_rank(...)`; at every block boundary
  the ranks all-gather (i) the block's metric rows and their convergence flags and, (ii) when a chain has just made a MAP
  check, every chain's window statistics — mode of A, renormalised window means of P and E: N + K N + N G doubles per
  chain (SURVEY.md 8e ii; get_MAP_ R/utils.R:194-288) — (`ChainSync`), on GPUs over RCCL (backend "nccl"), in the CPU
  tests over "gloo"; (iii) `gather_window` collects the chains' last recorded samples at the end.  Ranks that finish early
  keep answering the collectives until every rank is done, so their number is the same on every rank.
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


def gather_window(sampler, dist, what=("P", "E"), last_n=None, device=None):
    """the chains' last `last_n` recorded samples of the arrays in `what` on every rank:
    {name: (world, last_n, ...)}.  All chains must hold at least `last_n` samples (default: the smallest count over the chains)."""
    import torch
    have = min(sampler.specs["convergence_control"]["MAP_over"], sampler.state["iter"])
    t = torch.tensor([have], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    n = int(t.item()) if last_n is None else min(int(last_n), int(t.item()))
    out = {}
    for name in what:
        w = np.stack([np.asarray(x, dtype=np.float64) for x in sampler._chain.window(name, n)])
        out[name] = gather_rows(w.reshape(n, -1), dist, device).reshape((dist.get_world_size(),) + w.shape)
    return out


class ChainSync:
    """Per-block exchange between the ranks of a multi-process run: every rank contributes
    [n_rows, converged, done, iter, has_map] + its block of metric rows (padded to `max_rows`); every rank receives all of
    them.  `history[c]` accumulates chain c's metric rows on every rank; `converged[c]` / `done[c]` its flags.  When any
    chain's header says it has just made a MAP check, a second all-gather carries every chain's MAP (zeros from the chains
    without one): `maps[c]` is the list of chain c's checks, each {iter, A (N), P (K x N), E (N x G)} with excluded
    signatures as zero columns / rows."""

    def __init__(self, dist, max_rows, device=None, dims=None):
        self.dist, self.max_rows, self.device = dist, int(max_rows), device
        self.world = dist.get_world_size()
        self.history = [[] for _ in range(self.world)]
        self.maps = [[] for _ in range(self.world)]
        self.converged = [False] * self.world
        self.done = [False] * self.world
        self.n_collectives = 0
        self.dims = dims                                     # (K, N, G): fixes the size of the MAP exchange
        self._n_checks_seen = 0

    def _map_vector(self, sampler):
        K, N, G = self.dims
        v = np.zeros(1 + N + K * N + N * G)
        if sampler is None:
            return v
        mp = sampler.MAP
        keep = np.asarray(mp["keep_sigs"], dtype=int)
        A = np.zeros(N); A[keep] = np.ravel(mp["A"])
        P = np.zeros((K, N)); P[:, keep] = np.asarray(mp["P"])
        E = np.zeros((N, G)); E[keep, :] = np.asarray(mp["E"])
        v[0] = sampler.state["iter"]
        v[1:1 + N] = A
        v[1 + N:1 + N + K * N] = P.ravel(order="F")
        v[1 + N + K * N:] = E.ravel(order="F")
        return v

    def _exchange(self, rows, converged, done, it, map_of=None):
        buf = np.zeros((self.max_rows + 1, NMETRIC))
        n = 0 if rows is None else len(rows)
        buf[0, :5] = [n, float(converged), float(done), it, 1.0 if map_of is not None else 0.0]
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
        has = [bool(g[c, 0, 4]) for c in range(self.world)]
        if any(has) and self.dims is not None:              # every rank sees the same headers, so every rank joins
            K, N, G = self.dims
            gm = gather_rows(self._map_vector(map_of).reshape(1, -1), self.dist, self.device)
            self.n_collectives += 1
            for c in range(self.world):
                if has[c]:
                    v = gm[c, 0]
                    self.maps[c].append(dict(iter=int(v[0]), A=v[1:1 + N].copy(), P=v[1 + N:1 + N + K * N].reshape((K, N), order="F").copy(),
                                             E=v[1 + N + K * N:].reshape((N, G), order="F").copy()))

    def _fresh_map(self, sampler):
        n = len(sampler.state["MAP_metrics"])
        fresh = n > self._n_checks_seen and "keep_sigs" in sampler.MAP
        self._n_checks_seen = n
        return sampler if fresh else None

    def block(self, sampler, rows):
        """block hook of bayesNMF_sampler.run_gibbs_sampler: called after every engine block (and its MAP check, if one fell due)."""
        self._exchange(rows, sampler.state["converged"], False, sampler.state["iter"], self._fresh_map(sampler))

    def finish(self, sampler):
        """this rank is done: its final MAP travels with the first `done` message; then keep answering until every rank is."""
        self._exchange(None, sampler.state["converged"], True, sampler.state["iter"], sampler if "keep_sigs" in sampler.MAP else None)
        while not all(self.done):
            self._exchange(None, sampler.state["converged"], True, sampler.state["iter"])

    def metrics(self, c):
        return np.concatenate(self.history[c]) if self.history[c] else np.zeros((0, NMETRIC))


def run_rank(data, rank, dist, device=None, tensor_device=None, **kw):
    """One chain on this rank (chain_id = dist rank).  Returns (sampler, sync): `sync.metrics(c)` holds every chain's
    metric rows (from iteration 2 on), `sync.converged` every chain's convergence flag, `sync.maps[c]` every chain's MAP at
    each of its checks and at its end."""
    from .sampler import bayesNMF_sampler
    cc = kw.get("convergence_control") or {}
    r = dist.get_rank()
    out = kw.pop("output_dir", None) or f"nmf_{kw.get('likelihood', 'poisson')}_{kw.get('prior', 'truncnormal')}"
    s = bayesNMF_sampler(data, rank, chain_id=r, device=0 if device is None else device, output_dir=f"{out}_chain{r}", **kw)
    sync = ChainSync(dist, max_rows=(cc.get("MAP_every", 100) if isinstance(cc, dict) else 100), device=tensor_device,
                     dims=(s.dims["K"], s.dims["N"], s.dims["G"]))
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
