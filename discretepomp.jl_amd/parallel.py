"""Multi-GPU sharding of independent filters.

One particle filter does not shard (normalisation and resampling couple all its particles),
but the callers run many independent ones: SMC^2 theta-particles (src/hmm_ibis.jl:53-56) and
MCMC chains (src/arq_main.jl:65-71).  Segment ``b`` of a batch goes to rank
``shard_of(b)`` in contiguous blocks; every rank runs its block on its own GPU with the
*global* segment index in the RNG key, so the assembled vector is identical to a single-GPU
evaluation of the whole batch.  The only collective is one all-gather of the per-segment
log-likelihood estimates (8 bytes per segment; RCCL when the process group is ``nccl``).
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np


def shard_bounds(B: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of the B segments owned by ``rank`` (sizes differ by at most 1)."""
    q, r = divmod(B, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_of(b: int, B: int, world: int) -> int:
    q, r = divmod(B, world)
    cut = r * (q + 1)
    return b // (q + 1) if b < cut else r + (b - cut) // max(q, 1)


def all_gather_segments(local: np.ndarray, B: int, group=None) -> np.ndarray:
    """Assemble the full length-B vector from every rank's block.  Single-process: identity."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return np.asarray(local, dtype=np.float64)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    width = -(-B // world)
    device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    mine = torch.full((width,), float("nan"), dtype=torch.float64, device=device)
    mine[: len(local)] = torch.as_tensor(np.asarray(local, dtype=np.float64), device=device)
    out = torch.empty(world * width, dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(out, mine, group=group)
    out = out.cpu().numpy().reshape(world, width)
    pieces = []
    for r in range(world):
        lo, hi = shard_bounds(B, world, r)
        pieces.append(out[r, : hi - lo])
    return np.concatenate(pieces)


def sharded_loglik(evaluate: Callable[[np.ndarray, int], np.ndarray], theta: np.ndarray, group=None,
                   world: Optional[int] = None, rank: Optional[int] = None) -> np.ndarray:
    """Evaluate B independent filters, theta K x B, sharded over the process group.

    ``evaluate(theta_block (K x Bl), seg_base) -> ll[Bl]`` runs this rank's block (e.g.
    ``lambda th, base: pf.loglik(th, P, seed=s, seg_base=base)``).  Every rank returns the full
    length-B vector (the host-side evidence / MH bookkeeping of the callers is then done
    redundantly on every rank, src/hmm_ibis.jl:57-65).
    """
    import torch.distributed as dist

    theta = np.asarray(theta, dtype=np.float64)
    if theta.ndim == 1:
        theta = theta[:, None]
    B = theta.shape[1]
    if world is None or rank is None:
        if dist.is_available() and dist.is_initialized():
            world, rank = dist.get_world_size(group), dist.get_rank(group)
        else:
            world, rank = 1, 0
    lo, hi = shard_bounds(B, world, rank)
    local = np.asarray(evaluate(theta[:, lo:hi], lo), dtype=np.float64) if hi > lo else np.empty(0)
    if local.shape != (hi - lo,):
        raise ValueError("evaluate must return one value per segment of its block")
    return all_gather_segments(local, B, group)


class LocalBank:
    """The bank interface of the SMC^2 driver on one process: all N theta-particles on this GPU."""

    def __init__(self, bank, N: int, npf: int):
        self.bank, self.N = bank, N
        self.pop = bank.alloc_population(N, npf)
        self.pop_f = bank.alloc_population(N, npf)

    def increment(self, theta, ymin, ymax, seed):
        return self.bank.partial_log_likelihood(self.pop, theta, ymin, ymax, seed=seed)

    def resample(self, nidx):
        self.pop.gather(nidx)

    def propose(self, theta_f, ymin, ymax, seed):
        return self.bank.partial_log_likelihood(self.pop_f, theta_f, ymin, ymax, seed=seed)

    def accept(self, mask):
        self.pop.take(self.pop_f, np.where(mask, np.arange(self.N), -1))

    def close(self):
        for p in (self.pop, self.pop_f):
            c = getattr(p, "close", None)
            if c:
                c()


class ShardedBank(LocalBank):
    """The same interface with the N theta-particles sharded over the process group: rank r owns the block
    ``shard_bounds(N, world, r)`` and its device-resident populations.  Filters run locally with the GLOBAL
    segment index in the RNG key; only log-likelihood vectors are all-gathered (8 B per theta-particle), except at
    theta-level resampling (src/hmm_ibis.jl:70-75), where ancestors may live on another GPU: every rank
    all-gathers the populations (int32 state, directly from device memory over RCCL) and keeps the rows of its
    new block.  Results are bit-identical to :class:`LocalBank`."""

    def __init__(self, bank, N: int, npf: int, group=None):
        import torch.distributed as dist
        self.bank, self.N, self.group = bank, N, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.lo, self.hi = shard_bounds(N, self.world, self.rank)
        self.width = -(-N // self.world)
        nb = max(self.hi - self.lo, 1)
        self.empty = self.hi == self.lo
        self.pop = bank.alloc_population(nb, npf)
        self.pop_f = bank.alloc_population(nb, npf)

    def _local(self, pop, theta, ymin, ymax, seed):
        if self.empty:
            return np.empty(0)
        return self.bank.partial_log_likelihood(pop, np.ascontiguousarray(theta[:, self.lo:self.hi]), ymin, ymax,
                                                seed=seed, seg_base=self.lo)

    def increment(self, theta, ymin, ymax, seed):
        return all_gather_segments(self._local(self.pop, theta, ymin, ymax, seed), self.N, self.group)

    def propose(self, theta_f, ymin, ymax, seed):
        return all_gather_segments(self._local(self.pop_f, theta_f, ymin, ymax, seed), self.N, self.group)

    def resample(self, nidx):
        import torch
        import torch.distributed as dist
        mine = self.pop.export_state()
        if dist.get_backend(self.group) != "nccl" and mine.is_cuda:
            mine = mine.cpu()  # gloo group (tests, CPU-side rendezvous): stage the migration through the host
        pad = torch.zeros((self.width,) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device)
        if not self.empty:
            pad[: self.hi - self.lo] = mine
        everything = torch.empty((self.world * self.width,) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device)
        dist.all_gather_into_tensor(everything, pad, group=self.group)
        if self.empty:
            return
        rows = []
        for g in np.asarray(nidx[self.lo:self.hi]):
            r = shard_of(int(g), self.N, self.world)
            rows.append(r * self.width + int(g) - shard_bounds(self.N, self.world, r)[0])
        self.pop.import_state(everything[torch.as_tensor(rows, device=mine.device)])

    def accept(self, mask):
        if self.empty:
            return
        m = np.asarray(mask[self.lo:self.hi])
        self.pop.take(self.pop_f, np.where(m, np.arange(self.hi - self.lo), -1))
