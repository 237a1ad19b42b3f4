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


def migration_plan(nidx, N: int, world: int, rank: int, loopback: bool = False):
    """Who sends which populations to whom when the theta-particles are resampled with global ancestors ``nidx``
    (src/hmm_ibis.jl:70-75) and block ``shard_bounds(N, world, r)`` lives on rank r.  A rank sends each of its
    populations at most ONCE to each rank that needs it (duplicates are expanded by the receiver), so a rank receives
    at most one population per slot of its block -- never the whole bank.  Everything is derived from ``nidx`` alone,
    which every rank holds, so no plan has to be exchanged.

    Returns a dict for ``rank``: ``send_rows`` (local row indices, grouped by destination rank in rank order),
    ``send_counts`` / ``recv_counts`` ([world]), ``local_anc`` (for every slot of the block: the local row it copies
    if its ancestor lives here, else the slot itself), ``remote_slots`` and ``remote_pick`` (slot b takes row
    ``remote_pick`` of the receive buffer).

    ``loopback`` (test hook): a rank treats ITSELF like any other rank -- populations whose ancestor is local also
    travel through the collective (send to self) instead of being copied on the device.  Same result; it makes the
    all-to-all carry data even in a world of one rank."""
    nidx = np.asarray(nidx, dtype=np.int64)
    bounds = [shard_bounds(N, world, r) for r in range(world)]
    owner = np.empty(N, dtype=np.int64)
    for r, (lo, hi) in enumerate(bounds):
        owner[lo:hi] = r
    lo, hi = bounds[rank]
    send_rows, send_counts = [], np.zeros(world, dtype=np.int64)
    for dst, (dlo, dhi) in enumerate(bounds):  # what this rank sends: its rows that block `dst` descends from
        if dst == rank and not loopback:
            continue
        anc = nidx[dlo:dhi]
        mine = np.unique(anc[owner[anc] == rank])
        send_rows.append(mine - lo)
        send_counts[dst] = len(mine)
    anc = nidx[lo:hi]
    src = owner[anc] if hi > lo else np.empty(0, dtype=np.int64)
    recv_counts = np.zeros(world, dtype=np.int64)
    local_anc = np.arange(hi - lo, dtype=np.int64)
    here = (src == rank) & (not loopback)
    local_anc[here] = anc[here] - lo
    remote_slots = np.flatnonzero(~here)
    remote_pick = np.empty(len(remote_slots), dtype=np.int64)
    offset = 0
    for r in range(world):
        if r == rank and not loopback:
            continue
        sel = src[remote_slots] == r
        uniq = np.unique(anc[remote_slots][sel])  # the order rank r sends them in
        recv_counts[r] = len(uniq)
        remote_pick[sel] = offset + np.searchsorted(uniq, anc[remote_slots][sel])
        offset += len(uniq)
    return {"send_rows": np.concatenate(send_rows) if send_rows else np.empty(0, dtype=np.int64),
            "send_counts": send_counts, "recv_counts": recv_counts, "local_anc": local_anc,
            "remote_slots": remote_slots, "remote_pick": remote_pick}


class ShardedBank(LocalBank):
    """The same interface with the N theta-particles sharded over the process group: rank r owns the block
    ``shard_bounds(N, world, r)`` and its device-resident populations.  Filters run locally with the GLOBAL
    segment index in the RNG key; only log-likelihood vectors are all-gathered (8 B per theta-particle), except at
    theta-level resampling (src/hmm_ibis.jl:70-75), where ancestors may live on another GPU: populations whose
    ancestor is local are copied on the device, the others travel in ONE all-to-all in which every rank sends each
    needed population once per destination, device to device when the group is ``nccl`` (RCCL; xGMI is a full mesh of
    point-to-point links, so an all-to-all of direct sends is its natural collective).  A rank therefore receives at
    most its own block's worth of populations (``last_migration``), not the whole bank.  Results are bit-identical to
    :class:`LocalBank`."""

    def __init__(self, bank, N: int, npf: int, group=None, loopback: bool = False):
        import torch.distributed as dist
        self.bank, self.N, self.group, self.loopback = bank, N, group, loopback
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.lo, self.hi = shard_bounds(N, self.world, self.rank)
        nb = max(self.hi - self.lo, 1)
        self.empty = self.hi == self.lo
        self.pop = bank.alloc_population(nb, npf)
        self.pop_f = bank.alloc_population(nb, npf)
        self.last_migration = {"rows_sent": 0, "rows_received": 0, "rows_local": 0}
        self.total_rows_received = 0

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
        plan = migration_plan(nidx, self.N, self.world, self.rank, self.loopback)
        on_device = dist.get_backend(self.group) == "nccl"
        send = self.pop.export_rows(plan["send_rows"])  # a copy of the rows that leave, taken before anything moves
        if not on_device and send.is_cuda:
            send = send.cpu()  # gloo group (tests, CPU-side rendezvous): stage the migration through the host
        n_recv = int(plan["recv_counts"].sum())
        recv = torch.empty((n_recv,) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        dist.all_to_all_single(recv, send.contiguous(), [int(c) for c in plan["recv_counts"]],
                               [int(c) for c in plan["send_counts"]], group=self.group)
        self.last_migration = {"rows_sent": int(plan["send_counts"].sum()), "rows_received": n_recv,
                               "rows_local": int(len(plan["local_anc"]) - len(plan["remote_slots"]))}
        self.total_rows_received += n_recv
        if self.empty:
            return
        self.pop.gather(plan["local_anc"])  # pop[b] <- pop[local ancestor] on the device
        if len(plan["remote_slots"]):
            self.pop.import_rows(plan["remote_slots"], recv[torch.as_tensor(plan["remote_pick"], device=recv.device)])

    def accept(self, mask):
        if self.empty:
            return
        m = np.asarray(mask[self.lo:self.hi])
        self.pop.take(self.pop_f, np.where(m, np.arange(self.hi - self.lo), -1))
