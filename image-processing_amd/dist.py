"""Sharding of independent units (images, point sets, files) across ranks and
the single collective of the path: an all-gather of the per-rank feature block.

Every unit is independent, so the data path has no exchange step; one rank
owns one GPU (one process per GPU, torch.distributed over RCCL on the GPU box,
gloo in CPU tests).  Shards are contiguous blocks of the sorted unit list, the
reference's deterministic lexicographic order
(extract_2d_hu.py:75, extract_3d_pca_features.py:318-320).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


# A process group of fewer ranks than this has nothing to exchange and the helpers below hand their input back.
# (A test on one GPU sets it to 1, so that the very collective calls of the N-rank case run on the RCCL backend.)
MIN_RANKS_FOR_COLLECTIVES = 2


def _exchanging() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() >= MIN_RANKS_FOR_COLLECTIVES


def shard_range(n_units: int, rank: int, world: int) -> Tuple[int, int]:
    """[lo, hi) of rank's contiguous block: ceil(n/world) units per rank, the
    last ranks may get fewer (or none)."""
    per = -(-n_units // world) if world > 0 else n_units
    lo = min(n_units, rank * per)
    return lo, min(n_units, lo + per)


def shard_sizes(n_units: int, world: int) -> List[int]:
    return [hi - lo for lo, hi in (shard_range(n_units, r, world) for r in range(world))]


def lpt_assign(costs: Sequence[float], world: int) -> List[List[int]]:
    """Longest-processing-time-first assignment of units with unequal cost
    (variable-size clouds): returns the unit indices per rank."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads = [0.0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        out[r].append(i)
        loads[r] += costs[i]
    return [sorted(v) for v in out]


def all_gather_rows(local: torch.Tensor, n_units: int) -> torch.Tensor:
    """Gathers the per-rank [rows_r, F] blocks of a contiguous sharding of
    n_units rows into the [n_units, F] table on every rank.  Without an
    initialised process group (single process) returns `local`."""
    if not _exchanging():
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    per = -(-n_units // world)
    F = local.shape[1]
    lo, hi = shard_range(n_units, rank, world)
    if local.shape[0] != hi - lo:
        raise ValueError(f"rank {rank} holds {local.shape[0]} rows, its shard has {hi - lo}")
    # equal-sized slots so that one all_gather_into_tensor moves everything
    slot = torch.zeros((per, F), dtype=local.dtype, device=local.device)
    slot[: hi - lo] = local
    table = torch.empty((world * per, F), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(table, slot)
    return table[:n_units]


def all_gather_tagged_rows(rows: torch.Tensor, ids: Sequence[int], slot: int, n_units: int) -> torch.Tensor:
    """Gathers rows of units that are NOT sharded contiguously (clouds handed to
    the ranks by cost, see lpt_assign): every rank passes its [k_r, F] rows and
    the unit index of each, `slot` >= every k_r is the padded block height, and
    every rank gets the [n_units, F] table in unit order.  One all-gather of
    [slot, F + 1] blocks whose first column carries the unit index (-1 in the
    padding).  Raises if a unit is missing or arrives twice.  Without an
    initialised process group the rows are put in order locally."""
    if n_units >= 1 << 24:
        raise ValueError("unit indices travel as float32: fewer than 2^24 units")
    k, F = rows.shape
    if k != len(ids) or k > slot:
        raise ValueError(f"{k} rows, {len(ids)} ids, slot of {slot}")
    block = torch.zeros((slot, F + 1), dtype=rows.dtype, device=rows.device)
    block[:, 0] = -1.0
    if k:
        block[:k, 0] = torch.as_tensor(list(ids), dtype=rows.dtype, device=rows.device)
        block[:k, 1:] = rows
    if _exchanging():
        gathered = torch.empty((dist.get_world_size() * slot, F + 1), dtype=rows.dtype, device=rows.device)
        dist.all_gather_into_tensor(gathered, block)
    else:
        gathered = block
    tag = gathered[:, 0]
    real = tag >= 0
    where = tag[real].to(torch.int64)
    seen = torch.bincount(where, minlength=n_units) if where.numel() else torch.zeros(n_units, dtype=torch.int64, device=rows.device)
    if seen.numel() != n_units or not bool((seen == 1).all()):
        raise ValueError("tagged gather: a unit is missing, out of range or present twice")
    table = torch.empty((n_units, F), dtype=rows.dtype, device=rows.device)
    table[where] = gathered[real][:, 1:]
    return table


def describe_ranks(step_ms: float, device_name: str) -> dict:
    """What a reader needs to confirm that a multi-process run was one: the size
    of the process group as the backend reports it, every rank's device, and the
    spread of the ranks' own step times."""
    if not _exchanging():
        return {"ranks": 1, "backend": None, "devices": [device_name],
                "ms_per_step_rank_min": step_ms, "ms_per_step_rank_max": step_ms}
    world = dist.get_world_size()
    got: list = [None] * world
    dist.all_gather_object(got, (dist.get_rank(), float(step_ms), str(device_name)))
    got.sort()
    return {"ranks": world, "backend": dist.get_backend(), "devices": [g[2] for g in got],
            "ms_per_step_rank_min": min(g[1] for g in got), "ms_per_step_rank_max": max(g[1] for g in got)}


# --------------------------------------------------------------------------
# One image spread over the ranks as row bands (images too tall for one GPU's
# share of a batch): the second, and last, exchange step of the path.
# --------------------------------------------------------------------------
LIMB_BITS = 48


def band_range(height: int, rank: int, world: int, align: int = 256) -> Tuple[int, int]:
    """Rows [y0, y1) of rank's band: equal shares rounded up to `align` rows
    (the tile height of the moment kernel), later ranks may get fewer or none."""
    per = -(-height // world) if world > 0 else height
    per = -(-per // align) * align
    y0 = min(height, rank * per)
    return y0, min(height, y0 + per)


def to_limbs(value: int) -> List[int]:
    """A non-negative Python integer as three 48-bit limbs (what fx_hu_partial_u8 writes)."""
    mask = (1 << LIMB_BITS) - 1
    return [value & mask, (value >> LIMB_BITS) & mask, value >> (2 * LIMB_BITS)]


def from_limbs(limbs) -> int:
    """The integer a (possibly summed, unnormalised) limb triple stands for."""
    return int(limbs[0]) + (int(limbs[1]) << LIMB_BITS) + (int(limbs[2]) << (2 * LIMB_BITS))


def all_reduce_partial(partial: torch.Tensor) -> torch.Tensor:
    """Sums the int64 [B, 10, 3] limb arrays of all ranks in place (one
    all-reduce; exact, a limb grows by at most log2(world) bits).  Without an
    initialised process group returns `partial` unchanged."""
    if partial.dtype != torch.int64 or partial.dim() != 3 or tuple(partial.shape[1:]) != (10, 3):
        raise ValueError("partial must be int64 [B, 10, 3]")
    if _exchanging():
        if dist.get_world_size() > (1 << 15):
            raise ValueError("limb sums are exact for up to 2^15 ranks")
        dist.all_reduce(partial, op=dist.ReduceOp.SUM)
    return partial


def hu_rowbands(band: torch.Tensor, y_offset: int, thresh: int = -1, on_value: int = 255, **outputs) -> dict:
    """Hu features of images held as row bands across the ranks: every rank
    passes its band [B, rows, W] (device tensor, possibly zero rows) and the
    row it starts at; all ranks get the features of the whole images.  Exact:
    integer partial moments, an integer all-reduce, one rounding."""
    from . import engine
    B = band.shape[0]
    if band.shape[1] > 0:
        partial = engine.hu_partial(band, y_offset, thresh, on_value)
    else:
        partial = torch.zeros((B, 10, 3), dtype=torch.int64, device=band.device)
    return engine.hu_from_partial(all_reduce_partial(partial), **outputs)


class GatherPipeline:
    """All-gathers of equal-sized per-rank blocks that overlap with the next
    step's kernels: submit() enqueues the collective asynchronously and returns
    at once; at most `depth` collectives are in flight (the oldest is waited for
    before another one is issued), drain() waits for the rest and returns the
    last table.  Without a process group every call is a no-op that hands the
    block back."""

    def __init__(self, depth: int = 2):
        self.depth = max(1, int(depth))
        self.pending: List[tuple] = []
        self.last = None
        self.active = _exchanging()

    def submit(self, block: torch.Tensor) -> None:
        if not self.active:
            self.last = block
            return
        while len(self.pending) >= self.depth:
            self._pop()
        block = block.contiguous()
        table = torch.empty((dist.get_world_size() * block.shape[0],) + tuple(block.shape[1:]), dtype=block.dtype,
                            device=block.device)
        work = dist.all_gather_into_tensor(table, block, async_op=True)
        self.pending.append((work, table, block))   # the block must stay alive until the collective has read it

    def _pop(self) -> None:
        work, table, _block = self.pending.pop(0)
        work.wait()
        self.last = table

    def drain(self):
        while self.pending:
            self._pop()
        return self.last
