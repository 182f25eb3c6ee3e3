"""Sharding of a batch of signals across the GPUs of a node.

Signals are independent given the sensing matrix, so the batch is cut into contiguous
blocks, one per rank (A is replicated, no data-path collective).  The only exchange is one
all_gather of fixed-size support records at the end: for signal s the KMAX largest-|x|
entries as (index, value) pairs, zero-padded — RCCL over xGMI with the `nccl` backend, gloo
on CPU tensors in the tests.
"""
import numpy as np


def shard_range(total, rank, world):
    """Contiguous block [lo, hi) of `total` signals owned by `rank`; blocks differ in size by
    at most one and cover 0..total exactly once."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def pack_records(X, kmax):
    """X: (S, n) solutions (torch tensor) -> (S, 2*kmax) float records [idx..., val...]."""
    import torch
    S, n = X.shape
    k = min(int(kmax), n)
    _, idx = torch.topk(X.abs(), k, dim=1)
    idx, _ = torch.sort(idx, dim=1)
    val = torch.gather(X, 1, idx)
    # entries that are exactly zero carry index -1 (support smaller than kmax)
    idxf = torch.where(val != 0, idx.to(X.dtype), torch.full_like(val, -1))
    rec = torch.full((S, 2 * int(kmax)), -1, dtype=X.dtype, device=X.device)
    rec[:, int(kmax):] = 0
    rec[:, :k] = idxf
    rec[:, int(kmax):int(kmax) + k] = val
    return rec.contiguous()


def unpack_records(rec, kmax, n):
    """(S, 2*kmax) records -> list of (support indices int64 array, values array)"""
    rec = np.asarray(rec)
    out = []
    for row in rec:
        idx = row[:kmax]
        val = row[kmax:]
        keep = idx >= 0
        out.append((idx[keep].astype(np.int64), val[keep]))
    return out


def gather_records(rec, world, max_rows=None, collective=None):
    """all_gather of per-rank records.  Ranks may own different numbers of signals: rows are
    padded to `max_rows` (default: this rank's row count, for equal shards).
    Returns a (world, max_rows, width) tensor on every rank."""
    import torch
    import torch.distributed as dist
    rows = rec.shape[0] if max_rows is None else int(max_rows)
    if rec.shape[0] != rows:
        pad = torch.full((rows - rec.shape[0], rec.shape[1]), -1, dtype=rec.dtype, device=rec.device)
        rec = torch.cat([rec, pad], dim=0)
    rec = rec.contiguous()
    if world == 1 and not collective:          # collective=True: run the all_gather even for one rank
        return rec.unsqueeze(0)
    # concatenated form (world*rows, width): accepted by both the nccl (RCCL) and gloo backends
    out = torch.empty((world * rec.shape[0], rec.shape[1]), dtype=rec.dtype, device=rec.device)
    dist.all_gather_into_tensor(out, rec)
    return out.view(world, rec.shape[0], rec.shape[1])
