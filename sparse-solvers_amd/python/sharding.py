"""Sharding of a batch of signals across the GPUs of a node.

Signals are independent given the sensing matrix, so the batch is cut into contiguous
blocks, one per rank (A is replicated, no data-path collective).  The only exchange is one
all_gather of fixed-size records at the end — RCCL over xGMI with the `nccl` backend, gloo on
CPU tensors in the tests.  A record is what ss_hip_homotopy_solve_batch_compact_* writes on the
device (include/ss_hip.h):

    uint32 K | uint32 iter | float64 err | uint32 idx[kmax] | T val[kmax]      (padded to 8 bytes)

so the multi-GPU path never materialises a dense B x n solution array: the solver packs the records
from its own support lists and the collective moves them as bytes.
"""
import numpy as np


def shard_range(total, rank, world):
    """Contiguous block [lo, hi) of `total` signals owned by `rank`; blocks differ in size by
    at most one and cover 0..total exactly once."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def record_bytes(kmax, dtype):
    """size of one record (== ss_hip_record_bytes(kmax, dtype is float64))"""
    item = np.dtype(dtype).itemsize
    return (16 + int(kmax) * (4 + item) + 7) & ~7


def record_dtype(kmax, dtype):
    """numpy structured dtype that overlays a record buffer"""
    kmax = int(kmax)
    item = np.dtype(dtype).itemsize
    return np.dtype({"names": ["K", "iter", "err", "idx", "val"],
                     "formats": [np.uint32, np.uint32, np.float64, (np.uint32, (kmax,)), (np.dtype(dtype), (kmax,))],
                     "offsets": [0, 4, 8, 16, 16 + 4 * kmax],
                     "itemsize": record_bytes(kmax, dtype)})


def pack_records_host(X, iters, errs, kmax):
    """Host statement of the device packer (k_pack_records): dense solutions (S, n) -> (S, record_bytes)
    uint8.  Used by the CPU tests of the N > 1 path and as the checker of the device records."""
    X = np.asarray(X)
    S = X.shape[0]
    kmax = int(kmax)
    rec = np.zeros(S, dtype=record_dtype(kmax, X.dtype))
    for s in range(S):
        nz = np.nonzero(X[s])[0]
        rec["K"][s] = len(nz)
        rec["iter"][s] = iters[s]
        rec["err"][s] = errs[s]
        keep = nz[:kmax]
        rec["idx"][s, :len(keep)] = keep
        rec["val"][s, :len(keep)] = X[s][keep]
    return rec.view(np.uint8).reshape(S, -1)


def unpack_records(buf, kmax, dtype):
    """(S, record_bytes) uint8 (numpy, or anything np.asarray accepts) -> list of dicts
    {K, iter, err, idx (int64), val} with the unused tail of idx / val cut off"""
    buf = np.ascontiguousarray(np.asarray(buf, dtype=np.uint8))
    rec = buf.reshape(-1).view(record_dtype(kmax, dtype))
    out = []
    for r in rec:
        k = min(int(r["K"]), int(kmax))
        out.append({"K": int(r["K"]), "iter": int(r["iter"]), "err": float(r["err"]),
                    "idx": r["idx"][:k].astype(np.int64), "val": r["val"][:k].copy()})
    return out


def gather_records(rec, world, max_rows=None, collective=None):
    """all_gather of per-rank record buffers (torch uint8 tensors (rows, record_bytes), CPU or device).
    Ranks may own different numbers of signals: rows are zero-padded to `max_rows` (default: this rank's
    row count, for equal shards; a padding record has K = 0, iter = 0).
    Returns a (world, max_rows, record_bytes) tensor on every rank."""
    import torch
    import torch.distributed as dist
    rows = rec.shape[0] if max_rows is None else int(max_rows)
    if rec.shape[0] != rows:
        pad = torch.zeros((rows - rec.shape[0], rec.shape[1]), dtype=rec.dtype, device=rec.device)
        rec = torch.cat([rec, pad], dim=0)
    rec = rec.contiguous()
    if world == 1 and not collective:          # collective=True: run the all_gather even for one rank
        return rec.unsqueeze(0)
    # concatenated form (world*rows, width): accepted by both the nccl (RCCL) and gloo backends
    out = torch.empty((world * rec.shape[0], rec.shape[1]), dtype=rec.dtype, device=rec.device)
    dist.all_gather_into_tensor(out, rec)
    return out.view(world, rec.shape[0], rec.shape[1])
