"""The N > 1 path on CPU: world_size-2 / -3 gloo processes shard a batch of signals, each rank
solves its block (the CPU oracle stands in for the device solver here, and the host statement of the
device's record packer for k_pack_records), the fixed-size records {K, iter, err, idx[kmax], val[kmax]}
are all_gathered as bytes, and the result must equal the single-process solve of the whole batch."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

KMAX = 16


def test_shard_range_partitions_exactly():
    from sharding import shard_range
    for total in (0, 1, 5, 8, 4096, 32768, 33):
        for world in (1, 2, 3, 8):
            blocks = [shard_range(total, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == total
            for a, b in zip(blocks, blocks[1:]):
                assert a[1] == b[0]
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1


def _make_batch(B=6, m=48, n=160, k=4):
    rng = np.random.default_rng(77)
    A = (rng.standard_normal((m, n)) / np.sqrt(m)).astype(np.float64)
    Y = []
    for _ in range(B):
        x0 = np.zeros(n)
        x0[rng.choice(n, k, replace=False)] = 1 + np.abs(rng.standard_normal(k))
        Y.append(A @ x0)
    return A, np.stack(Y)


def _solve_block(A, Y):
    """-> (records as a torch uint8 tensor (rows, record_bytes), dense X)"""
    import oracle
    import torch
    from sharding import pack_records_host
    res = [oracle.homotopy(A, y, 1e-9, 40) for y in Y]
    X = np.stack([r[0] for r in res]) if len(Y) else np.zeros((0, A.shape[1]))
    rec = pack_records_host(X, [r[1] for r in res], [r[2] for r in res], KMAX)
    return torch.from_numpy(rec.copy()), X


def _worker(rank, world, port, tmpdir):
    import torch
    import torch.distributed as dist
    from sharding import gather_records, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    A, Y = _make_batch()
    lo, hi = shard_range(len(Y), rank, world)
    rec, _ = _solve_block(A, Y[lo:hi])
    max_rows = max(shard_range(len(Y), r, world)[1] - shard_range(len(Y), r, world)[0] for r in range(world))
    allrec = gather_records(rec, world, max_rows=max_rows)
    np.save(os.path.join(tmpdir, "rank%d.npy" % rank), allrec.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_sharded_batch_matches_single_process(tmp_path, world):
    import torch.multiprocessing as mp
    from sharding import record_bytes, shard_range, unpack_records
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    A, Y = _make_batch()
    ref_t, Xref = _solve_block(A, Y)
    ref = ref_t.numpy()
    assert ref.shape[1] == record_bytes(KMAX, np.float64)
    got0 = np.load(tmp_path / "rank0.npy")
    for r in range(1, world):
        assert np.array_equal(got0, np.load(tmp_path / ("rank%d.npy" % r)))   # same on every rank
    rows = []
    for r in range(world):
        lo, hi = shard_range(len(Y), r, world)
        rows.append(got0[r, :hi - lo])
    got = np.concatenate(rows, axis=0)
    assert np.array_equal(got, ref)
    # records decode back to the solutions and their reports
    import oracle
    for s, r in enumerate(unpack_records(got, KMAX, np.float64)):
        assert r["K"] == len(r["idx"]) and np.array_equal(r["idx"], np.nonzero(Xref[s])[0])
        assert np.array_equal(r["val"], Xref[s][r["idx"]])
        _, it, e = oracle.homotopy(A, Y[s], 1e-9, 40)
        assert r["iter"] == it and r["err"] == e


def test_record_layout_and_truncation():
    """the record layout of include/ss_hip.h and the K > kmax rule (first kmax entries by column)"""
    from sharding import pack_records_host, record_bytes, record_dtype, unpack_records
    for dt, item in ((np.float32, 4), (np.float64, 8)):
        assert record_bytes(96, dt) == (16 + 96 * (4 + item) + 7) // 8 * 8
        assert record_dtype(5, dt).itemsize == record_bytes(5, dt)
    X = np.zeros((2, 50), np.float32)
    X[0, [3, 7, 20, 41]] = [1.5, -2.0, 3.25, 4.0]
    X[1, [0, 49]] = [9.0, -1.0]
    rec = pack_records_host(X, [4, 2], [1e-4, 0.0], 3)
    assert rec.dtype == np.uint8 and rec.shape == (2, record_bytes(3, np.float32))
    out = unpack_records(rec, 3, np.float32)
    assert out[0]["K"] == 4 and list(out[0]["idx"]) == [3, 7, 20] and list(out[0]["val"]) == [1.5, -2.0, 3.25]
    assert out[1]["K"] == 2 and list(out[1]["idx"]) == [0, 49] and out[1]["iter"] == 2 and out[0]["err"] == 1e-4
    # raw words: K, iter, err, idx[0]
    w = rec[0].view(np.uint32)
    assert w[0] == 4 and w[1] == 4 and w[4] == 3 and rec[0][8:16].view(np.float64)[0] == 1e-4
