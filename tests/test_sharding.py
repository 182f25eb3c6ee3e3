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


# ---------------------------------------------------------------- the PRODUCT under world_size 2 (one MI355X)

def _gpu_batch(B=64, m=256, n=2048, k=8):
    rng = np.random.default_rng(505)
    A = (rng.standard_normal((m, n)) / np.sqrt(m)).astype(np.float32)
    Y = np.zeros((B, m), np.float32)
    for b in range(B):
        x0 = np.zeros(n)
        x0[rng.choice(n, k, replace=False)] = 1 + np.abs(rng.standard_normal(k))
        Y[b] = (A.astype(np.float64) @ x0).astype(np.float32)
    return A, Y


def _gpu_rank(rank, world, port, tmpdir, B):
    """one rank of configs[3]'s code path: its shard_range block through ss_hip_homotopy_solve_batch_compact_f32
    (device inputs, records packed on the device by k_pack_records), then the gather of the record bytes"""
    import torch
    import torch.distributed as dist
    import sship
    from sharding import gather_records, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    A, Y = _gpu_batch(B)
    lo, hi = shard_range(len(Y), rank, world)
    with sship.Homotopy(torch.from_numpy(A).to("cuda:0")) as h:
        Yd = torch.from_numpy(Y[lo:hi]).to("cuda:0")
        rec = torch.zeros((hi - lo, h.record_bytes(KMAX)), dtype=torch.uint8, device="cuda:0")
        h.solve_batch_compact(Yd, 1e-3, 64, kmax=KMAX, out=rec)
        torch.cuda.synchronize()
        st = h.stats()
    max_rows = max(shard_range(len(Y), r, world)[1] - shard_range(len(Y), r, world)[0] for r in range(world))
    # (two ranks on ONE GPU cannot form an RCCL communicator: the bytes cross through gloo; on an 8-GPU node
    # bench.py --gpus N gathers the same device buffers with the nccl backend)
    allrec = gather_records(rec.cpu(), world, max_rows=max_rows)
    np.savez(os.path.join(tmpdir, "rank%d.npz" % rank), rec=allrec.numpy(), col_rounds=st["batch_col_rounds"],
             solves=st["solves"])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("B", [64, 51])
def test_product_sharded_batch_two_ranks_one_gpu(tmp_path, B):
    """The product under world_size 2: two gloo ranks share the one MI355X of the test box, each solves its
    shard_range block with libss_hip.so (lock-step column form, compact records packed on the device) and the
    records are gathered as bytes.  The gathered buffer must equal, byte for byte, what ONE process writes for the
    whole batch (a signal's arithmetic does not depend on which other signals share its batch), and decode to the
    oracle's supports.  B = 51: ragged shards (26 + 25 rows, zero-padded gather)."""
    import oracle
    import sship
    import torch
    import torch.multiprocessing as mp
    from sharding import shard_range, unpack_records
    world = 2
    port = 31800 + (os.getpid() % 1000) + B
    mp.spawn(_gpu_rank, args=(world, port, str(tmp_path), B), nprocs=world, join=True)
    A, Y = _gpu_batch(B)
    with sship.Homotopy(torch.from_numpy(A).to("cuda:0")) as h:
        rec1 = torch.zeros((B, h.record_bytes(KMAX)), dtype=torch.uint8, device="cuda:0")
        h.solve_batch_compact(torch.from_numpy(Y).to("cuda:0"), 1e-3, 64, kmax=KMAX, out=rec1)
        torch.cuda.synchronize()
    ref = rec1.cpu().numpy()
    got0 = np.load(tmp_path / "rank0.npz")
    got1 = np.load(tmp_path / "rank1.npz")
    assert np.array_equal(got0["rec"], got1["rec"])                    # every rank holds the whole result
    assert int(got0["col_rounds"]) > 0 and int(got1["col_rounds"]) > 0     # the lock-step device path ran in both
    rows = []
    for r in range(world):
        lo, hi = shard_range(B, r, world)
        rows.append(got0["rec"][r, :hi - lo])
        assert not got0["rec"][r, hi - lo:].any()                       # padding records are zero
    got = np.concatenate(rows, axis=0)
    assert np.array_equal(got, ref), "sharded records differ from the single-process records"
    for b, r in enumerate(unpack_records(got, KMAX, np.float32)):
        if b % 8:
            continue
        xo, ito, eo = oracle.homotopy(A, Y[b], 1e-3, 64)
        assert r["iter"] == ito and np.array_equal(r["idx"], np.nonzero(xo)[0])
        assert np.abs(r["val"] - xo[r["idx"]]).max() <= 1e-5 * np.abs(xo).max()


@pytest.mark.gpu
def test_bench_multi_gpu_launch_rehearsed_on_one_gpu():
    """configs[3]'s launch line, rehearsed on the one GPU a test box has: `python -m torch.distributed.run --nproc-per-node 1 bench.py
    --workload batched ...` with SS_BENCH_DIST=1 walks everything an N > 1 launch executes — RCCL init, the broadcast of A from rank 0,
    the barriers, the per-step all_gather of the records, the all_reduce of the timing — and prints the one JSON line the driver
    parses: its metric / unit / scaling, `value` = `scale_value` (the batched workload's signals/s: the one axis of a --gpus sweep), the
    roofline block, every planted support exact.  (No scaling curve comes out of one GPU: that stays unmeasured.)"""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SS_BENCH_DIST="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29500 + (os.getpid() % 400)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--workload", "batched", "--batch", "1536",
           "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["unit"] == "signals/s" and out["scaling"] == "weak" and out["n_gpus"] == 1 and out["steps"] == 2
    assert out["value"] > 0 and out["scale_value"] == out["value"]
    assert out["roofline"]["frac"] > 0 and out["roofline"]["bound"] in ("hbm", "mfma")
    assert out["recovered"]["support_exact"] == out["recovered"]["signals_checked"] == 1536
    assert "configs[3]" in out["config"]["workload"]
