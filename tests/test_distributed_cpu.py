"""world_size-2 test of the one-process-per-GPU sharded search (faiss_amd/distributed.py) on CPU:
gloo backend, per-rank "local search" played by the oracle on each rank's shard, host merge
through the C ABI (faiss_amd_merge_knn_results, no GPU needed).  The sharded result must equal
the unsharded one exactly, ties included (reference: faiss/gpu/test/test_multi_gpu.py:31-48)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, metric, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import faiss_amd
    from faiss_amd.distributed import ShardedSearcher, shard_bounds
    from oracle.pyoracle import Oracle, integer_dataset

    xb, xq = integer_dataset(16, 3001, 37, seed=4, hi=5)  # uneven split, many exact ties
    k = 25
    bounds = shard_bounds(len(xb), world)
    lo, hi = bounds[rank]

    def local_search(xq_t, kk):
        D, I = Oracle.flat_search(metric, xb[lo:hi], xq_t.numpy(), kk)
        return torch.from_numpy(D), torch.from_numpy(I)

    def merge(aD, aI, base):
        D, I = faiss_amd.merge_knn_results(metric, aD.numpy(), aI.numpy(), base)
        return torch.from_numpy(D), torch.from_numpy(I)

    s = ShardedSearcher(local_search, merge, [b - a for a, b in bounds], torch.device("cpu"))
    out = None
    for _ in range(2):  # second call reuses the gather buffers
        out = s.search(torch.from_numpy(xq), k)
    if rank == 0:
        Df, If = Oracle.flat_search(metric, xb, xq, k)
        ok = np.array_equal(out[1].numpy(), If) and np.array_equal(out[0].numpy(), Df)
        with open(out_path, "w") as f:
            f.write("OK" if ok else "MISMATCH")
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("metric", [1, 0])
def test_sharded_search_two_ranks_gloo(tmp_path, metric):
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(2, _free_port(), metric, out), nprocs=2, join=True)
    assert open(out).read() == "OK"


def _ivf_shard_worker(rank, world, port, kind, out_path):
    """IndexShards over IVF indexes that share one trained coarse quantizer (+ PQ codebook): what
    index_cpu_to_gpu_multiple(shard=True) builds (faiss/gpu/GpuCloner.cpp:325-439) and bench.py's multi-GPU IVFPQ leg
    runs, one process per shard.  Rank 0 'trains' and broadcasts the quantizers (the only collective), every rank
    adds its row range with GLOBAL ids, searches all queries on its shard (oracle restatement in the role of the
    device), the per-rank top-k are gathered and merged on rank 0 without label translation."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import faiss_amd
    from faiss_amd.distributed import ShardedSearcher, shard_bounds
    from oracle.pyoracle import METRIC_L2, Oracle, synthetic_dataset

    d, nlist, M, nb, nq, k, nprobe = 32, 16, 8, 6001, 41, 20, 5
    xt, xb, xq = synthetic_dataset(d, 500, nb, nq, seed=9)
    cent = torch.zeros((nlist, d), dtype=torch.float32)
    pq = torch.zeros((M, 256, d // M), dtype=torch.float32)
    if rank == 0:
        rs = np.random.RandomState(1)
        cent.copy_(torch.from_numpy(xt[rs.choice(len(xt), nlist, replace=False)]))
        pq.copy_(torch.from_numpy(((rs.rand(M, 256, d // M) - 0.5) * 0.6).astype(np.float32)))
    dist.broadcast(cent, 0)
    dist.broadcast(pq, 0)
    cent_np, pq_np = cent.numpy(), (pq.numpy() if kind == 1 else None)
    lo, hi = shard_bounds(nb, world)[rank]
    sizes, codes, ids, _ = Oracle.build_ivf_lists(kind, METRIC_L2, cent_np, xb[lo:hi], ids=np.arange(lo, hi), pq=pq_np)

    def local_search(xq_t, kk):
        D, I, _, _ = Oracle.ivf_search(kind, METRIC_L2, cent_np, sizes, codes, ids, xq_t.numpy(), nprobe, kk,
                                       M=M if kind else 0, pq=pq_np)
        return torch.from_numpy(D), torch.from_numpy(I)

    def merge(aD, aI, _base):
        D, I = faiss_amd.merge_knn_results(METRIC_L2, aD.numpy(), aI.numpy(), None)  # ids are global already
        return torch.from_numpy(D), torch.from_numpy(I)

    s = ShardedSearcher(local_search, merge, [0] * world, torch.device("cpu"))
    out = s.search(torch.from_numpy(xq), k)
    if rank == 0:
        fs, fc, fi, _ = Oracle.build_ivf_lists(kind, METRIC_L2, cent_np, xb, pq=pq_np)
        Df, If, _, _ = Oracle.ivf_search(kind, METRIC_L2, cent_np, fs, fc, fi, xq, nprobe, k, M=M if kind else 0, pq=pq_np)
        # faiss/gpu/test/test_multi_gpu.py:74-90: np.testing.assert_array_equal(Iref, Inew)
        ok = np.array_equal(out[1].numpy(), If) and np.array_equal(out[0].numpy(), Df)
        with open(out_path, "w") as f:
            f.write("OK" if ok else "MISMATCH")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", [0, 1])
def test_sharded_ivf_two_ranks_gloo(tmp_path, kind):
    out = str(tmp_path / "result.txt")
    mp.spawn(_ivf_shard_worker, args=(2, _free_port(), kind, out), nprocs=2, join=True)
    assert open(out).read() == "OK"


def test_shard_bounds_match_reference_split():
    from faiss_amd.distributed import shard_bounds
    # IndexShards::add: shard `no` gets rows [no*n/nshard, (no+1)*n/nshard) (faiss/IndexShards.cpp:172-175)
    assert shard_bounds(10, 3) == [(0, 3), (3, 6), (6, 10)]
    assert shard_bounds(1000000, 8)[7] == (875000, 1000000)


def _replica_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from faiss_amd.distributed import ReplicatedSearcher
    from oracle.pyoracle import Oracle, integer_dataset

    xb, xq = integer_dataset(16, 900, 301, seed=6, hi=5)  # 301 queries: blocks of 256 and 45
    k = 7
    s = ReplicatedSearcher(None, len(xq), torch.device("cpu"))
    per = s.per

    def local_search(lo, hi, kk):
        D = np.full((per, kk), np.nan, dtype=np.float32)
        I = np.full((per, kk), -7, dtype=np.int64)
        if hi > lo:
            D[: hi - lo], I[: hi - lo] = Oracle.flat_search(1, xb, xq[lo:hi], kk)
        return torch.from_numpy(D), torch.from_numpy(I)

    s.local_search = local_search
    out = None
    for _ in range(2):
        out = s.search(k)
    if rank == 0:
        Df, If = Oracle.flat_search(1, xb, xq, k)
        ok = out[0].shape == (len(xq), k) and np.array_equal(out[1].numpy(), If) and np.array_equal(out[0].numpy(), Df)
        with open(out_path, "w") as f:
            f.write("OK" if ok else "MISMATCH")
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def test_replicated_search_two_ranks_gloo(tmp_path):
    out = str(tmp_path / "result.txt")
    mp.spawn(_replica_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert open(out).read() == "OK"


def test_replica_bounds():
    from faiss_amd.distributed import replica_bounds
    # IndexReplicas::search deals out ceil(n / count) queries per replica (faiss/IndexReplicas.cpp:141-151),
    # here rounded up to whole 128-query wavefronts
    assert replica_bounds(10000, 8) == ([(0, 1280), (1280, 2560), (2560, 3840), (3840, 5120), (5120, 6400),
                                         (6400, 7680), (7680, 8960), (8960, 10000)], 1280)
    assert replica_bounds(10000, 1) == ([(0, 10000)], 10112)
    assert replica_bounds(100, 4) == ([(0, 100), (100, 100), (100, 100), (100, 100)], 128)


def _scale_shard_worker(rank, world, port, out_path):
    """The id / chunk / merge / parity logic of bench.py's configs[4] leg (sharded_scale_leg: N x rows_per_rank rows,
    chunks seeded by their GLOBAL chunk number, global ids, quantizers broadcast once, per-rank top-k gathered + merged,
    sharded_sample_check) with the oracle restatement in the role of the device and a numpy generator in the role of
    synthetic_more_device.  The merged result must equal the search of ONE index over the union of the shards, and
    sharded_sample_check must say so (and must notice a corrupted shard result)."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    import faiss_amd
    from faiss_amd.distributed import ShardedSearcher, broadcast_arrays, shard_chunks
    from oracle.pyoracle import METRIC_L2, Oracle, synthetic_dataset

    d, nlist, M, nq, k, nprobe = 32, 16, 8, 33, 20, 5
    rows_per_rank, chunk_rows = 2500, 1000  # chunks of 1000, 1000, 500 rows per rank
    xt, _, xq, dmap = synthetic_dataset(d, 500, 10, nq, seed=9, return_map=True)

    def gen(gchunk, n):  # stand-in for synthetic_more_device: rows depend on the GLOBAL chunk number only
        from faiss_amd.datasets import synthetic_more
        return synthetic_more(dmap, n, seed=1338 + gchunk)

    if rank == 0:
        rs = np.random.RandomState(1)
        cent = xt[rs.choice(len(xt), nlist, replace=False)]
        pq = ((rs.rand(M, 256, d // M) - 0.5) * 0.6).astype(np.float32)
    else:
        cent, pq = np.zeros((nlist, d), np.float32), np.zeros((M, 256, d // M), np.float32)
    cent, pq = broadcast_arrays([cent, pq], torch.device("cpu"))
    plan = shard_chunks(rows_per_rank, rank, chunk_rows)
    assert [p[2] for p in plan] == [1000, 1000, 500] and plan[0][0] == rank * 3 and plan[0][1] == rank * rows_per_rank
    rows = np.concatenate([gen(g, n) for g, _, n in plan])
    gids = np.concatenate([np.arange(i0, i0 + n) for _, i0, n in plan])
    sizes, codes, ids, _ = Oracle.build_ivf_lists(1, METRIC_L2, cent, rows, ids=gids, pq=pq)

    def local_search(xq_t, kk):
        D, I, _, _ = Oracle.ivf_search(1, METRIC_L2, cent, sizes, codes, ids, xq_t.numpy(), nprobe, kk, M=M, pq=pq)
        return torch.from_numpy(D), torch.from_numpy(I)

    def merge(aD, aI, _base):
        D, I = faiss_amd.merge_knn_results(METRIC_L2, aD.numpy(), aI.numpy(), None)
        return torch.from_numpy(D), torch.from_numpy(I)

    s = ShardedSearcher(local_search, merge, [0] * world, torch.device("cpu"))
    out = s.search(torch.from_numpy(xq), k)
    sel = np.array([0, 7, 32])
    Dl, Il = local_search(torch.from_numpy(xq), k)
    Do, Io = Dl.numpy()[sel], Il.numpy()[sel]  # "the oracle on this rank's lists" (here the local search IS the oracle)
    mD, mI = (out[0].numpy()[sel], out[1].numpy()[sel]) if rank == 0 else (None, None)
    par = bench.sharded_sample_check(True, Do, Io, mD, mI, METRIC_L2, dist, rank, world)
    bad = Do.copy()
    if rank == 1:
        bad[1, 0] = np.float32(0.0)  # a shard whose restated result differs: the merged comparison must fail
    par_bad = bench.sharded_sample_check(rank != 1, bad, Io, mD, mI, METRIC_L2, dist, rank, world)
    if rank == 0:
        # one index over the union (rows in global-id order)
        allrows = np.concatenate([gen(g, n) for r in range(world) for g, _, n in shard_chunks(rows_per_rank, r, chunk_rows)])
        fs, fc, fi, _ = Oracle.build_ivf_lists(1, METRIC_L2, cent, allrows, pq=pq)
        Df, If, _, _ = Oracle.ivf_search(1, METRIC_L2, cent, fs, fc, fi, xq, nprobe, k, M=M, pq=pq)
        ok = (np.array_equal(out[1].numpy(), If) and np.array_equal(out[0].numpy(), Df)
              and par == {"per_shard_bit_exact": [True, True], "merged_bit_exact": True}
              and par_bad["per_shard_bit_exact"] == [True, False] and not par_bad["merged_bit_exact"])
        with open(out_path, "w") as f:
            f.write("OK" if ok else "MISMATCH %r %r" % (par, par_bad))
    else:
        assert par is None and par_bad is None
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_scale_leg_logic_two_ranks_gloo(tmp_path):
    out = str(tmp_path / "result.txt")
    mp.spawn(_scale_shard_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert open(out).read() == "OK"


def test_shard_chunks():
    from faiss_amd.distributed import shard_chunks
    # 8 x 125M rows in 1M-row chunks: rank 7's first chunk is global chunk 875 with ids from 875M
    c = shard_chunks(125000000, 7)
    assert len(c) == 125 and c[0] == (875, 875000000, 1000000) and c[-1] == (999, 999000000, 1000000)
    assert shard_chunks(10, 0, 4) == [(0, 0, 4), (1, 4, 4), (2, 8, 2)] and shard_chunks(10, 1, 4)[0] == (3, 10, 4)
