"""Worker of tests/test_gpu_rccl_world1.py, launched by `python -m torch.distributed.run --nproc-per-node 1`: the multi-GPU path
of bench.py / faiss_amd/distributed.py on the "nccl" (= RCCL) backend with a group of ONE rank, collectives forced on.
What it exercises on a one-GPU box: RCCL communicator set-up in a process that also holds the library's HIP runtime state
(tests/conftest.py: torch first), all_reduce (bench.py's ranks_seen), all_gather of the device ids, broadcast of the trained
quantizers, barrier, the point-to-point gather of the per-rank results, the stream fence, the device merge kernel.
Reference behaviour: faiss/gpu/test/test_multi_gpu.py:31-48 (sharded == unsharded ids), faiss/IndexShards.cpp:196-265.
Prints one JSON line; BACKEND=gloo + DEVICE=cpu runs the same protocol on the host (CPU test of this file's logic)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    backend = os.environ.get("BACKEND", "nccl")
    world = int(os.environ["WORLD_SIZE"])
    rank = int(os.environ["RANK"])
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    out = {"backend": backend, "world": world}
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)
    else:
        dev = torch.device("cpu")
        dist.init_process_group("gloo")
    ones = torch.ones(1, dtype=torch.int32, device=dev)
    dist.all_reduce(ones)
    out["ranks_seen"] = int(ones.item())
    mine = torch.tensor([local_rank], dtype=torch.int32, device=dev)
    box = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(box, mine)
    out["devices"] = [int(b.item()) for b in box]

    import faiss_amd
    from faiss_amd.distributed import ReplicatedSearcher, ShardedSearcher, broadcast_arrays, replica_bounds, shard_bounds
    from oracle.pyoracle import METRIC_L2, synthetic_dataset

    d, nb, nq, k, nlist, M = 64, 30000, 300, 20, 32, 16
    xt, xb, xq = synthetic_dataset(d, 3000, nb, nq, seed=31)
    if backend != "nccl":
        # host protocol only: the local search is played by an array, the merge by the C ABI's host merge
        rs = np.random.RandomState(rank)
        Dl = np.sort(rs.rand(nq, k).astype(np.float32), axis=1)
        Il = rs.randint(0, nb, (nq, k)).astype(np.int64)
        s = ShardedSearcher(lambda _x, _k: (torch.from_numpy(Dl), torch.from_numpy(Il)),
                            lambda aD, aI, base: tuple(torch.from_numpy(a) for a in faiss_amd.merge_knn_results(
                                METRIC_L2, aD.numpy(), aI.numpy(), base)), [nb] * world, dev, force_collectives=True)
        got = s.search(torch.from_numpy(xq), k)
        cent, = broadcast_arrays([xb[:nlist]], dev, force=True)
        out["gather_ok"] = bool(rank != 0 or (np.array_equal(got[0].numpy(), Dl) and np.array_equal(got[1].numpy(), Il)))
        out["broadcast_ok"] = bool(np.array_equal(cent, xb[:nlist]))
        dist.barrier()
        dist.destroy_process_group()
        if rank == 0:
            print(json.dumps(out))
        return

    res = faiss_amd.StandardGpuResources(local_rank)
    xq_dev = torch.from_numpy(xq).to(dev)
    # ---- IVFPQ shards with global ids: quantizers trained on rank 0, broadcast through RCCL, gather + device merge
    idx = faiss_amd.GpuIndexIVFPQ(res, d, nlist, M, 8, METRIC_L2)
    if rank == 0:
        idx.train(xt)
        cent, pqc = idx.get_centroids(), idx.get_pq_centroids()
    else:
        cent, pqc = np.zeros((nlist, d), np.float32), np.zeros((M, 256, d // M), np.float32)
    cent2, pqc2 = broadcast_arrays([cent, pqc], dev, force=True)
    out["broadcast_ok"] = bool(np.array_equal(cent2, cent) and np.array_equal(pqc2, pqc))
    if rank != 0:
        idx.copy_centroids(cent2)
        idx.copy_pq_centroids(pqc2)
    lo, hi = shard_bounds(nb, world)[rank]
    idx.add_with_ids(xb[lo:hi], np.arange(lo, hi, dtype=np.int64))
    idx.nprobe = 8
    D_loc = torch.empty((nq, k), dtype=torch.float32, device=dev)
    I_loc = torch.empty((nq, k), dtype=torch.int64, device=dev)
    D_out = torch.empty((nq, k), dtype=torch.float32, device=dev)
    I_out = torch.empty((nq, k), dtype=torch.int64, device=dev)

    def local_search(_xq, kk):
        idx.search_ptr(nq, xq_dev.data_ptr(), kk, D_loc.data_ptr(), I_loc.data_ptr())
        return D_loc, I_loc

    def merge(all_D, all_I, _base):
        torch.cuda.current_stream().synchronize()
        faiss_amd.merge_knn_results_device(res, METRIC_L2, nq, k, all_D.shape[0], all_D.data_ptr(), all_I.data_ptr(), None,
                                           D_out.data_ptr(), I_out.data_ptr())
        return D_out, I_out

    s = ShardedSearcher(local_search, merge, [0] * world, dev, force_collectives=True)
    for _ in range(3):  # (the second and third call reuse the gather buffers, like the timed loop of bench.py)
        got = s.search(xq_dev, k)
    torch.cuda.synchronize()
    Dd, Id = idx.search(xq, k)
    out["ivfpq_shards_ok"] = bool(rank != 0 or (np.array_equal(got[0].cpu().numpy(), Dd) and np.array_equal(got[1].cpu().numpy(), Id)))

    # ---- Flat replicas: query blocks, gather of the result blocks
    flat = faiss_amd.GpuIndexFlatL2(res, d)
    flat.add(xb)
    per = replica_bounds(nq, world)[1]
    Db = torch.empty((per, k), dtype=torch.float32, device=dev)
    Ib = torch.empty((per, k), dtype=torch.int64, device=dev)

    def block(qlo, qhi, kk):
        if qhi > qlo:
            flat.search_ptr(qhi - qlo, xq_dev.data_ptr() + qlo * d * 4, kk, Db.data_ptr(), Ib.data_ptr())
        return Db, Ib

    rep = ReplicatedSearcher(block, nq, dev, force_collectives=True)
    got = rep.search(k)
    torch.cuda.synchronize()
    Df, If = flat.search(xq, k)
    out["flat_replicas_ok"] = bool(rank != 0 or (np.array_equal(got[0].cpu().numpy(), Df) and np.array_equal(got[1].cpu().numpy(), If)))
    el = torch.tensor([1.5 + rank], dtype=torch.float64, device=dev)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    out["max_reduce_ok"] = bool(float(el.item()) == 1.5 + world - 1)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
