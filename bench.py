#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X similarity-search backend.

Metric (BASELINE.json): QPS at nq=10 000, k=100 on SIFT1M-shaped synthetic data (d=128,
nb=1M; the reference's SyntheticDataset recipe, seed 1338).  A "step" is one search of all
10 000 queries; queries and results are resident in HBM when the timed region starts.

  N = 1 : GpuIndexFlatL2 (BASELINE.json configs[1]); `value` = Flat QPS.  The same line also
          carries the IVF4096,PQ64 numbers (`ivfpq`), the roofline of the dominant kernel and
          the reference CPU path timed on this node's host cores (`cpu_baseline`).
  N > 1 : one process per GPU (torch.distributed, backend "nccl" = RCCL), total work fixed =>
          "scaling": "strong".  --multi-gpu replicas (default): what the reference builds for a
          database that fits one GPU (index_cpu_to_gpu_multiple, GpuMultipleClonerOptions::shard =
          false -> IndexReplicas): every rank holds the 1M vectors and searches its block of the
          queries; the result blocks are gathered onto rank 0 (1.5 MB per rank at N = 8, no
          merge).  --multi-gpu shards: IndexShards-style (rank r holds rows [r*nb/N, (r+1)*nb/N)),
          every rank searches all queries on its shard, the per-rank top-k are gathered
          point-to-point onto rank 0 over xGMI and merged there by the device select kernel --
          the layout for databases beyond one GPU (BASELINE.json configs[4]); at nb = 1M its
          replicated per-query work (re-rank, gather, merge) dominates.

Run:  python bench.py [--gpus N --steps K --warmup W]
      python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

D, NT, NB, NQ, K = 128, 100000, 1000000, 10000, 100
PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32
PEAK_F16_MFMA_TFLOPS = 2500.0  # same guide: dense f16/bf16 MFMA (v_mfma_f32_32x32x16_f16)
PEAK_HBM_GBS = 8000.0


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def effective_cores():
    """CPUs this process may really use: the scheduler affinity capped by the cgroup CPU quota
    (the GPU boxes expose 256 logical CPUs but cap the pod at cpu.max = 16 cores; running 256
    OpenMP threads under that quota throttles the reference to ~1% of its 16-thread speed)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(xb, xq, k, gpu_D, gpu_I, budget_s=25.0):
    """Reference CPU path (faiss IndexFlatL2, compiled unmodified into oracle/_ref) timed on this
    node's host cores on a bounded sample of the same queries; falls back to the scalar C
    restatement (kind "port") when oracle/_ref was not shipped.

    The reference blocks the search into 4096-query x 1024-row sgemm tiles
    (faiss/utils/distances.cpp:424-511), so the sample is ONE large query batch (as
    benchs/bench_gpu_sift1m.py does), never many small ones."""
    from oracle.pyoracle import METRIC_L2, Oracle, Ref
    cores = effective_cores()
    if Ref.available():
        os.environ.setdefault("MKL_NUM_THREADS", str(cores))
        Ref.set_threads(cores)
        idx = Ref.index_factory(xb.shape[1], "Flat")
        idx.add(xb)
        ns = min(len(xq), 2048)
        idx.search(xq[:ns], k)  # warm-up (MKL init, page faults)
        # probe the full core allowance and half of it (SMT siblings rarely help sgemm), keep the faster
        best = None
        for nthr in sorted({cores, max(1, cores // 2)}, reverse=True):
            Ref.set_threads(nthr)
            t0 = time.time()
            Dr, Ir = idx.search(xq[:ns], k)
            dt = time.time() - t0
            log("cpu baseline probe: %d threads -> %.1f QPS" % (nthr, ns / dt))
            if best is None or dt < best[0]:
                best = (dt, nthr, Dr, Ir)
        dt, nthr, Dr, Ir = best
        Ref.set_threads(nthr)
        # grow the sample towards the full query set while it stays inside the budget
        if dt * (len(xq) / ns) < 0.5 * budget_s and ns < len(xq):
            ns = len(xq)
            t0 = time.time()
            Dr, Ir = idx.search(xq[:ns], k)
            dt = time.time() - t0
        kind, threads = "reference", Ref.max_threads()
        sample = "faiss 1.15.0 IndexFlatL2.search, one batch of the first %d of the %d queries, nb=%d, k=%d, %d OpenMP threads" % (
            ns, len(xq), len(xb), k, threads)
    else:
        ns = 8
        t0 = time.time()
        Dr, Ir = Oracle.flat_search(METRIC_L2, xb, xq[:ns], k)
        dt = time.time() - t0
        kind, threads = "port", cores
        sample = "oracle/faiss_oracle.c restatement (OpenMP over queries), first %d queries" % ns
    out = {"value": round(ns / dt, 1), "unit": "QPS", "cores": int(threads), "kind": kind, "sample": sample}
    if gpu_I is not None:
        eq = float((gpu_I[:ns] == Ir).mean())
        rel = float(np.max(np.abs(gpu_D[:ns] - Dr) / np.maximum(np.abs(Dr), 1e-30)))
        out["parity_vs_gpu"] = {"labels_equal_frac": round(eq, 6), "max_rel_dist_err": float("%.3g" % rel),
                                "recall_at_1": float((gpu_I[:ns, 0] == Ir[:, 0]).mean())}
    return out


def cpu_baseline_ivfpq(gpu_index, xb, xq, gt_first):
    """Reference CPU IndexIVFPQ (index_factory "IVF4096,PQ64", default search parameters, nprobe=32) holding
    the SAME coarse centroids and PQ codebook as the GPU index (installed through the shim, the reverse of
    GpuIndexIVFPQ::copyTo), filled with the same vectors by the reference's own add(), timed on the node's
    core allowance.  About 10-20 s of CPU work (add 1M vectors + search)."""
    from oracle.pyoracle import Ref
    if not Ref.available():
        return None
    cores = effective_cores()
    Ref.set_threads(cores)
    idx = Ref.index_factory(xb.shape[1], "IVF4096,PQ64")
    idx.set_trained(gpu_index.get_centroids(), gpu_index.get_pq_centroids())
    t0 = time.time()
    idx.add(xb)
    t_add = time.time() - t0
    idx.set_nprobe(32)
    idx.search(xq[:256], K)
    t0 = time.time()
    Dr, Ir = idx.search(xq, K)
    dt = time.time() - t0
    return {"value": round(len(xq) / dt, 1), "unit": "QPS", "cores": int(cores), "kind": "reference",
            "sample": "faiss 1.15.0 index_factory('IVF4096,PQ64') with the GPU-trained quantizers, nprobe=32, all %d "
                      "queries, nb=%d, k=%d, use_precomputed_table=%d" % (len(xq), len(xb), K,
                                                                          idx.pq_info()["use_precomputed_table"]),
            "add_s": round(t_add, 1),
            "recall_at_1": round(float((Ir[:, 0] == gt_first).mean()), 4),
            "recall_at_100": round(float((Ir == gt_first[:, None]).any(axis=1).mean()), 4)}, (Dr, Ir)


def ivfpq_leg(res, xt, xb, xq_dev, gt_first, steps, warmup, torch, with_cpu=True):
    """IVF4096,PQ64 (second half of the metric): native train + add on the GPU, nprobe=32."""
    import faiss_amd
    t0 = time.time()
    idx = faiss_amd.GpuIndexIVFPQ(res, D, 4096, 64, 8, faiss_amd.METRIC_L2)
    idx.train(xt)
    t_train = time.time() - t0
    t0 = time.time()
    idx.add(xb)
    t_add = time.time() - t0
    idx.nprobe = 32
    Dd = torch.empty((NQ, K), dtype=torch.float32, device=xq_dev.device)
    Id = torch.empty((NQ, K), dtype=torch.int64, device=xq_dev.device)
    for _ in range(max(1, warmup)):
        idx.search_ptr(NQ, xq_dev.data_ptr(), K, Dd.data_ptr(), Id.data_ptr())
    res.profile_enable(True)
    res.profile_reset()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(steps):
        idx.search_ptr(NQ, xq_dev.data_ptr(), K, Dd.data_ptr(), Id.data_ptr())
    torch.cuda.synchronize()
    dt = (time.time() - t0) / steps
    I = Id.cpu().numpy()
    scan_ms, scan_n = res.profile_get("ivfpq_fused_kernel")
    sel_ms, sel_n = res.profile_get("select_k_kernel")  # coarse quantizer's selection
    res.profile_enable(False)
    cpu = None
    if with_cpu:
        try:
            cpu, (Dr, Ir) = cpu_baseline_ivfpq(idx, xb, xq_dev.cpu().numpy(), gt_first)
            # same quantizers, same codes (up to encode near-ties): how close are the two result sets
            Dg = Dd.cpu().numpy()
            cpu["parity_vs_gpu"] = {
                "top1_label_equal_frac": round(float((I[:, 0] == Ir[:, 0]).mean()), 4),
                "top100_set_overlap": round(float(np.mean([len(np.intersect1d(a, b)) / float(K) for a, b in
                                                           zip(I[:200], Ir[:200])])), 4),
                "median_rel_dist_err_top1": float("%.3g" % np.median(np.abs(Dg[:, 0] - Dr[:, 0]) /
                                                                      np.maximum(np.abs(Dr[:, 0]), 1e-30)))}
        except Exception as e:  # noqa: BLE001
            cpu = {"error": repr(e)[:200]}
    codes_per_query = 32.0 * NB / 4096.0
    out = {
        "qps": round(NQ / dt, 1), "ms_per_step": round(dt * 1e3, 3), "nprobe": 32,
        "recall_at_1": round(float((I[:, 0] == gt_first).mean()), 4),
        "recall_at_100": round(float((I == gt_first[:, None]).any(axis=1).mean()), 4),
        "train_s": round(t_train, 2), "add_s": round(t_add, 2),
        "scan_kernel": "ivfpq_fused_kernel (per-query LUT + code scan of the probed lists as one position stream + top-k, all in LDS; 2 workgroups per CU)",
        "scan_kernel_ms": round(scan_ms / max(scan_n, 1), 3), "select_kernel_ms": round(sel_ms / max(sel_n, 1), 3),
        # algorithmic HBM bytes of the code scan: nprobe * nb/nlist * M bytes per query
        "scan_algorithmic_GBps": round(codes_per_query * 64 * NQ / (scan_ms / max(scan_n, 1) * 1e-3) / 1e9, 1)
        if scan_n else None,
        "cpu_baseline": cpu,
    }
    if cpu and cpu.get("value"):
        out["speedup_vs_cpu"] = round(out["qps"] / cpu["value"], 1)
    return out, idx


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ivfpq", action="store_true")
    ap.add_argument("--multi-gpu", choices=["replicas", "shards"], default="replicas",
                    help="N > 1: replicas = every GPU holds the database, queries are split (IndexReplicas, the "
                         "reference's default for databases that fit one GPU); shards = rows are split (IndexShards)")
    ap.add_argument("--flat-path", choices=["filter", "exact"], default="filter",
                    help="filter: fp16 MFMA candidate filter + exact fp32 re-rank (default, bit-identical results); "
                         "exact: fp32 MFMA scan only")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    import faiss_amd  # after torch: both then share one HIP runtime in this process
    from faiss_amd.datasets import synthetic_dataset
    from faiss_amd.distributed import ReplicatedSearcher, ShardedSearcher, replica_bounds, shard_bounds

    res = faiss_amd.StandardGpuResources(local_rank)
    t0 = time.time()
    xt, xb, xq = synthetic_dataset(D, NT, NB, NQ, seed=1338)
    replicas = world > 1 and args.multi_gpu == "replicas"
    bounds = [(0, NB)] * world if replicas else shard_bounds(NB, world)
    lo, hi = bounds[rank]
    index = faiss_amd.GpuIndexFlatL2(res, D)
    index.set_use_filter_kernel(args.flat_path == "filter")
    index.add(xb[lo:hi])
    if rank == 0:
        log("data + add: %.1fs (shard rows %d..%d of %d)" % (time.time() - t0, lo, hi, NB))

    xq_dev = torch.from_numpy(xq).to(dev)
    nq_loc = replica_bounds(NQ, world)[1] if replicas else NQ
    D_loc = torch.empty((nq_loc, K), dtype=torch.float32, device=dev)
    I_loc = torch.empty((nq_loc, K), dtype=torch.int64, device=dev)
    D_out = torch.empty((NQ, K), dtype=torch.float32, device=dev)
    I_out = torch.empty((NQ, K), dtype=torch.int64, device=dev)

    def local_search(_xq, k):
        index.search_ptr(NQ, xq_dev.data_ptr(), k, D_loc.data_ptr(), I_loc.data_ptr())
        return D_loc, I_loc

    def merge(all_D, all_I, base):
        if all_D.shape[0] == 1:
            return all_D[0], all_I[0]
        torch.cuda.current_stream().synchronize()  # gathered tensors complete before our stream reads them
        faiss_amd.merge_knn_results_device(res, faiss_amd.METRIC_L2, NQ, K, all_D.shape[0], all_D.data_ptr(),
                                           all_I.data_ptr(), base, D_out.data_ptr(), I_out.data_ptr())
        return D_out, I_out

    def local_search_block(qlo, qhi, k):
        if qhi > qlo:
            index.search_ptr(qhi - qlo, xq_dev.data_ptr() + qlo * D * 4, k, D_loc.data_ptr(), I_loc.data_ptr())
        return D_loc, I_loc

    if replicas:
        rep = ReplicatedSearcher(local_search_block, NQ, dev)

        class _S:  # same call shape as ShardedSearcher.search
            @staticmethod
            def search(_xq, k):
                return rep.search(k)
        searcher = _S
    else:
        searcher = ShardedSearcher(local_search, merge, [b - a for a, b in bounds], dev)

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        searcher.search(xq_dev, K)
    res.profile_enable(True)
    res.profile_reset()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = searcher.search(xq_dev, K)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    used_filter, n_overflow = index.filter_stats()
    dom = "flat_filter_kernel" if used_filter else "flat_scan_kernel"
    scan_ms, scan_n = res.profile_get(dom)
    rr_ms, rr_n = res.profile_get("flat_rerank_kernel")
    mxp_ms, mxp_n = res.profile_get("flat_filter_kernel_max")
    tg_ms, tg_n = res.profile_get("flat_tighten_kernel")
    cv_ms, cv_n = res.profile_get("convert_f16_query")
    sel_ms, sel_n = res.profile_get("select_k_kernel")
    res.profile_enable(False)

    if rank != 0:
        # leave together with rank 0 (which still assembles and prints the line)
        dist.barrier()
        dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    qps = NQ * args.steps / elapsed
    gD, gI = out[0].cpu().numpy(), out[1].cpu().numpy()
    avg_scan_ms = scan_ms / max(scan_n, 1)
    # dominant kernel: the scan of this rank's shard.  Algorithmic work per launch =
    # 2*nq*nb_shard*d flops (256 MFLOP/query at nb=1M, d=128), priced against the
    # dense MFMA peak of the unit the kernel runs on (f16 for the filter, f32 for the exact scan).
    nq_rank0 = (replica_bounds(NQ, world)[0][0][1]) if replicas else NQ  # queries of the launches timed on rank 0
    flops = 2.0 * nq_rank0 * (hi - lo) * D
    achieved = flops / (avg_scan_ms * 1e-3) / 1e12
    peak = PEAK_F16_MFMA_TFLOPS if used_filter else PEAK_F32_MFMA_TFLOPS
    # one sweep of the shard + queries + results is the algorithmic HBM traffic of the launch
    row_bytes = D * (2.0 if used_filter else 4.0)
    hbm_bytes = (hi - lo) * row_bytes + nq_rank0 * row_bytes + nq_rank0 * K * 12.0
    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process; the
    # figure is the rocprofv3 --pmc FETCH_SIZE pass committed under profiles/ (same kernel, same
    # workload, corrected x2 as the MI355X guide prescribes for 16 B/lane reads on gfx950)
    traffic = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_e_pmc_counters.json")))
        if used_filter and world == 1:
            ent = [v for k, v in pmc.items() if "flat_filter_kernel<1, 1," in k and "FETCH_SIZE" in v][0]
            traffic = {"hbm_read_bytes_per_launch": round(ent["hbm_read_bytes_corrected"]),
                       "algorithmic_bytes_per_launch": round(hbm_bytes),
                       "source": "profiles/r01_e_pmc_counters.txt (rocprofv3 --pmc FETCH_SIZE, separate pass, tools/flat_only.py; "
                                 "gfx950 2x correction for 16 B/lane reads applied)"}
    except (OSError, KeyError, ValueError, IndexError):
        pass
    line = {
        "metric": "QPS @ recall@1 (nq=10k, k=100) FlatL2, SIFT1M-shaped synthetic",
        "value": round(qps, 1), "unit": "QPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32 in/out; f16 MFMA candidate filter + f32 exact re-rank, results bit-identical to the f32 scan"
        if used_filter else "f32", "data": "synthetic",
        "config": {"workload": "GpuIndexFlatL2 d=128 nb=1M nq=10k k=100 (BASELINE.json configs[1])",
                   "generator": "SyntheticDataset(d=128, nt=100k, nb=1M, nq=10k, seed=1338)",
                   "sharding": ("single GPU" if world == 1 else
                                "IndexReplicas-style: database on every GPU, %d-query blocks, result blocks gathered to rank 0"
                                % nq_loc if replicas else
                                "IndexShards-style rows/%d per GPU, gather to rank 0 + device merge" % world),
                   "inputs": "queries/results resident in HBM",
                   "flat_path": "filter" if used_filter else "exact"},
        "roofline": {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 2),
                     "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                     "traffic": traffic, "avg_kernel_ms": round(avg_scan_ms, 3), "launches": int(scan_n),
                     "algorithmic_hbm_GBps": round(hbm_bytes / (avg_scan_ms * 1e-3) / 1e9, 1),
                     "hbm_frac": round(hbm_bytes / (avg_scan_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 5)},
        "other_kernels_ms": {"flat_filter_kernel<MODE_MAX> (chunk maxima on a 1/4 tile sample)": round(mxp_ms / max(mxp_n, 1), 3),
                             "flat_tighten_kernel": round(tg_ms / max(tg_n, 1), 3),
                             "flat_rerank_kernel": round(rr_ms / max(rr_n, 1), 3),
                             "convert_f16_query+norms": round(cv_ms / max(cv_n, 1), 3),
                             "select_k_kernel": round(sel_ms / max(sel_n, 1), 3)},
        "filter_overflow_queries": int(n_overflow),
    }
    if world == 1:
        if not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(xb, xq, K, gD, gI)
                line["recall_at_1"] = line["cpu_baseline"].get("parity_vs_gpu", {}).get("recall_at_1")
            except Exception as e:  # noqa: BLE001
                line["cpu_baseline"] = {"error": repr(e)[:200]}
        if not args.no_ivfpq:
            try:
                line["ivfpq"], _ = ivfpq_leg(res, xt, xb, xq_dev, gI[:, 0], max(1, args.steps // 2), 1, torch,
                                             with_cpu=not args.no_cpu_baseline)
            except Exception as e:  # noqa: BLE001
                line["ivfpq"] = {"error": repr(e)[:300]}
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
