#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X similarity-search backend.

Metric (BASELINE.json): QPS at nq=10 000, k=100 on SIFT1M-shaped synthetic data (d=128, nb=1M; the reference's
SyntheticDataset recipe, seed 1338).  A "step" is one search of all 10 000 queries.

  N = 1 : `value` = GpuIndexFlatL2 QPS (BASELINE.json configs[1]) with queries and results resident in HBM when the
          timed region starts; `value_host_buffers` beside it is the same search handed pageable HOST buffers (how
          benchs/bench_gpu_sift1m.py times the reference; PCIe copies inside the timed region).  The same JSON line
          carries the IVF4096,PQ64 / IVF4096,Flat / IVF4096,SQ8 legs at nb=1M (`ivfpq`, `ivfflat`, `ivfsq`), each with its
          own `roofline` block (query-major scans: HBM-bound, algorithmic bytes of SURVEY.md 8d over the HIP-event kernel
          time; the list-major scan that serves large IVFFlat batches since round 3: f32-MFMA-bound, 2*nq*nprobe*(nb/nlist)*d
          flop over the time of its two scan launches, with the SURVEY 8d byte figure beside it), the parity of every leg
          against the reference CPU index on ALL queries (every label mismatch classified as near-tie or counted as
          real) and the reference CPU path timed on this node's host cores (`cpu_baseline`).  `ivfflat_10m` and
          `ivfpq_100m` are BASELINE.json configs[2] and configs[3] (built chunk by chunk, chunks drawn on the device; a query sample bit-exact
          against the oracle run on the probed lists read back from the device; --budget-s bounds the run).
  N > 1 : one process per GPU (torch.distributed, backend "nccl" = RCCL), total work fixed => "scaling": "strong".
          Flat leg (`value`): --multi-gpu replicas (default; what the reference builds for a database that fits one
          GPU, GpuMultipleClonerOptions::shard = false -> IndexReplicas: every rank holds the 1M vectors and searches
          its block of the queries, the result blocks are gathered onto rank 0) or shards (IndexShards: rows split).
          IVFPQ leg (`ivfpq`): ALWAYS IndexShards-style, the layout the north star names for databases beyond one
          GPU (BASELINE.json configs[4]): rank 0 trains the coarse quantizer and the PQ codebook and broadcasts them
          (the only collective), rank r adds rows [r*nb/N, (r+1)*nb/N) with their global ids, every rank searches all
          queries on its shard (its own coarse quantization: identical on every rank, cheaper than shipping it), the
          per-rank top-k are gathered point-to-point onto rank 0 over xGMI and merged by the device select kernel.

Run:  python bench.py [--gpus N --steps K --warmup W]
      python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

D, NT, NB, NQ, K = 128, 100000, 1000000, 10000, 100
NLIST, NPROBE, PQ_M = 4096, 32, 64
PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32
PEAK_F16_MFMA_TFLOPS = 2500.0  # same guide: dense f16/bf16 MFMA (v_mfma_f32_32x32x16_f16)
PEAK_HBM_GBS = 8000.0
# committed rocprofv3 --pmc summaries (tools/profile_round.sh; one file per profiled search loop)
PROFILE_DIR = os.path.join(ROOT, "profiles")
PROFILE_JSON = {"flat": "r6_pmc_flat.json", "ivfpq": "r6_pmc_ivfpq_1m.json", "ivfsq": "r6_pmc_ivfsq_1m.json",
                "ivfflat": "r6_pmc_ivfflat_1m.json", "ivfflat_10m": "r6_pmc_ivfflat_10m.json",
                "ivfpq_10m": "r6_pmc_ivfpq_10m.json", "ivfpq_100m": "r6_pmc_ivfpq_100m.json"}


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


def classify_parity(D, I, Dr, Ir, rtol=1e-4):
    """Every label that differs from the reference's is either a near-tie permutation / a tie at the k-th boundary
    (reference distances within 2e-5 relative, tests/compare.py check_knn -- the north star's "ties broken
    consistently") or REAL.  Returns the counts; a real mismatch or a distance off by more than rtol is reported as
    such, never hidden in an average."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from compare import check_knn
    out = {"queries": int(I.shape[0]), "k": int(I.shape[1]), "labels_equal_frac": round(float((I == Ir).mean()), 6),
           "max_rel_dist_err": float("%.3g" % np.max(np.abs(D - Dr) / np.maximum(np.abs(Dr), 1e-30))),
           "top1_label_equal_frac": round(float((I[:, 0] == Ir[:, 0]).mean()), 6)}
    try:
        st = check_knn(D, I, Dr, Ir, rtol=rtol, max_tie_frac=1.0, name="bench")
        out.update({"near_tie_mismatches": int(st["tie_swaps"] + st["boundary_ties"]), "real_mismatches": 0,
                    "distances_within_1e-4": True})
    except AssertionError as e:  # a real mismatch: say so
        out.update({"near_tie_mismatches": None, "real_mismatches": "FAILED: " + str(e)[:160],
                    "distances_within_1e-4": False})
    return out


def cpu_baseline_flat(xb, xq, k, gpu_D, gpu_I, budget_s=25.0):
    """Reference CPU path (faiss IndexFlatL2, compiled unmodified into oracle/_ref) timed on this node's host cores
    on a bounded sample of the same queries; falls back to the scalar C restatement (kind "port") when oracle/_ref was
    not shipped.  The reference blocks the search into 4096-query x 1024-row sgemm tiles
    (faiss/utils/distances.cpp:424-511), so the sample is ONE large query batch (as benchs/bench_gpu_sift1m.py does)."""
    from oracle.pyoracle import METRIC_L2, Oracle, Ref
    cores = effective_cores()
    if Ref.available():
        os.environ.setdefault("MKL_NUM_THREADS", str(cores))
        Ref.set_threads(cores)
        idx = Ref.index_factory(xb.shape[1], "Flat")
        idx.add(xb)
        ns = min(len(xq), 2048)
        idx.search(xq[:ns], k)  # warm-up (MKL init, page faults)
        best = None
        for nthr in sorted({cores, max(1, cores // 2)}, reverse=True):  # SMT siblings rarely help sgemm
            Ref.set_threads(nthr)
            t0 = time.time()
            Dr, Ir = idx.search(xq[:ns], k)
            dt = time.time() - t0
            log("cpu baseline probe: %d threads -> %.1f QPS" % (nthr, ns / dt))
            if best is None or dt < best[0]:
                best = (dt, nthr, Dr, Ir)
        dt, nthr, Dr, Ir = best
        Ref.set_threads(nthr)
        if dt * (len(xq) / ns) < 0.5 * budget_s and ns < len(xq):  # the full query set fits the budget
            ns = len(xq)
            t0 = time.time()
            Dr, Ir = idx.search(xq[:ns], k)
            dt = time.time() - t0
        kind, threads = "reference", Ref.max_threads()
        sample = "faiss 1.15.0 IndexFlatL2.search k=%d nb=%d, one batch of the first %d of the %d queries, %d OpenMP threads" % (
            k, len(xb), ns, len(xq), threads)
    else:
        ns = 8
        t0 = time.time()
        Dr, Ir = Oracle.flat_search(METRIC_L2, xb, xq[:ns], k)
        dt = time.time() - t0
        kind, threads = "port", cores
        sample = "oracle/faiss_oracle.c restatement (OpenMP over queries), first %d queries" % ns
    out = {"value": round(ns / dt, 1), "unit": "QPS", "cores": int(threads), "cpu_model": cpu_model(), "kind": kind, "sample": sample,
           "k": int(k), "queries": int(ns)}
    if gpu_I is not None:
        out["parity_vs_gpu"] = classify_parity(gpu_D[:ns], gpu_I[:ns], Dr, Ir)
        out["parity_vs_gpu"]["recall_at_1"] = float((gpu_I[:ns, 0] == Ir[:, 0]).mean())
    return out


def time_search(index, torch, n, xq_ptr, d_ptr, i_ptr, steps, warmup):
    for _ in range(warmup):
        index.search_ptr(n, xq_ptr, K, d_ptr, i_ptr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        index.search_ptr(n, xq_ptr, K, d_ptr, i_ptr)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def time_search_each(index, torch, n, xq_ptr, d_ptr, i_ptr, steps):
    """every search between two synchronisations: (median seconds, [ms per step]).  For the legs that run a handful of
    10 ms searches: one stalled step would otherwise be the figure (seen once in the shard leg: 3 steps averaging 70 ms
    between runs of 8.4 and 8.6 ms)."""
    each = []
    for _ in range(steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        index.search_ptr(n, xq_ptr, K, d_ptr, i_ptr)
        torch.cuda.synchronize()
        each.append(time.perf_counter() - t0)
    return float(np.median(each)), [round(v * 1e3, 3) for v in each]


def ivf_leg(kind, res, xt, xb, xq, xq_dev, gt_first, steps, torch, with_cpu=True, sweep=False):
    """IVF4096,PQ64 / IVF4096,Flat at nb = 1M (BASELINE.json configs[3] / configs[2] at the metric's database size):
    native train + add on the GPU, nprobe = 32, the reference CPU index with the SAME quantizers as baseline and as
    parity reference (it is filled by its own add(); its lists are then also loaded into a second GPU index, so the
    parity comparison scans identical lists on both sides)."""
    import faiss_amd
    from oracle.pyoracle import Ref
    pq = kind == "ivfpq"
    sq = kind == "ivfsq"

    def make():
        if pq:
            return faiss_amd.GpuIndexIVFPQ(res, D, NLIST, PQ_M, 8, faiss_amd.METRIC_L2)
        if sq:
            return faiss_amd.GpuIndexIVFScalarQuantizer(res, D, NLIST, faiss_amd.ScalarQuantizer.QT_8bit, faiss_amd.METRIC_L2, True)
        return faiss_amd.GpuIndexIVFFlat(res, D, NLIST, faiss_amd.METRIC_L2)

    factory = "IVF4096,PQ64" if pq else "IVF4096,SQ8" if sq else "IVF4096,Flat"
    title = "GpuIndexIVFPQ PQ%dx8" % PQ_M if pq else "GpuIndexIVFScalarQuantizer QT_8bit" if sq else "GpuIndexIVFFlat"
    t0 = time.time()
    idx = make()
    idx.train(xt)
    t_train = time.time() - t0
    t0 = time.time()
    idx.add(xb)
    t_add = time.time() - t0
    idx.nprobe = NPROBE
    Dd = torch.empty((NQ, K), dtype=torch.float32, device=xq_dev.device)
    Id = torch.empty((NQ, K), dtype=torch.int64, device=xq_dev.device)
    time_search(idx, torch, NQ, xq_dev.data_ptr(), Dd.data_ptr(), Id.data_ptr(), 1, 1)
    # throughput from a loop without instrumentation; kernel times from a second loop with a HIP-event span around every
    # launch (an event record is a barrier packet: the ~20 of an IVF search cost these 1 ms searches ~0.1 ms each)
    dt = time_search(idx, torch, NQ, xq_dev.data_ptr(), Dd.data_ptr(), Id.data_ptr(), steps, 0)
    res.profile_enable(True)
    res.profile_reset()
    dt_spans = time_search(idx, torch, NQ, xq_dev.data_ptr(), Dd.data_ptr(), Id.data_ptr(), max(3, steps // 2), 0)
    spans = collect_spans(res)
    res.profile_enable(False)
    list_major = idx.scan_info()[1] == 2
    Dh = np.empty((NQ, K), dtype=np.float32)
    Ih = np.empty((NQ, K), dtype=np.int64)
    dt_host = time_search(idx, torch, NQ, xq.ctypes.data, Dh.ctypes.data, Ih.ctypes.data, max(2, steps // 2), 1)
    I = Id.cpu().numpy()
    row_bytes = PQ_M if pq else D if sq else D * 4
    out = {
        "workload": "%s nlist=%d nprobe=%d d=%d nb=%d nq=%d k=%d" % (title, NLIST, NPROBE, D, NB, NQ, K),
        "scan": scan_name(list_major, idx.last_scan_arith()),
        "qps": round(NQ / dt, 1), "ms_per_step": round(dt * 1e3, 3), "steps": int(steps),
        "ms_per_step_with_event_spans": round(dt_spans * 1e3, 3),
        "timing": "qps / ms_per_step: loop of searches without instrumentation; kernels_ms and roofline: a second loop with a "
                  "HIP-event span around every launch",
        "qps_host_buffers": round(NQ / dt_host, 1),
        "recall_at_1": round(float((I[:, 0] == gt_first).mean()), 4),
        "recall_at_100": round(float((I == gt_first[:, None]).any(axis=1).mean()), 4),
        "recall_gate": ("R@100 >= 0.95 (PQ64 saturates R@1 near 0.9 on this data; SURVEY.md 8d)" if pq
                        else "R@1 >= 0.95"),
        "train_s": round(t_train, 2), "add_s": round(t_add, 2),
        "roofline": ivf_roofline(spans, list_major, "ivfpq" if pq else "ivfsq" if sq else "ivfflat", NB, row_bytes),
        "kernels_ms": {k: round(v[0] / max(v[1], 1), 3) for k, v in spans.items() if v[1]},
    }
    if with_cpu and Ref.available():
        try:
            cores = effective_cores()
            Ref.set_threads(cores)
            ref = Ref.index_factory(D, factory)
            if pq:
                ref.set_trained(idx.get_centroids(), idx.get_pq_centroids())
            elif sq:
                ref.set_sq_trained(idx.get_centroids(), idx.get_trained())
            else:
                ref.set_centroids(idx.get_centroids())
            t0 = time.time()
            ref.add(xb)
            t_cadd = time.time() - t0
            ref.set_nprobe(NPROBE)
            ref.search(xq[:256], K)
            t0 = time.time()
            Dr, Ir = ref.search(xq, K)
            dtc = time.time() - t0
            cpu = {"value": round(NQ / dtc, 1), "unit": "QPS", "cores": int(cores), "cpu_model": cpu_model(), "kind": "reference",
                   "sample": "faiss 1.15.0 index_factory('%s') with the GPU-trained quantizers, nprobe=%d, all %d queries, "
                             "nb=%d, k=%d" % (factory, NPROBE, NQ, NB, K),
                   "add_s": round(t_cadd, 1),
                   "recall_at_1": round(float((Ir[:, 0] == gt_first).mean()), 4),
                   "recall_at_100": round(float((Ir == gt_first[:, None]).any(axis=1).mean()), 4)}
            # parity on identical lists: the reference's own lists copied to a GPU index (copyFrom)
            g2 = make()
            g2.copy_centroids(idx.get_centroids())
            if pq:
                g2.copy_pq_centroids(idx.get_pq_centroids())
            if sq:
                g2.copy_trained(idx.get_trained())
            sizes, codes, lids = ref.lists()
            g2.copy_lists(sizes, codes, lids)
            g2.nprobe = NPROBE
            D2, I2 = g2.search(xq, K)
            cpu["parity_vs_gpu"] = classify_parity(D2, I2, Dr, Ir)
            cpu["parity_vs_gpu"]["native_add_top1_equal_frac"] = round(float((I[:, 0] == Ir[:, 0]).mean()), 4)
            out["cpu_baseline"] = cpu
            out["speedup_vs_cpu"] = round(out["qps"] / cpu["value"], 1)
        except Exception as e:  # noqa: BLE001
            out["cpu_baseline"] = {"error": repr(e)[:300]}
            ref = None
    else:
        ref = None
    if sweep:
        try:
            out["nprobe_sweep"] = nprobe_sweep(idx, ref, xq, xq_dev, gt_first, torch, pq)
        except Exception as e:  # noqa: BLE001
            out["nprobe_sweep"] = {"error": repr(e)[:300]}
        idx.nprobe = NPROBE
    return out, idx


SWEEP_NPROBES = (1, 2, 4, 8, 16, 32, 64, 128, 256, 512)


def nprobe_sweep(idx, ref, xq, xq_dev, gt_first, torch, pq):
    """QPS @ recall over nprobe = 1 ... 512 (the operating-point sweep of benchs/bench_gpu_sift1m.py:81-89) at nb = 1M,
    k = 100, all 10 000 queries: the GPU index (queries / results in HBM, 3 timed searches per point after one warm-up)
    and -- when the compiled reference is there -- the reference CPU index holding the same quantizers (`ref`, filled by its
    own add()), one timed search of all queries per point on the effective host cores.  R@1 / R@100 against the exact
    ground truth (the Flat leg's labels)."""
    Dd = torch.empty((NQ, K), dtype=torch.float32, device=xq_dev.device)
    Id = torch.empty((NQ, K), dtype=torch.int64, device=xq_dev.device)
    rows = []
    for npb in SWEEP_NPROBES:
        idx.nprobe = npb
        dt = time_search(idx, torch, NQ, xq_dev.data_ptr(), Dd.data_ptr(), Id.data_ptr(), 3, 1)
        I = Id.cpu().numpy()
        row = {"nprobe": npb, "qps": round(NQ / dt, 1), "ms": round(dt * 1e3, 3),
               "scan": ("list-major" if idx.scan_info()[1] == 2 else "query-major") + (" (f32)" if idx.last_scan_arith() else ""),
               "recall_at_1": round(float((I[:, 0] == gt_first).mean()), 4),
               "recall_at_100": round(float((I == gt_first[:, None]).any(axis=1).mean()), 4)}
        if ref is not None:
            ref.set_nprobe(npb)
            t0 = time.time()
            _, Ir = ref.search(xq, K)
            dtc = time.time() - t0
            row.update({"cpu_qps": round(NQ / dtc, 1), "cpu_recall_at_1": round(float((Ir[:, 0] == gt_first).mean()), 4),
                        "cpu_recall_at_100": round(float((Ir == gt_first[:, None]).any(axis=1).mean()), 4),
                        "speedup_vs_cpu": round(dtc / dt, 1)})
        rows.append(row)
    if ref is not None:
        ref.set_nprobe(NPROBE)
    best = max(rows, key=lambda r: r["recall_at_1"])
    first95 = next((r for r in rows if r["recall_at_1"] >= 0.95), None)
    first95_100 = next((r for r in rows if r["recall_at_100"] >= 0.95), None)
    return {"points": rows,
            "what": "nb=1M, nq=10k, k=100; GPU: queries / results in HBM; cpu_*: faiss 1.15.0 CPU index with the same quantizers, "
                    "%d OpenMP threads" % effective_cores(),
            "best_recall_at_1": {"nprobe": best["nprobe"], "recall_at_1": best["recall_at_1"], "qps": best["qps"]},
            "smallest_nprobe_with_recall_at_1_ge_0.95": ({"nprobe": first95["nprobe"], "qps": first95["qps"],
                                                           "cpu_qps": first95.get("cpu_qps"),
                                                           "speedup_vs_cpu": first95.get("speedup_vs_cpu")} if first95 else None),
            "smallest_nprobe_with_recall_at_100_ge_0.95": ({"nprobe": first95_100["nprobe"], "qps": first95_100["qps"],
                                                             "cpu_qps": first95_100.get("cpu_qps"),
                                                             "speedup_vs_cpu": first95_100.get("speedup_vs_cpu")}
                                                            if first95_100 else None),
            "note": ("PQ64 quantisation error caps R@1 (the exact nearest neighbour is not always the nearest code): the "
                     "curve above is that cap measured; R@100 is the gate this leg claims" if pq else
                     "IVFFlat distances are exact: R@1 = the fraction of queries whose nearest neighbour lies in a probed list")}


SPAN_NAMES = ("ivf_lmf_prepare", "ivf_lm_plan", "ivf_lmf_sweep_min", "ivf_lmf_bound", "ivf_lmf_sweep_collect", "ivf_lmf_tighten", "ivf_lmf_rerank",
              "ivf_lm_scan_pass1", "ivf_lm_threshold", "ivf_lm_scan_pass2", "select_k_kernel",
              "ivfflat_fused_kernel", "ivfpq_fused_kernel", "ivfsq_fused_kernel", "ivf_finish_kernel", "flat_filter_kernel",
              "flat_filter_kernel_max", "flat_tighten_kernel", "flat_rerank_kernel", "convert_f16_query")


def scan_name(list_major, arith):
    if not list_major:
        return "query-major (ivf_fused.hip)"
    return ("list-major behind the f16 filter (ivf_lm_filter.hip): results bit-identical to the query-major scan" if arith == 0
            else "list-major on the f32 matrix pipe (ivf_listmajor.hip)")


def collect_spans(res):
    """(total ms, launches) per kernel span of the library's HIP-event profile (events on the library's own stream)"""
    return {k: res.profile_get(k) for k in SPAN_NAMES}


def ivf_roofline(spans, list_major, kind, nb, row_bytes, profile=None):
    """roofline block of an IVF leg.  Query-major scan: HBM-bound, SURVEY.md 8d's algorithmic bytes (nprobe * nb/nlist *
    bytes per entry per query) over the scan launch.  List-major scan: every list is read once per group of up to 64 of
    the queries probing it and the distances are f32 MFMA dot products, so the bound is the f32 matrix pipe: 2 * nq *
    nprobe * (nb/nlist) * d flop over the two scan launches; the 8d byte figure is reported beside it (it may exceed the
    HBM peak: those bytes are no longer moved once per query) together with the bytes a batch really has to read."""
    alg_bytes = float(NPROBE) * nb / NLIST * row_bytes * NQ
    if not list_major:
        kname = {"ivfpq": "ivfpq_fused_kernel", "ivfsq": "ivfsq_fused_kernel", "ivfflat": "ivfflat_fused_kernel"}[kind]
        ms, n = spans[kname]
        avg = ms / max(n, 1)
        ach = alg_bytes / (avg * 1e-3) / 1e9 if n else None
        return {"bound": "hbm", "kernel": kname, "achieved": round(ach, 1) if ach else None, "peak": PEAK_HBM_GBS,
                "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 4) if ach else None, "avg_kernel_ms": round(avg, 3),
                "launches": int(n), "algorithmic_bytes_per_launch": int(alg_bytes),
                "traffic": committed_traffic(kname, alg_bytes, profile or kind)}
    if spans.get("ivf_lmf_sweep_collect", (0, 0))[1]:
        return lmf_roofline(spans, kind, nb, row_bytes, profile)
    (m1, n1), (m2, n2) = spans["ivf_lm_scan_pass1"], spans["ivf_lm_scan_pass2"]
    searches = max(n2, 1)
    scan_ms = (m1 + m2) / searches  # both scan launches of one search (pass 1 may run twice when queries are redone)
    flops = 2.0 * NQ * NPROBE * (nb / float(NLIST)) * D
    ach = flops / (scan_ms * 1e-3) / 1e12
    unique = nb * float(row_bytes) + NQ * D * 4.0
    kernels = ("ivf_lm_scan_kernel (pass 1) + ivf_lm_flat_reg_kernel (pass 2)" if kind == "ivfflat"
               else "ivf_lm_scan_kernel<kind 2> (pass 1) + ivf_lm_flat_reg_kernel<8-bit codes> (pass 2)" if kind == "ivfsq"
               else "ivf_lm_pq_kernel (pass 1 + pass 2)")
    return {"bound": "mfma", "kernel": kernels + ": the scan launches of one search",
            "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4),
            "avg_kernel_ms": round(scan_ms, 3), "launches": int(n1 + n2),
            "algorithmic_flop_per_search": int(flops),
            "survey_8d_bytes": {"algorithmic_bytes_per_search": int(alg_bytes),
                                "GBps_over_the_scan_launches": round(alg_bytes / (scan_ms * 1e-3) / 1e9, 1),
                                "frac_of_hbm_peak": round(alg_bytes / (scan_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 3),
                                "note": "bytes the query-major formulation moves per search; the list-major scan reads a "
                                        "list once per group of <= 64 queries, so this figure is not bound by the HBM peak"},
            "unique_bytes_per_batch": int(unique),
            "traffic": lm_traffic_note(committed_lm_traffic(profile or kind, unique), kind)}


def lmf_roofline(spans, kind, nb, row_bytes, profile=None):
    """roofline block of a list-major leg behind the f16 filter (ivf_lm_filter.hip).  The two sweeps dominate; each reads
    every probed list once per group of up to 96 of the queries probing it: the fp16 shadow rows + their fp32 norms
    (IVFFlat: 2 d + 4 bytes per row) or the code bytes + norms (IVFPQ: M + 4 bytes per row).  Bound: the row stream (HBM);
    achieved = unique bytes of ONE sweep / the average sweep time.  The f16 matrix pipe's share is reported beside it
    (useful flop of one sweep = 2 * nq * nprobe * (nb / nlist) * d over the dense f16 MFMA peak), and the SURVEY 8d byte
    figure of the query-major formulation for comparison."""
    (m1, n1), (m2, n2) = spans["ivf_lmf_sweep_min"], spans["ivf_lmf_sweep_collect"]
    sweep_ms = m2 / max(n2, 1)  # sweep 2 reads every probed row (sweep 1 may sample the 32-row blocks of long lists)
    # (the scalar quantizer's sweeps read an fp16 copy of its centred codes: the IVFFlat kernel, the IVFFlat bytes)
    per_row = (2.0 * D + 4.0) if kind in ("ivfflat", "ivfsq") else (float(row_bytes) + 4.0)
    unique = nb * per_row
    ach = unique / (sweep_ms * 1e-3) / 1e9
    flops = 2.0 * NQ * NPROBE * (nb / float(NLIST)) * D
    alg_bytes = float(NPROBE) * nb / NLIST * row_bytes * NQ
    search_ms = sum(v[0] for k, v in spans.items() if k.startswith("ivf_lm") or k in ("select_k_kernel",)) / max(n2, 1)
    return {"bound": "hbm", "kernel": ("ivf_lmf_flat_kernel" if kind in ("ivfflat", "ivfsq") else "ivf_lmf_pq_kernel") +
                                       " in sweep 2 (collect: every probed row; sweep 1 = the same loop over a quarter of every work item's rows: granule minima)",
            "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 4),
            "avg_kernel_ms": round(sweep_ms, 3), "launches": int(n2),
            "sweep1_avg_kernel_ms": round(m1 / max(n1, 1), 3),
            "algorithmic_bytes_per_launch": int(unique),
            "bytes_per_row": per_row,
            "f16_mfma": {"useful_flop_per_sweep": int(flops), "TFLOPs": round(flops / (sweep_ms * 1e-3) / 1e12, 1),
                         "frac_of_f16_peak": round(flops / (sweep_ms * 1e-3) / 1e12 / PEAK_F16_MFMA_TFLOPS, 4)},
            "list_major_kernels_ms_per_search": round(search_ms, 3),
            "survey_8d_bytes": {"algorithmic_bytes_per_search": int(alg_bytes),
                                "note": "bytes the query-major formulation moves per search (once per query and probe)"},
            "traffic": committed_lm_traffic(profile or kind, unique, filt=True)}


def sample_vs_oracle(idx, pq, xq, sel, Dg, Ig, arith):
    """The queries `sel` of a search of an IVF index, BIT-EXACT against the oracle restatement (arith =
    idx.last_scan_arith() of the search) run on the lists they probe, read back from the device; the coarse assignment is compared too.
    Returns (exact, entries read back, (Do, Io) = the oracle's results for the sample)."""
    from oracle.pyoracle import METRIC_L2, Oracle
    cent = idx.get_centroids()
    pqc = idx.get_pq_centroids() if pq else None
    Dq, Iq = idx.quantizer_search(xq[sel], NPROBE)
    sizes = np.zeros(NLIST, dtype=np.uint32)
    codes, ids = [], []
    for l in np.unique(Iq):
        sizes[l] = idx.get_list_size(int(l))
        codes.append(idx.get_list_codes(int(l)))
        ids.append(idx.get_list_ids(int(l)))
    codes, ids = np.concatenate(codes), np.concatenate(ids)
    Do, Io, cD, cI = Oracle.ivf_search(1 if pq else 0, METRIC_L2, cent, sizes, codes, ids, xq[sel], NPROBE, K,
                                       M=PQ_M if pq else 0, pq=pqc, arith=arith)
    exact = bool(np.array_equal(cI, Iq) and np.array_equal(cD, Dq) and np.array_equal(Io, Ig[sel]) and np.array_equal(Do, Dg[sel]))
    return exact, len(ids), (Do, Io)


def scale_leg(kind, nb, res, xt, xb, xq, xq_dev, dmap, torch, leg_1m, nsample=16, with_cpu=True, cpu_budget_s=20.0):
    """BASELINE.json configs[2] (GpuIndexIVFFlat nb = 10M) / configs[3] (GpuIndexIVFPQ PQ64 nb = 100M) on one MI355X:
    the index is built chunk by chunk from the generator (never more than 1M rows on the host), all 10 000 queries are
    searched (k = 100, nprobe = 32, queries / results in HBM), `nsample` of them are checked BIT-EXACTLY against the
    oracle restatement run on the lists they probe, read back from the device; all results are checked for order and
    label validity.  `cpu_baseline` is MEASURED at this size (cpu_baseline_on_gpu_lists): a reference CPU index with the
    same quantizers is filled with the very lists the GPU index holds and searched on the node's host cores."""
    import faiss_amd
    from faiss_amd.datasets import synthetic_more, synthetic_more_device
    from oracle.pyoracle import METRIC_L2, Oracle
    pq = kind == "ivfpq"
    idx = (faiss_amd.GpuIndexIVFPQ(res, D, NLIST, PQ_M, 8, faiss_amd.METRIC_L2) if pq
           else faiss_amd.GpuIndexIVFFlat(res, D, NLIST, faiss_amd.METRIC_L2))
    t0 = time.time()
    idx.train(xt)
    t_add = t_gen = 0.0
    done = chunk = 0
    # Chunks 1.. are drawn ON THE DEVICE (faiss_amd.datasets.synthetic_more_device: the SyntheticDataset map in fp64 on fresh
    # latent draws of a CUDA generator seeded 1338 + chunk) and handed to add() as device pointers: the host recipe costs
    # 2 s per million rows (205 s of a 100M build; nothing of it is measured work).  FAISS_AMD_BENCH_HOST_GEN=1 restores
    # the host generator.
    host_gen = os.environ.get("FAISS_AMD_BENCH_HOST_GEN") == "1"
    dev = xq_dev.device
    while done < nb:
        n_c = len(xb) if chunk == 0 else min(1000000, nb - done)
        t1 = time.time()
        if chunk == 0:
            xbc = xb
        elif host_gen:
            xbc = synthetic_more(dmap, n_c, seed=1338 + chunk)
        else:
            xbc = synthetic_more_device(dmap, n_c, 1338 + chunk, dev)
        t_gen += time.time() - t1
        t1 = time.time()
        if isinstance(xbc, np.ndarray):
            idx.add(xbc)
        else:
            idx.add_ptr(n_c, xbc.data_ptr())
        t_add += time.time() - t1
        done += n_c
        chunk += 1
        del xbc
    t_build = time.time() - t0
    idx.nprobe = NPROBE
    Dd = torch.empty((NQ, K), dtype=torch.float32, device=xq_dev.device)
    Id = torch.empty((NQ, K), dtype=torch.int64, device=xq_dev.device)
    time_search(idx, torch, NQ, xq_dev.data_ptr(), Dd.data_ptr(), Id.data_ptr(), 1, 1)
    steps = 9 if nb <= 20000000 else 7
    dt, step_ms = time_search_each(idx, torch, NQ, xq_dev.data_ptr(), Dd.data_ptr(), Id.data_ptr(), steps)  # no instrumentation
    res.profile_enable(True)
    res.profile_reset()
    dt_spans = time_search(idx, torch, NQ, xq_dev.data_ptr(), Dd.data_ptr(), Id.data_ptr(), 2, 0)
    spans = collect_spans(res)
    res.profile_enable(False)
    list_major = idx.scan_info()[1] == 2
    arith = idx.last_scan_arith()
    Dg, Ig = Dd.cpu().numpy(), Id.cpu().numpy()
    ok_order = bool((np.diff(Dg, axis=1) >= 0).all() and (Ig >= 0).all() and (Ig < nb).all())
    # ---- sample parity: oracle on the probed lists read back from the device
    sel = np.random.RandomState(3).choice(NQ, nsample, replace=False)
    exact, nread, _ = sample_vs_oracle(idx, pq, xq, sel, Dg, Ig, arith)
    row_bytes = PQ_M if pq else D * 4
    used, holes, alloc = idx.arena_stats()
    out = {
        "workload": "%s nlist=%d nprobe=%d d=%d nb=%d nq=%d k=%d (BASELINE.json configs[%d])" % (
            "GpuIndexIVFPQ PQ%dx8" % PQ_M if pq else "GpuIndexIVFFlat", NLIST, NPROBE, D, nb, NQ, K, 3 if pq else 2),
        "scan": scan_name(list_major, arith),
        "qps": round(NQ / dt, 1), "ms_per_step": round(dt * 1e3, 3), "steps": steps,
        "step_ms": step_ms, "timing": "every step between two synchronisations; ms_per_step / qps = the MEDIAN step",
        "ms_per_step_with_event_spans": round(dt_spans * 1e3, 3),
        "generator": ("chunk 0 = the flat leg's 1M database; chunks 1.. " +
                      ("synthetic_more(seed 1338 + chunk) on the host" if host_gen else
                       "drawn on the device (torch CUDA generator seeded 1338 + chunk through the SyntheticDataset map in fp64) "
                       "and added from device buffers")),
        "build_s": round(t_build, 1), "add_s": round(t_add, 1), "data_generation_s": round(t_gen, 1),
        "add_M_vectors_per_s": round(nb / t_add / 1e6, 2),
        "arena_rows_over_vectors": round(alloc / float(nb), 3), "overflow_queries": int(idx.scan_info()[2]),
        "roofline": ivf_roofline(spans, list_major, kind, nb, row_bytes, profile="%s_%dm" % (kind, nb // 1000000)),
        "kernels_ms": {k: round(v[0] / max(v[1], 1), 3) for k, v in spans.items() if v[1]},
        "parity": {"sampled_queries_bit_exact_vs_oracle_on_probed_lists": exact, "sampled_queries": int(nsample),
                   "probed_list_entries_read_back": int(nread), "all_results_ordered_and_labels_valid": ok_order},
    }
    if with_cpu:
        try:
            out["cpu_baseline"] = cpu_baseline_on_gpu_lists(kind, idx, nb, xq, Dg, Ig, leg_1m, cpu_budget_s)
            if isinstance(out["cpu_baseline"].get("value"), (int, float)):
                out["speedup_vs_cpu"] = round(out["qps"] / out["cpu_baseline"]["value"], 1)
        except Exception as e:  # noqa: BLE001
            out["cpu_baseline"] = {"error": repr(e)[:300]}
    else:
        out["cpu_baseline"] = {"not_measured": "--no-cpu-baseline"}
    del idx
    return out


def cpu_baseline_on_gpu_lists(kind, idx, nb, xq, Dg, Ig, leg_1m, budget_s):
    """The reference CPU index (faiss 1.15.0 IndexIVFFlat / IndexIVFPQ, compiled unmodified into oracle/_ref) AT THE SCALE
    of the leg: it gets the GPU index's quantizers and the GPU index's inverted lists (read back list by list and appended
    with InvertedLists::add_entries -- what GpuIndexIVF::copyTo does, faiss/gpu/impl/IVFBase.cu:328-344), so both sides
    scan identical lists; no CPU add() of 10M-100M vectors is paid.  Timed: IndexIVF::search of a bounded query sample
    (sized from the nb = 1M figure of this run so that it takes ~budget_s; the figure printed is measured, never
    scaled), on the effective host cores.  The sample's results are compared with the GPU's (classify_parity)."""
    from oracle.pyoracle import Ref
    if not Ref.available():
        return {"not_measured": "oracle/_ref (the compiled reference) was not shipped"}
    pq = kind == "ivfpq"
    cores = effective_cores()
    Ref.set_threads(cores)
    t0 = time.time()
    ref = Ref.index_factory(D, "IVF%d,PQ%d" % (NLIST, PQ_M) if pq else "IVF%d,Flat" % NLIST)
    if pq:
        ref.set_trained(idx.get_centroids(), idx.get_pq_centroids())
    else:
        ref.set_centroids(idx.get_centroids())
    for l in range(NLIST):
        ref.add_list_entries(l, idx.get_list_ids(l), idx.get_list_codes(l))
    t_fill = time.time() - t0
    assert ref.ntotal == nb, (ref.ntotal, nb)
    ref.set_nprobe(NPROBE)
    qps_1m = ((leg_1m or {}).get("cpu_baseline") or {}).get("value")
    est_qps = qps_1m * NB / float(nb) if isinstance(qps_1m, (int, float)) else 200.0  # only sizes the sample
    ns = int(max(64, min(NQ, budget_s * est_qps)))
    ref.search(xq[:min(ns, 64)], K)  # warm-up
    t0 = time.time()
    Dr, Ir = ref.search(xq[:ns], K)
    dt = time.time() - t0
    out = {"value": round(ns / dt, 1), "unit": "QPS", "cores": int(cores), "cpu_model": cpu_model(), "kind": "reference",
           "sample": "faiss 1.15.0 %s.search MEASURED at nb=%d: the first %d of the %d queries in one batch (%.1f s), nprobe=%d, "
                     "k=%d, %d OpenMP threads; the index holds the GPU index's quantizers and its inverted lists (read back "
                     "and appended with add_entries in %.1f s: no CPU add at this size)"
                     % ("IndexIVFPQ" if pq else "IndexIVFFlat", nb, ns, NQ, dt, NPROBE, K, cores, t_fill),
           "search_s": round(dt, 2), "queries": int(ns), "fill_s": round(t_fill, 1),
           "parity_vs_gpu": classify_parity(Dg[:ns], Ig[:ns], Dr, Ir)}
    del ref
    return out


def predicted_per_rank(res, flat_index, ivfpq_index, xb, xq_dev, torch):
    """What ONE rank of an N-GPU run of this bench would execute, timed on this single GPU: the Flat leg replicates the
    database and splits the queries (rank block = nq / N queries, rounded up to 128), the IVFPQ leg shards the rows
    (nb / N rows per rank, shared quantizers, all queries).  A PREDICTION of the per-rank compute step, not a scaling
    measurement: no gather, no merge, no second GPU (no multi-GPU node was available to the builder in rounds 1-3)."""
    import faiss_amd
    from faiss_amd.distributed import replica_bounds
    out = {"what": "per-rank compute step of an N-GPU run, timed on ONE GPU: a prediction, not a scaling measurement "
                   "(no gather / merge / second GPU involved)", "flat_replicas_query_block": {}, "ivfpq_row_shard": {}}
    Dd = torch.empty((NQ, K), dtype=torch.float32, device=xq_dev.device)
    Id = torch.empty((NQ, K), dtype=torch.int64, device=xq_dev.device)
    for n in (2, 4, 8):
        per = replica_bounds(NQ, n)[1]
        dt = time_search(flat_index, torch, per, xq_dev.data_ptr(), Dd.data_ptr(), Id.data_ptr(), 5, 2)
        out["flat_replicas_query_block"][str(n)] = {"queries": per, "ms": round(dt * 1e3, 3),
                                                    "implied_aggregate_qps_before_gather": round(NQ / dt, 0)}
    if ivfpq_index is not None:
        cent, pqc = ivfpq_index.get_centroids(), ivfpq_index.get_pq_centroids()
        for n in (2, 4, 8):
            rows = NB // n
            sh = faiss_amd.GpuIndexIVFPQ(res, D, NLIST, PQ_M, 8, faiss_amd.METRIC_L2)
            sh.copy_centroids(cent)
            sh.copy_pq_centroids(pqc)
            sh.add_with_ids(xb[:rows], np.arange(rows, dtype=np.int64))
            sh.nprobe = NPROBE
            dt = time_search(sh, torch, NQ, xq_dev.data_ptr(), Dd.data_ptr(), Id.data_ptr(), 5, 2)
            out["ivfpq_row_shard"][str(n)] = {"rows": rows, "ms": round(dt * 1e3, 3),
                                              "implied_aggregate_qps_before_merge": round(NQ / dt, 0)}
            del sh
    return out


def _pmc(profile):
    path = os.path.join(PROFILE_DIR, PROFILE_JSON.get(profile, ""))
    return json.load(open(path)), os.path.relpath(path, ROOT)


def committed_traffic(kernel_substr, alg_bytes, profile):
    """HBM bytes per launch of a kernel from the rocprofv3 --pmc FETCH_SIZE pass committed under profiles/ (PMC
    counters cannot be read from inside this process; same kernel, same workload, corrected x2 as the MI355X guide
    prescribes for 16 B/lane reads on gfx950).  None when the summary holds no entry."""
    try:
        pmc, rel = _pmc(profile)
        ent = [v for k, v in pmc.items() if kernel_substr in k and "FETCH_SIZE" in v]
        if not ent:
            return None
        e = max(ent, key=lambda v: v.get("avg_duration_ns", 0))
        return {"hbm_read_bytes_per_launch_from_committed_profile": round(e["hbm_read_bytes_corrected"]),
                "algorithmic_bytes_per_launch": round(alg_bytes),
                "source": rel + " (rocprofv3 --pmc FETCH_SIZE, separate pass; gfx950 2x correction for 16 B/lane reads applied)"}
    except (OSError, KeyError, ValueError, IndexError):
        return None


def committed_lm_traffic(profile, unique_bytes, filt=False):
    """L2-miss bytes of the list-major scan launches of ONE search (pass 1 + pass 2 kernels) from the committed FETCH_SIZE
    pass of the same workload, next to the bytes a batch has to read at least once."""
    try:
        pmc, rel = _pmc(profile)
        if filt:  # the sweeps of ivf_lm_filter.hip (only profiles taken since they exist hold them)
            ent = {k: v for k, v in pmc.items() if "ivf_lmf_" in k and "_kernel<" in k and "FETCH_SIZE" in v}
        else:
            ent = {k: v for k, v in pmc.items() if "ivf_lm_" in k and "_kernel<" in k and "FETCH_SIZE" in v
                   and ("scan_kernel" in k or "reg_kernel" in k or "pq_kernel" in k)}
        if not ent:
            return None
        return {"hbm_read_bytes_per_search_from_committed_profile": round(sum(v["hbm_read_bytes_corrected"] for v in ent.values())),
                "unique_bytes_per_batch": round(unique_bytes),
                "kernels": {k.split("(")[0].replace("void faiss_amd::", ""): {"bytes": round(v["hbm_read_bytes_corrected"]),
                                                                              "avg_ms": round(v["avg_duration_ns"] / 1e6, 3)}
                            for k, v in ent.items()},
                "source": rel + " (rocprofv3 --pmc FETCH_SIZE = L2 misses, separate pass; gfx950 2x correction for 16 B/lane reads applied)"}
    except (OSError, KeyError, ValueError, IndexError):
        return None


def lm_traffic_note(t, kind):
    if t and kind == "ivfsq":
        t["note"] = ("the scalar quantizer's kernels read 4 bytes per lane and load: the x2 correction (prescribed for 16 B/lane "
                     "reads) makes these figures an upper bound, the raw FETCH_SIZE is half of them")
    return t


def committed_mfma_busy(kernel_substr, profile="flat"):
    """MFMA-busy share of the dominant flat kernel from the committed SQ pass: SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed
    over the 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs = active cycles of the launch) / 1024."""
    try:
        pmc, _ = _pmc(profile)
        ent = [v for k, v in pmc.items() if kernel_substr in k and "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v]
        e = max(ent, key=lambda v: v.get("avg_duration_ns", 0))
        return round(e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 4)
    except (OSError, KeyError, ValueError, IndexError):
        return None


def sharded_ivfpq_leg(res, rank, world, dev, xt, xb, xq_dev, steps, torch, dist):
    """IndexShards over the ranks for IVF4096,PQ64 (docstring of this file, N > 1)."""
    import faiss_amd
    from faiss_amd.distributed import ShardedSearcher, shard_bounds
    idx = faiss_amd.GpuIndexIVFPQ(res, D, NLIST, PQ_M, 8, faiss_amd.METRIC_L2)
    cent = torch.empty((NLIST, D), dtype=torch.float32, device=dev)
    pqc = torch.empty((PQ_M, 256, D // PQ_M), dtype=torch.float32, device=dev)
    t0 = time.time()
    if rank == 0:
        idx.train(xt)
        cent.copy_(torch.from_numpy(idx.get_centroids()))
        pqc.copy_(torch.from_numpy(idx.get_pq_centroids()))
    dist.broadcast(cent, 0)
    dist.broadcast(pqc, 0)
    if rank != 0:
        idx.copy_centroids(cent.cpu().numpy())
        idx.copy_pq_centroids(pqc.cpu().numpy())
    t_train = time.time() - t0
    lo, hi = shard_bounds(NB, world)[rank]
    idx.add_with_ids(xb[lo:hi], np.arange(lo, hi, dtype=np.int64))  # global ids: no translation at the merge
    idx.nprobe = NPROBE
    D_loc = torch.empty((NQ, K), dtype=torch.float32, device=dev)
    I_loc = torch.empty((NQ, K), dtype=torch.int64, device=dev)
    D_out = torch.empty((NQ, K), dtype=torch.float32, device=dev)
    I_out = torch.empty((NQ, K), dtype=torch.int64, device=dev)

    def local_search(_xq, k):
        idx.search_ptr(NQ, xq_dev.data_ptr(), k, D_loc.data_ptr(), I_loc.data_ptr())
        return D_loc, I_loc

    def merge(all_D, all_I, _base):
        torch.cuda.current_stream().synchronize()  # gathered tensors complete before the library's stream reads them
        faiss_amd.merge_knn_results_device(res, faiss_amd.METRIC_L2, NQ, K, all_D.shape[0], all_D.data_ptr(),
                                           all_I.data_ptr(), None, D_out.data_ptr(), I_out.data_ptr())
        return D_out, I_out

    s = ShardedSearcher(local_search, merge, [0] * world, dev)
    s.search(xq_dev, K)
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = s.search(xq_dev, K)
    torch.cuda.synchronize()
    dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    dt = float(el.item()) / steps
    info = {"qps": round(NQ / dt, 1), "ms_per_step": round(dt * 1e3, 3), "train_broadcast_s": round(t_train, 2),
            "sharding": "IndexShards-style: rows [r*nb/N, (r+1)*nb/N) per rank with global ids, quantizers trained on "
                        "rank 0 and broadcast, per-rank top-k gathered to rank 0 + device merge",
            "rows_per_rank": hi - lo}
    return info, (out if rank == 0 else None)


def sharded_sample_check(local_exact, Do, Io, merged_D, merged_I, metric, dist_mod, rank, world):
    """Parity of a sharded search on a query sample, without moving the shards: every rank has compared ITS local top-k
    bit-exactly with the oracle restatement run on ITS lists (`local_exact`, oracle results Do / Io with global ids);
    top-k of a union = top-k of the per-part top-k's, so the oracle's answer for the union of the shards is the k-way
    merge of the per-rank oracle results under (distance, id) (Oracle.merge_shards, the rule of the device merge kernel
    and of faiss/utils/Heap.cpp:166-240 merge_knn_results).  Rank 0 compares the merged GPU result with it.
    Returns on rank 0: {"per_shard_bit_exact": [...], "merged_bit_exact": bool}; None elsewhere."""
    from oracle.pyoracle import Oracle
    if world > 1:
        box = [None] * world
        dist_mod.all_gather_object(box, (bool(local_exact), Do, Io))
    else:
        box = [(bool(local_exact), Do, Io)]
    if rank != 0:
        return None
    aD = np.stack([b[1] for b in box])
    aI = np.stack([b[2] for b in box])
    Dm, Im = Oracle.merge_shards(metric, aD, aI)
    return {"per_shard_bit_exact": [b[0] for b in box],
            "merged_bit_exact": bool(np.array_equal(Dm, merged_D) and np.array_equal(Im, merged_I))}


def sharded_scale_leg(res, rank, world, dev, xt, xb, xq, xq_dev, dmap, rows_per_rank, steps, torch, dist, nsample):
    """BASELINE.json configs[4]: IndexShards over the ranks for IVF4096,PQ64 at nb = world x rows_per_rank (8 x 125M = 1B).
    Rank 0 trains the coarse quantizer and the PQ codebook, one broadcast ships them (the only collective); rank r draws
    ITS rows on ITS device (chunk seeds = 1338 + global chunk number: the union of the shards is one database whatever
    the world size) and adds them with their GLOBAL ids; every rank searches all 10 000 queries on its shard (nprobe 32,
    k 100); the per-rank top-k are gathered point-to-point onto rank 0 and merged by the device select kernel -- gather
    and merge INSIDE the timed region (faiss/IndexShards.cpp:196-265, gpu/GpuCloner.cpp:368-391).  Per-GPU work is fixed
    as N grows: weak scaling.  Parity: sharded_sample_check on `nsample` queries."""
    import faiss_amd
    from faiss_amd.datasets import synthetic_more_device
    from faiss_amd.distributed import ShardedSearcher, broadcast_arrays, shard_chunks
    idx = faiss_amd.GpuIndexIVFPQ(res, D, NLIST, PQ_M, 8, faiss_amd.METRIC_L2)
    t0 = time.time()
    if rank == 0:
        idx.train(xt)
        cent, pqc = idx.get_centroids(), idx.get_pq_centroids()
    else:
        cent = np.empty((NLIST, D), dtype=np.float32)
        pqc = np.empty((PQ_M, 256, D // PQ_M), dtype=np.float32)
    cent, pqc = broadcast_arrays([cent, pqc], dev)
    if rank != 0:
        idx.copy_centroids(cent)
        idx.copy_pq_centroids(pqc)
    t_train = time.time() - t0
    t0 = time.time()
    for gchunk, id0, n_c in shard_chunks(rows_per_rank, rank):
        if gchunk == 0:
            x0 = np.ascontiguousarray(xb[:n_c])  # global chunk 0 = the flat leg's database, like the single-GPU scale legs
            idx.add_with_ids(x0, np.arange(id0, id0 + n_c, dtype=np.int64))
        else:
            xbc = synthetic_more_device(dmap, n_c, 1338 + gchunk, dev)
            idx.add_with_ids_ptr(n_c, xbc.data_ptr(), np.arange(id0, id0 + n_c, dtype=np.int64))
            del xbc
    t_build = time.time() - t0
    idx.nprobe = NPROBE
    D_loc = torch.empty((NQ, K), dtype=torch.float32, device=dev)
    I_loc = torch.empty((NQ, K), dtype=torch.int64, device=dev)
    D_out = torch.empty((NQ, K), dtype=torch.float32, device=dev)
    I_out = torch.empty((NQ, K), dtype=torch.int64, device=dev)

    def local_search(_xq, k):
        idx.search_ptr(NQ, xq_dev.data_ptr(), k, D_loc.data_ptr(), I_loc.data_ptr())
        return D_loc, I_loc

    def merge(all_D, all_I, _base):
        if all_D.shape[0] == 1:
            return all_D[0], all_I[0]
        torch.cuda.current_stream().synchronize()  # gathered tensors complete before the library's stream reads them
        faiss_amd.merge_knn_results_device(res, faiss_amd.METRIC_L2, NQ, K, all_D.shape[0], all_D.data_ptr(),
                                           all_I.data_ptr(), None, D_out.data_ptr(), I_out.data_ptr())
        return D_out, I_out

    s = ShardedSearcher(local_search, merge, [0] * world, dev)
    for _ in range(2):
        s.search(xq_dev, K)
    # every step timed on its own (barrier + synchronize on both sides, max over ranks): with a handful of steps of ~10 ms
    # one stalled step (seen once: 3 steps averaging 70 ms between runs of 8.4 and 8.6 ms) would otherwise be the figure
    step_s = []
    for _ in range(steps):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = s.search(xq_dev, K)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        step_s.append(time.perf_counter() - t0)
    el = torch.tensor(step_s, dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    step_ms = [round(float(v) * 1e3, 3) for v in el.tolist()]
    dt = float(np.median(el.cpu().numpy()))
    # kernel spans of this rank's local search (a second, instrumented loop)
    res.profile_enable(True)
    res.profile_reset()
    for _ in range(2):
        local_search(None, K)
    torch.cuda.synchronize()
    spans = collect_spans(res)
    res.profile_enable(False)
    list_major = idx.scan_info()[1] == 2
    arith = idx.last_scan_arith()
    # ---- parity on a query sample (every rank: its shard vs the oracle; rank 0: the merge)
    sel = np.random.RandomState(5).choice(NQ, nsample, replace=False)
    Dl, Il = D_loc.cpu().numpy(), I_loc.cpu().numpy()
    local_exact, nread, (Do, Io) = sample_vs_oracle(idx, True, xq, sel, Dl, Il, arith)
    merged = (out[0].cpu().numpy()[sel], out[1].cpu().numpy()[sel]) if rank == 0 else (None, None)
    par = sharded_sample_check(local_exact, Do, Io, merged[0], merged[1], faiss_amd.METRIC_L2, dist, rank, world)
    ovf = int(idx.scan_info()[2])
    del idx
    if rank != 0:
        return None
    nb_total = rows_per_rank * world
    Ig = out[1].cpu().numpy()
    Dg = out[0].cpu().numpy()
    par.update({"sampled_queries": int(nsample), "probed_list_entries_read_back_on_rank0": int(nread),
                "all_results_ordered_and_labels_valid": bool((np.diff(Dg, axis=1) >= 0).all() and (Ig >= 0).all()
                                                             and (Ig < nb_total).all())})
    return {
        "workload": "IndexShards over %d x MI355X: GpuIndexIVFPQ PQ%dx8 nlist=%d nprobe=%d d=%d nb=%d (%d rows per GPU) nq=%d "
                    "k=%d (BASELINE.json configs[4] at %d of its 8 shards)" % (world, PQ_M, NLIST, NPROBE, D, nb_total,
                                                                              rows_per_rank, NQ, K, world),
        "scaling": "weak (rows per GPU fixed; all queries to every shard)",
        "qps": round(NQ / dt, 1), "ms_per_step": round(dt * 1e3, 3), "steps": int(steps),
        "step_ms": step_ms, "ms_per_step_mean": round(float(np.mean(step_ms)), 3),
        "timed_region": "local search of all queries on every rank + point-to-point gather of the per-rank top-k onto rank 0 "
                        "+ device merge; every step between barrier + synchronize, max over ranks; ms_per_step / qps = the "
                        "MEDIAN step (step_ms lists all of them)",
        "scan": scan_name(list_major, arith),
        "train_broadcast_s": round(t_train, 2), "build_s": round(t_build, 1),
        "add_M_vectors_per_s_per_gpu": round(rows_per_rank / t_build / 1e6, 2), "overflow_queries": ovf,
        "roofline": _shard_roofline(ivf_roofline(spans, list_major, "ivfpq", rows_per_rank, PQ_M, profile="ivfpq_100m"),
                                    rows_per_rank),
        "kernels_ms_rank0": {k: round(v[0] / max(v[1], 1), 3) for k, v in spans.items() if v[1]},
        "parity": par,
    }


def _shard_roofline(r, rows_per_rank):
    """the shard leg has no PMC pass of its own: its traffic block is the committed nb = 100M pass of the same kernels, and
    says so (the bytes scale with the rows)"""
    t = r.get("traffic") if isinstance(r, dict) else None
    if t:
        t["note"] = ("counters of the nb = 100 000 000 pass of the same kernels (profiles/*_pmc_ivfpq_100m); this leg holds "
                     "%d rows per rank: scale the byte figures by %.2f" % (rows_per_rank, rows_per_rank / 1e8))
    return r



# ---------------------------------------------------------------------------------------------------------------------
# The ONE line the driver parses.  Everything measured lands in the `detail` dict (bench_detail.json, written next to this
# file and under gpurun_out/ when that exists); the printed line is the compact view of it: the contract's top-level keys
# for the headline Flat leg + one short record per BASELINE.json config.  Bounded: tests/test_bench_line_cpu.py asserts
# < 4096 bytes on a canned full record (round 4's 25 KB line could not be parsed by the driver).
MAX_LINE_BYTES = 4096
DETAIL_NAME = "bench_detail.json"


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return " ".join(ln.split(":", 1)[1].split())[:48]
    except OSError:
        pass
    return "unknown"


def _num(v, nd=4):
    return round(v, nd) if isinstance(v, float) else v


def _traffic_bytes(t):
    """HBM bytes of the dominant launch from a detail traffic block (per launch for the query-major / flat kernels; for the
    list-major legs the block lists the sweeps: the largest entry is sweep 2 = the kernel the roofline prices)"""
    if not isinstance(t, dict):
        return None, None
    if "hbm_read_bytes_per_launch_from_committed_profile" in t:
        return t["hbm_read_bytes_per_launch_from_committed_profile"], None
    ks = t.get("kernels") or {}
    per_launch = max((v.get("bytes", 0) for v in ks.values()), default=None)
    return per_launch, t.get("hbm_read_bytes_per_search_from_committed_profile")


def compact_roofline(r):
    if not isinstance(r, dict) or "bound" not in r:
        return None
    per_launch, per_search = _traffic_bytes(r.get("traffic"))
    out = {"bound": r["bound"], "kernel": str(r.get("kernel", "")).split(" in sweep")[0].split(":")[0][:60],
           "achieved": r.get("achieved"), "peak": r.get("peak"), "unit": r.get("unit"), "frac": r.get("frac"),
           "avg_kernel_ms": r.get("avg_kernel_ms"), "traffic": per_launch}
    alg = r.get("algorithmic_bytes_per_launch")
    if per_launch and alg:
        out["traffic_ratio"] = round(per_launch / float(alg), 2)
    if per_search and alg:
        out["search_traffic_ratio"] = round(per_search / float(alg), 2)
    return out


def compact_leg(name, leg):
    """one short record per BASELINE config: throughput, the dominant kernel's roofline fraction, parity, CPU baseline"""
    if not isinstance(leg, dict):
        return None
    if "error" in leg or "skipped" in leg:
        return {k: str(leg[k])[:120] for k in ("error", "skipped") if k in leg}
    wl = str(leg.get("workload", name)).split(" (BASELINE")[0].split(" (1250")[0]
    for a, b in ((" nq=%d k=%d" % (NQ, K), ""), (" d=%d" % D, ""), ("GpuIndex", "")):
        wl = wl.replace(a, b)
    wl = re.sub(r"nb=(\d+)000000\b", r"nb=\1M", wl)
    out = {"workload": wl[:80], "qps": leg.get("qps"), "ms_per_step": leg.get("ms_per_step"),
           "steps": leg.get("steps"),
           "scan": "lm-f16" if "f16 filter" in str(leg.get("scan")) else
                   "lm-f32" if "list-major" in str(leg.get("scan")) else "qm"}
    r = compact_roofline(leg.get("roofline"))
    if r:
        out.update({"bound": r["bound"], "frac": r["frac"], "kernel_ms": r["avg_kernel_ms"]})
        for k in ("traffic_ratio", "search_traffic_ratio"):
            if k in r and name != "ivfpq_shards":  # the shard leg borrows the nb = 100M counters: not its own ratio
                out[k] = r[k]
    for k in ("recall_at_1", "recall_at_100", "speedup_vs_cpu", "ranks_seen", "devices"):
        if leg.get(k) is not None:
            out[k] = leg[k]
    c = leg.get("cpu_baseline")
    if isinstance(c, dict) and "value" in c:
        out["cpu_qps"] = c["value"]
        if c.get("queries", NQ) != NQ:
            out["cpu_queries"] = c["queries"]  # (the CPU baseline ran a bounded sample of the queries)
        pv = c.get("parity_vs_gpu") or {}
        if "real_mismatches" in pv:
            out["real_mismatches"] = pv["real_mismatches"] if isinstance(pv["real_mismatches"], int) else "FAILED"
    par = leg.get("parity")
    if isinstance(par, dict):
        out["oracle_sample_bit_exact"] = bool(par.get("sampled_queries_bit_exact_vs_oracle_on_probed_lists",
                                                      par.get("merged_bit_exact") and all(par.get("per_shard_bit_exact", [False]))))
    return {k: v for k, v in out.items() if v is not None}


LEG_NAMES = ("ivfpq", "ivfflat", "ivfsq", "ivfflat_10m", "ivfpq_100m", "ivfpq_shards")


def compact_line(detail, detail_path=DETAIL_NAME):
    """the printed line: contract keys of the headline leg + `legs` (compact_leg of every leg present) + the path of the
    full record.  Always < MAX_LINE_BYTES: optional fields are dropped in a fixed order if a value ever pushes it over."""
    top = {k: detail.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                      "scaling", "vs_baseline", "dtype", "data")}
    cfg = detail.get("config") or {}
    top["config"] = {k: str(cfg[k])[:110] for k in ("workload", "generator", "sharding", "flat_path") if k in cfg}
    r = detail.get("roofline") or {}
    rc = compact_roofline(r) or {}
    for k in ("launches", "whole_search_frac", "hbm_frac"):
        if r.get(k) is not None:
            rc[k] = r[k]
    if r.get("mfma_busy_frac_from_committed_profile") is not None:
        rc["mfma_busy"] = r["mfma_busy_frac_from_committed_profile"]
    if r.get("bound") == "mfma":
        rc["note"] = "MFMA-bound: north star's HBM gate does not bind (hbm_frac)"
    top["roofline"] = rc
    c = detail.get("cpu_baseline")
    if isinstance(c, dict):
        top["cpu_baseline"] = {k: (str(c[k])[:160] if k in ("sample", "error", "cpu_model") else c[k])
                               for k in ("value", "unit", "cores", "cpu_model", "kind", "sample", "k", "queries", "error") if k in c}
        pv = c.get("parity_vs_gpu") or {}
        if "real_mismatches" in pv:
            top["parity_vs_cpu_reference"] = {"queries": pv.get("queries"),
                                              "real_mismatches": pv["real_mismatches"] if isinstance(pv["real_mismatches"], int) else "FAILED",
                                              "near_tie_mismatches": pv.get("near_tie_mismatches"),
                                              "max_rel_dist_err": pv.get("max_rel_dist_err")}
    for k in ("recall_at_1", "value_host_buffers", "filter_overflow_queries", "ranks_seen", "devices", "ranks_seen_via"):
        if detail.get(k) is not None:
            top[k] = detail[k]
    legs = {}
    for name in LEG_NAMES:
        if name in detail:
            legs[name] = compact_leg(name, detail[name])
    if legs:
        top["legs"] = legs
    top["detail"] = detail_path
    drop_order = ("cpu_queries", "steps", "kernel_ms", "recall_at_100", "scan", "workload")
    s = json.dumps(top, separators=(",", ":"))
    for key in drop_order:
        if len(s) < MAX_LINE_BYTES:
            break
        for leg in legs.values():
            if isinstance(leg, dict):
                leg.pop(key, None)
        s = json.dumps(top, separators=(",", ":"))
    if len(s) >= MAX_LINE_BYTES:  # last resort: the headline leg alone
        top.pop("legs", None)
        s = json.dumps(top, separators=(",", ":"))
    assert len(s) < MAX_LINE_BYTES, len(s)
    return s


def write_detail(detail):
    """the full record (nprobe sweeps, per-kernel spans, prose) next to this file, and under gpurun_out/ when run through
    gpurun (that directory is what travels back).  Returns the repo-relative path named in the printed line."""
    txt = json.dumps(detail, indent=1)
    rel = DETAIL_NAME
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, DETAIL_NAME), "w") as f:
                    f.write(txt + "\n")
            except OSError:
                rel = None if d == ROOT else rel
    return rel or "not written"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ivf-legs", default="ivfpq,ivfflat,ivfsq", help="comma-separated subset of the IVF legs to run")
    ap.add_argument("--no-ivf", action="store_true", help="skip the IVF4096,PQ64 / IVF4096,Flat / IVF4096,SQ8 legs")
    ap.add_argument("--scale-legs", default="ivfflat_10m,ivfpq_100m,ivfpq_shards",
                    help="BASELINE.json configs[2] / configs[3] on one GPU and the configs[4] shard (--shard-rows rows per "
                         "GPU through the sharded code path); comma-separated subset, empty = none")
    ap.add_argument("--shard-rows", type=int, default=125000000,
                    help="rows per GPU of the sharded IVF4096,PQ64 leg `ivfpq_shards` (BASELINE.json configs[4]: 8 x 125M = 1B)")
    ap.add_argument("--no-sweep", action="store_true", help="skip the nprobe sweeps (QPS @ recall curves) of the nb=1M IVF legs")
    ap.add_argument("--cpu-budget-s", type=float, default=15.0,
                    help="seconds of reference-CPU search per scale leg (sizes the query sample of its measured cpu_baseline)")
    ap.add_argument("--budget-s", type=float, default=900.0,
                    help="a scale leg is started only while the run is expected to stay inside this many seconds "
                         "(ivfflat_10m needs ~10 s, ivfpq_100m ~20 s since the chunks are drawn on the device)")
    ap.add_argument("--multi-gpu", choices=["replicas", "shards"], default="replicas",
                    help="layout of the FLAT leg at N > 1: replicas = every GPU holds the database, queries are split "
                         "(IndexReplicas, the reference's default for databases that fit one GPU); shards = rows are split "
                         "(IndexShards).  The IVFPQ leg is always sharded.")
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
    # launched by torch.distributed.run (RANK set): the process group is RCCL even for ONE rank, so that a 1-GPU box runs the
    # very init order / collectives / gather path an N-GPU node does (tests/test_gpu_rccl_world1.py); plain `python bench.py`
    # (the driver's N = 1 command) involves no process group at all
    use_dist = world > 1 or "RANK" in os.environ
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    ranks_seen, devices = 1, [local_rank]
    if use_dist:
        # self-validation of an N-GPU record: the ranks the RCCL backend really connected (all-reduce of ones) and the HIP
        # device each of them runs on (all-gather of hipGetDevice), printed in the line
        ones = torch.ones(1, dtype=torch.int32, device=dev)
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())
        mine = torch.tensor([torch.cuda.current_device()], dtype=torch.int32, device=dev)
        box = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(box, mine)
        devices = [int(b.item()) for b in box]

    import faiss_amd  # after torch: both then share one HIP runtime in this process
    from faiss_amd.datasets import synthetic_dataset
    from faiss_amd.distributed import ReplicatedSearcher, ShardedSearcher, replica_bounds, shard_bounds

    res = faiss_amd.StandardGpuResources(local_rank)
    t0 = time.time()
    t_start = time.time()
    xt, xb, xq, dmap = synthetic_dataset(D, NT, NB, NQ, seed=1338, return_map=True)
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
        rep = ReplicatedSearcher(local_search_block, NQ, dev, force_collectives=use_dist)

        class _S:  # same call shape as ShardedSearcher.search
            @staticmethod
            def search(_xq, k):
                return rep.search(k)
        searcher = _S
    else:
        searcher = ShardedSearcher(local_search, merge, [b - a for a, b in bounds], dev, force_collectives=use_dist)

    def barrier():
        if use_dist:
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
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    used_filter, n_overflow = index.filter_stats()
    dom = "flat_filter_kernel" if used_filter else "flat_scan_kernel"
    scan_ms, scan_n = res.profile_get(dom)
    others = {}
    for label, key in (("flat_filter_kernel<MODE_MAX> (chunk maxima on a 1/8 sample of the tiles)", "flat_filter_kernel_max"),
                       ("flat_tighten_kernel", "flat_tighten_kernel"), ("flat_rerank_kernel", "flat_rerank_kernel"),
                       ("convert_f16_query+norms", "convert_f16_query"), ("select_k_kernel", "select_k_kernel")):
        ms, n = res.profile_get(key)
        others[label] = round(ms / max(n, 1), 3)
    res.profile_enable(False)

    ivfpq_multi = None
    if world > 1 and not args.no_ivf:
        try:
            ivfpq_multi, ivf_out = sharded_ivfpq_leg(res, rank, world, dev, xt, xb, xq_dev, max(2, args.steps // 2), torch, dist)
        except Exception as e:  # noqa: BLE001
            ivfpq_multi, ivf_out = {"error": repr(e)[:300]}, None

    shards_multi = None
    if world > 1 and not args.no_ivf and "ivfpq_shards" in args.scale_legs.split(","):
        try:
            del index
            shards_multi = sharded_scale_leg(res, rank, world, dev, xt, xb, xq, xq_dev, dmap, args.shard_rows, 7, torch, dist, 8)
        except Exception as e:  # noqa: BLE001
            shards_multi = {"error": repr(e)[:300]}

    if rank != 0:
        # leave together with rank 0 (which still assembles and prints the line)
        dist.barrier()
        dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    qps = NQ * args.steps / elapsed
    gD, gI = out[0].cpu().numpy(), out[1].cpu().numpy()
    avg_scan_ms = scan_ms / max(scan_n, 1)
    # dominant kernel: the scan of this rank's shard.  Algorithmic work per launch = 2*nq*nb_shard*d flops
    # (256 MFLOP/query at nb=1M, d=128), priced against the dense MFMA peak of the unit the kernel
    # runs on (f16 for the filter, f32 for the exact scan).
    nq_rank0 = (replica_bounds(NQ, world)[0][0][1]) if replicas else NQ  # queries of the launches timed on rank 0
    flops = 2.0 * nq_rank0 * (hi - lo) * D
    achieved = flops / (avg_scan_ms * 1e-3) / 1e12
    peak = PEAK_F16_MFMA_TFLOPS if used_filter else PEAK_F32_MFMA_TFLOPS
    # one sweep of the shard + queries + results is the algorithmic HBM traffic of the launch
    row_bytes = D * (2.0 if used_filter else 4.0)
    hbm_bytes = (hi - lo) * row_bytes + nq_rank0 * row_bytes + nq_rank0 * K * 12.0
    line = {
        "metric": "QPS @ recall@1 (nq=10k, k=100) FlatL2, SIFT1M-shaped synthetic",
        "value": round(qps, 1), "unit": "QPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32 (f16 MFMA candidate filter + exact f32 re-rank: bit-identical to the f32 scan)" if used_filter else "f32", "data": "synthetic",
        "config": {"workload": "GpuIndexFlatL2 d=128 nb=1M nq=10k k=100 (BASELINE.json configs[1])",
                   "generator": "SyntheticDataset(d=128, nt=100k, nb=1M, nq=10k, seed=1338)",
                   "sharding": ("single GPU" if world == 1 else
                                "IndexReplicas-style: database on every GPU, %d-query blocks, result blocks gathered to rank 0"
                                % nq_loc if replicas else
                                "IndexShards-style rows/%d per GPU, gather to rank 0 + device merge" % world),
                   "inputs": "`value`: queries/results resident in HBM; `value_host_buffers`: pageable host buffers, "
                             "PCIe copies inside the timed region (SURVEY.md 8d / benchs/bench_gpu_sift1m.py)",
                   "flat_path": "filter" if used_filter else "exact"},
        "roofline": {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 2),
                     "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                     "whole_search_frac": round(2.0 * NQ * NB * D / (ms_per_step * 1e-3) / 1e12 / peak, 4) if world == 1 else None,
                     "mfma_busy_frac_from_committed_profile": committed_mfma_busy("flat_filter_kernel<1, 1,") if used_filter else None,
                     "traffic": committed_traffic("flat_filter_kernel<1, 1,", hbm_bytes, "flat") if used_filter and world == 1 else None,
                     "avg_kernel_ms": round(avg_scan_ms, 3), "launches": int(scan_n),
                     "algorithmic_hbm_GBps": round(hbm_bytes / (avg_scan_ms * 1e-3) / 1e9, 1),
                     "hbm_frac": round(hbm_bytes / (avg_scan_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 5)},
        "other_kernels_ms": others,
        "filter_overflow_queries": int(n_overflow),
        "ranks_seen": ranks_seen, "devices": devices, "ranks_seen_via": "rccl all_reduce" if use_dist else "no process group",
    }
    if world == 1:
        # the same search handed pageable host buffers (queries H2D, results D2H inside the timed region)
        Dh = np.empty((NQ, K), dtype=np.float32)
        Ih = np.empty((NQ, K), dtype=np.int64)
        dt_host = time_search(index, torch, NQ, xq.ctypes.data, Dh.ctypes.data, Ih.ctypes.data, max(3, args.steps // 2), 1)
        line["value_host_buffers"] = round(NQ / dt_host, 1)
        line["ms_per_step_host_buffers"] = round(dt_host * 1e3, 3)
        assert np.array_equal(Ih, gI) and np.array_equal(Dh, gD)
        if not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline_flat(xb, xq, K, gD, gI)
                line["recall_at_1"] = line["cpu_baseline"].get("parity_vs_gpu", {}).get("recall_at_1")
            except Exception as e:  # noqa: BLE001
                line["cpu_baseline"] = {"error": repr(e)[:200]}
        if not args.no_ivf:
            ivf_idx = {}
            for kind in [v for v in args.ivf_legs.split(",") if v in ("ivfpq", "ivfflat", "ivfsq")]:
                try:
                    line[kind], ivf_idx[kind] = ivf_leg(kind, res, xt, xb, xq, xq_dev, gI[:, 0], max(2, args.steps // 2), torch,
                                                        with_cpu=not args.no_cpu_baseline,
                                                        sweep=not args.no_sweep and kind in ("ivfpq", "ivfflat"))
                except Exception as e:  # noqa: BLE001
                    line[kind] = {"error": repr(e)[:300]}
            try:
                line["predicted_per_rank_ms"] = predicted_per_rank(res, index, ivf_idx.get("ivfpq"), xb, xq_dev, torch)
            except Exception as e:  # noqa: BLE001
                line["predicted_per_rank_ms"] = {"error": repr(e)[:300]}
            ivf_idx.clear()
            # seconds a leg needs (build + search + oracle sample + measured CPU baseline on the GPU's lists)
            need = {"ivfflat_10m": 60.0, "ivfpq_100m": 150.0, "ivfpq_shards": 90.0}
            for name in [v for v in args.scale_legs.split(",") if v in need]:
                if time.time() - t_start + need[name] > args.budget_s:
                    line[name] = {"skipped": "--budget-s %.0f would be exceeded (%.0f s used, ~%.0f s needed)"
                                             % (args.budget_s, time.time() - t_start, need[name])}
                    continue
                try:
                    del index  # the flat index (0.8 GB) is not needed any more
                except NameError:
                    pass
                try:
                    if name == "ivfpq_shards":
                        line[name] = sharded_scale_leg(res, 0, 1, dev, xt, xb, xq, xq_dev, dmap, args.shard_rows, 7, torch, dist, 16)
                    else:
                        kind, nbig = name.split("_")
                        line[name] = scale_leg(kind, 10000000 if nbig == "10m" else 100000000, res, xt, xb, xq, xq_dev, dmap,
                                               torch, line.get(kind), with_cpu=not args.no_cpu_baseline,
                                               cpu_budget_s=args.cpu_budget_s)
                except Exception as e:  # noqa: BLE001
                    line[name] = {"error": repr(e)[:300]}
    elif ivfpq_multi is not None:
        if ivf_out is not None:
            I2 = ivf_out[1].cpu().numpy()
            ivfpq_multi["recall_at_1"] = round(float((I2[:, 0] == gI[:, 0]).mean()), 4)
            ivfpq_multi["recall_at_100"] = round(float((I2 == gI[:, :1]).any(axis=1).mean()), 4)
        line["ivfpq"] = ivfpq_multi
    if world > 1 and shards_multi is not None:
        line["ivfpq_shards"] = shards_multi
    print(compact_line(line, write_detail(line)), flush=True)
    if use_dist:
        if world > 1:
            dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
