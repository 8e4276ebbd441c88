#!/usr/bin/env python
"""tools/lmf_sweep.py -- timing experiments of the list-major scan behind the f16 filter (ivf_lm_filter.hip) at the
BASELINE shapes: IVFFlat nb = 1M / 10M and IVFPQ nb = 1M / 10M / 100M (10 000 queries, nprobe 32, k 100), over the
tuning knobs of faiss_amd_GpuIndexIVF_set_lmf_tuning (rows of a list per work item, blocks per granule, candidate room)
and against the query-major and f32 list-major scans.  Results never change with the knobs (asserted here).

usage: python tools/lmf_sweep.py [ivfflat_1m,ivfflat_10m,ivfpq_10m,ivfpq_100m] [rows-per-item values] > gpurun_out/lmf_sweep.txt"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402  (before the library: one HIP runtime)

import faiss_amd  # noqa: E402
from faiss_amd.datasets import synthetic_dataset, synthetic_more_device  # noqa: E402

D, NT, NQ, K, NLIST, NPROBE = 128, 100000, 10000, 100, 4096, 32
SPANS = ("ivf_lmf_prepare", "ivf_lm_plan", "ivf_lmf_sweep_min", "ivf_lmf_bound", "ivf_lmf_sweep_collect", "ivf_lmf_rerank",
         "ivf_lm_scan_pass1", "ivf_lm_threshold", "ivf_lm_scan_pass2", "select_k_kernel", "ivfflat_fused_kernel",
         "ivfpq_fused_kernel", "ivf_finish_kernel", "flat_filter_kernel", "flat_rerank_kernel")


def timed(idx, res, xq_dev, Dd, Id, steps=5):
    idx.search_ptr(NQ, xq_dev.data_ptr(), K, Dd.data_ptr(), Id.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        idx.search_ptr(NQ, xq_dev.data_ptr(), K, Dd.data_ptr(), Id.data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    res.profile_enable(True)
    res.profile_reset()
    for _ in range(2):
        idx.search_ptr(NQ, xq_dev.data_ptr(), K, Dd.data_ptr(), Id.data_ptr())
    torch.cuda.synchronize()
    sp = {k: res.profile_get(k) for k in SPANS}
    res.profile_enable(False)
    return dt * 1e3, {k: round(v[0] / v[1], 3) for k, v in sp.items() if v[1]}


def main():
    legs = (sys.argv[1] if len(sys.argv) > 1 else "ivfflat_1m,ivfflat_10m,ivfpq_10m,ivfpq_100m").split(",")
    dev = torch.device("cuda", 0)
    res = faiss_amd.StandardGpuResources(0)
    xt, xb, xq, dmap = synthetic_dataset(D, NT, 1000000, NQ, seed=1338, return_map=True)
    xq_dev = torch.from_numpy(xq).to(dev)
    Dd = torch.empty((NQ, K), dtype=torch.float32, device=dev)
    Id = torch.empty((NQ, K), dtype=torch.int64, device=dev)
    for leg in legs:
        kind, size = leg.split("_")
        nb = int(size[:-1]) * 1000000
        idx = (faiss_amd.GpuIndexIVFPQ(res, D, NLIST, 64, 8, faiss_amd.METRIC_L2) if kind == "ivfpq"
               else faiss_amd.GpuIndexIVFFlat(res, D, NLIST, faiss_amd.METRIC_L2))
        idx.train(xt)
        idx.add(xb)
        for c in range(1, nb // 1000000):
            x = synthetic_more_device(dmap, 1000000, 1338 + c, dev)
            idx.add_ptr(1000000, x.data_ptr())
            del x
        idx.nprobe = NPROBE
        print("==== %s nb=%d" % (kind, nb), flush=True)
        ref = None
        for mode, name in ((1, "query-major"), (3, "list-major f32"), (2, "list-major filter")):
            if mode == 1 and nb > 10000000 and kind == "ivfflat":
                continue
            idx.set_scan_mode(mode)
            idx.set_lmf_tuning()
            ms, sp = timed(idx, res, xq_dev, Dd, Id, steps=3 if nb >= 100000000 else 5)
            print("%-20s %8.3f ms  redo %d  %s" % (name, ms, idx.scan_info()[2], sp), flush=True)
            if mode == 1:
                ref = (Dd.cpu().numpy().copy(), Id.cpu().numpy().copy())
            if mode == 2 and ref is not None:
                ok = np.array_equal(ref[0], Dd.cpu().numpy()) and np.array_equal(ref[1], Id.cpu().numpy())
                print("    filter == query-major on all queries: %s" % ok, flush=True)
        base = (Dd.cpu().numpy().copy(), Id.cpu().numpy().copy())
        gs = (1, 2) if nb <= 1000000 else (4, 8, 16) if nb <= 10000000 else (8, 16, 32)
        for g in gs:
            for ms, cap in ((1, 1024), (2, 1024)) + (((4, 1024), (4, 2048)) if nb >= 100000000 else ()):
                idx.set_lmf_tuning(0, g, cap, ms)
                ms_t, sp = timed(idx, res, xq_dev, Dd, Id, steps=3)
                same = np.array_equal(base[0], Dd.cpu().numpy()) and np.array_equal(base[1], Id.cpu().numpy())
                print("g %d min_stride %d cap %5d: %8.3f ms redo %6d same %s  prep %.3f plan %.3f min %.3f bound %.3f collect %.3f rerank %.3f select %.3f"
                      % (g, ms, cap, ms_t, idx.scan_info()[2], same, sp.get("ivf_lmf_prepare", 0), sp.get("ivf_lm_plan", 0),
                         sp.get("ivf_lmf_sweep_min", 0), sp.get("ivf_lmf_bound", 0), sp.get("ivf_lmf_sweep_collect", 0),
                         sp.get("ivf_lmf_rerank", 0), sp.get("select_k_kernel", 0)), flush=True)
        # rows of a list per work item (0 = the rule: a quarter of an average list, 1024 ... 8192)
        for rt in ([int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else (256, 512, 1024, 2048, 4096)):
            idx.set_lmf_tuning(rt, 0, 0, 0)
            ms_t, sp = timed(idx, res, xq_dev, Dd, Id, steps=3)
            same = np.array_equal(base[0], Dd.cpu().numpy()) and np.array_equal(base[1], Id.cpu().numpy())
            print("rows per item %5d: %8.3f ms same %s  min %.3f collect %.3f" % (rt, ms_t, same, sp.get("ivf_lmf_sweep_min", 0),
                                                                              sp.get("ivf_lmf_sweep_collect", 0)), flush=True)
        del idx


if __name__ == "__main__":
    main()
