#!/usr/bin/env python
"""tools/sample_sweep.py -- round 5: the first sweep of the filter path on a sample of the rows (prefix of every work item,
faiss_amd_GpuIndexIVF_set_lmf_sampling) + the tightening launch, at the BASELINE shapes (10 000 queries, nprobe 32, k 100).
Per sample shift: search time, kernel spans, candidates per query before / after tightening are visible in the rerank time.
Results never change with the knob (asserted against shift -1).

usage: python tools/sample_sweep.py ivfflat_1m,ivfflat_10m,ivfpq_1m,ivfpq_10m,ivfpq_100m > gpurun_out/sample_sweep.txt"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["FAISS_AMD_EXPERIMENTS"] = "1"
import torch  # noqa: E402  (before the library: one HIP runtime)

import faiss_amd  # noqa: E402
from faiss_amd.datasets import synthetic_dataset, synthetic_more_device  # noqa: E402

D, NT, NQ, K, NLIST, NPROBE = 128, 100000, 10000, 100, 4096, 32
SPANS = ("ivf_lmf_prepare", "ivf_lm_plan", "ivf_lmf_sweep_min", "ivf_lmf_bound", "ivf_lmf_sweep_collect", "ivf_lmf_tighten",
         "ivf_lmf_rerank", "select_k_kernel")


def timed(idx, res, xq_dev, Dd, Id, steps=5):
    idx.search_ptr(NQ, xq_dev.data_ptr(), K, Dd.data_ptr(), Id.data_ptr())
    torch.cuda.synchronize()
    each = []
    for _ in range(steps):
        t0 = time.perf_counter()
        idx.search_ptr(NQ, xq_dev.data_ptr(), K, Dd.data_ptr(), Id.data_ptr())
        torch.cuda.synchronize()
        each.append(time.perf_counter() - t0)
    res.profile_enable(True)
    res.profile_reset()
    for _ in range(2):
        idx.search_ptr(NQ, xq_dev.data_ptr(), K, Dd.data_ptr(), Id.data_ptr())
    torch.cuda.synchronize()
    sp = {k: res.profile_get(k) for k in SPANS}
    res.profile_enable(False)
    return float(np.median(each)) * 1e3, {k: round(v[0] / v[1], 3) for k, v in sp.items() if v[1]}


def main():
    legs = (sys.argv[1] if len(sys.argv) > 1 else "ivfflat_10m,ivfpq_10m").split(",")
    shifts = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [-1, 1, 2, 3]
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
               else faiss_amd.GpuIndexIVFScalarQuantizer(res, D, NLIST, faiss_amd.ScalarQuantizer.QT_8bit, faiss_amd.METRIC_L2, True) if kind == "ivfsq"
               else faiss_amd.GpuIndexIVFFlat(res, D, NLIST, faiss_amd.METRIC_L2))
        idx.train(xt)
        idx.add(xb)
        for c in range(1, nb // 1000000):
            x = synthetic_more_device(dmap, 1000000, 1338 + c, dev)
            idx.add_ptr(1000000, x.data_ptr())
            del x
        idx.nprobe = NPROBE
        idx.set_scan_mode(2)
        print("==== %s nb=%d" % (kind, nb), flush=True)
        base = None
        for two in ((1, 0) if kind == "ivfpq" else (1,)):
            if kind == "ivfpq":
                idx.set_lmf_two_copies(bool(two))
            for shift in shifts:
                idx.set_lmf_sampling(shift)
                ms, sp = timed(idx, res, xq_dev, Dd, Id, steps=5)
                got = (Dd.cpu().numpy().copy(), Id.cpu().numpy().copy())
                if base is None:
                    base = got
                same = np.array_equal(base[0], got[0]) and np.array_equal(base[1], got[1])
                print("two copies %d sample shift %2d: %8.3f ms redo %5d same %s  %s" % (two, shift, ms, idx.scan_info()[2], same, sp),
                      flush=True)
        idx.set_lmf_sampling(0)
        del idx


if __name__ == "__main__":
    main()
