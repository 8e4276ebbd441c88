#!/usr/bin/env python
"""tools/latency_sweep.py -- search latency against the batch size (1 ... 10 000 queries, device-resident queries and
results, k = 100) for GpuIndexFlatL2 nb = 1M, IVF4096,Flat nb = 1M / 10M and IVF4096,PQ64 nb = 10M (nprobe 32): the
small-batch end is bound by the number of launches and host synchronisations of a search, the large-batch end is the
bench line.  Prints ms per search, queries per second and the scan the rule picked.

usage: python tools/latency_sweep.py [flat_1m,ivfflat_1m,ivfflat_10m,ivfpq_10m] > gpurun_out/latency_sweep.txt"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import faiss_amd  # noqa: E402
from faiss_amd.datasets import synthetic_dataset, synthetic_more_device  # noqa: E402

D, NT, NQ, K, NLIST, NPROBE = 128, 100000, 10000, 100, 4096, 32


def timed(idx, n, xq_dev, Dd, Id):
    steps = 200 if n <= 256 else 40 if n <= 2048 else 10
    for _ in range(3):
        idx.search_ptr(n, xq_dev.data_ptr(), K, Dd.data_ptr(), Id.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        idx.search_ptr(n, xq_dev.data_ptr(), K, Dd.data_ptr(), Id.data_ptr())
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    legs = (sys.argv[1] if len(sys.argv) > 1 else "flat_1m,ivfflat_1m,ivfflat_10m,ivfpq_10m").split(",")
    dev = torch.device("cuda", 0)
    res = faiss_amd.StandardGpuResources(0)
    xt, xb, xq, dmap = synthetic_dataset(D, NT, 1000000, NQ, seed=1338, return_map=True)
    xq_dev = torch.from_numpy(xq).to(dev)
    Dd = torch.empty((NQ, K), dtype=torch.float32, device=dev)
    Id = torch.empty((NQ, K), dtype=torch.int64, device=dev)
    for leg in legs:
        kind, size = leg.split("_")
        nb = int(size[:-1]) * 1000000
        if kind == "flat":
            idx = faiss_amd.GpuIndexFlatL2(res, D)
        elif kind == "ivfpq":
            idx = faiss_amd.GpuIndexIVFPQ(res, D, NLIST, 64, 8, faiss_amd.METRIC_L2)
        else:
            idx = faiss_amd.GpuIndexIVFFlat(res, D, NLIST, faiss_amd.METRIC_L2)
        if kind != "flat":
            idx.train(xt)
            idx.nprobe = NPROBE
        idx.add(xb)
        for c in range(1, nb // 1000000):
            x = synthetic_more_device(dmap, 1000000, 1338 + c, dev)
            idx.add_ptr(1000000, x.data_ptr())
            del x
        print("==== %s nb=%d" % (kind, nb), flush=True)
        for n in (1, 4, 16, 64, 128, 256, 512, 1024, 4096, 10000):
            ms = timed(idx, n, xq_dev, Dd, Id)
            scan = "" if kind == "flat" else ("list-major" if idx.scan_info()[1] == 2 else "query-major")
            both = ""
            if kind != "flat" and 16 <= n <= 4096:  # both scans side by side: where the rule's switch point should sit
                idx.set_scan_mode(1)
                tq = timed(idx, n, xq_dev, Dd, Id)
                idx.set_scan_mode(2)
                tl = timed(idx, n, xq_dev, Dd, Id)
                idx.set_scan_mode(0)
                both = "  (query-major %.3f ms, list-major %.3f ms)" % (tq, tl)
            print("n %5d: %8.3f ms  %11.0f QPS  %s%s" % (n, ms, n / ms * 1e3, scan, both), flush=True)
        del idx


if __name__ == "__main__":
    main()
