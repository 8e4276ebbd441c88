#!/usr/bin/env python
"""tools/dim_sweep.py -- IVFFlat beyond d = 128: query-major vs list-major (behind the f16 filter: 16 / 24 / 32 k-steps per
row, two or one 32-query block per work item) at nb = 500 000, nlist 1024, nprobe 16, k = 100, 10 000 and 1000 queries.
Results of the two scans are asserted identical on every query.

usage: python tools/dim_sweep.py > gpurun_out/dim_sweep.txt"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import faiss_amd  # noqa: E402
from faiss_amd.datasets import synthetic_dataset  # noqa: E402

NT, NB, NQ, K, NLIST, NPROBE = 50000, 500000, 10000, 100, 1024, 16


def timed(idx, n, xq_dev, Dd, Id, steps=5):
    for _ in range(2):
        idx.search_ptr(n, xq_dev.data_ptr(), K, Dd.data_ptr(), Id.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        idx.search_ptr(n, xq_dev.data_ptr(), K, Dd.data_ptr(), Id.data_ptr())
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    dev = torch.device("cuda", 0)
    res = faiss_amd.StandardGpuResources(0)
    for d in (128, 200, 256, 384, 512):
        xt, xb, xq = synthetic_dataset(d, NT, NB, NQ, seed=1338)
        xq_dev = torch.from_numpy(xq).to(dev)
        Dd = torch.empty((NQ, K), dtype=torch.float32, device=dev)
        Id = torch.empty((NQ, K), dtype=torch.int64, device=dev)
        idx = faiss_amd.GpuIndexIVFFlat(res, d, NLIST, faiss_amd.METRIC_L2)
        idx.train(xt)
        idx.add(xb)
        idx.nprobe = NPROBE
        for n in (NQ, 1000):
            idx.set_scan_mode(1)
            tq = timed(idx, n, xq_dev, Dd, Id)
            ref = (Dd[:n].cpu().numpy().copy(), Id[:n].cpu().numpy().copy())
            idx.set_scan_mode(2)
            tl = timed(idx, n, xq_dev, Dd, Id)
            same = np.array_equal(ref[0], Dd[:n].cpu().numpy()) and np.array_equal(ref[1], Id[:n].cpu().numpy())
            print("d %3d n %5d: query-major %8.3f ms  list-major %7.3f ms  (%.1fx)  identical %s  rule picks %s  redo %d"
                  % (d, n, tq, tl, tq / tl, same, "list-major" if idx.list_major_rule(n, NPROBE, K) else "query-major",
                     idx.scan_info()[2]), flush=True)
        del idx


if __name__ == "__main__":
    main()
