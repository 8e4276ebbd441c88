#!/usr/bin/env python
"""tools/rule_sweep.py -- validates GpuIndexIVF::list_major_rule (scan_mode 0) away from the bench shape: query-major vs
list-major search time of IVFFlat and IVFPQ (PQ64) over nlist in {1024, 4096, 16384}, nprobe in {8, 32, 128} and batches of
512 ... 10 000 queries at nb = 1M (d = 128, k = 100), with what the rule picks beside the faster one.

usage: python tools/rule_sweep.py [ivfflat,ivfpq] > gpurun_out/rule_sweep.txt"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import faiss_amd  # noqa: E402
from faiss_amd.datasets import synthetic_dataset  # noqa: E402

D, NT, NB, K = 128, 100000, 1000000, 100


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
    xt, xb, xq = synthetic_dataset(D, NT, NB, 10000, seed=1338)
    xq_dev = torch.from_numpy(xq).to(dev)
    Dd = torch.empty((10000, K), dtype=torch.float32, device=dev)
    Id = torch.empty((10000, K), dtype=torch.int64, device=dev)
    wrong = total = 0
    for nlist in (1024, 4096, 16384):
        for kind in (sys.argv[1].split(",") if len(sys.argv) > 1 else ("ivfflat", "ivfpq")):
            idx = (faiss_amd.GpuIndexIVFPQ(res, D, nlist, 64, 8, faiss_amd.METRIC_L2) if kind == "ivfpq"
                   else faiss_amd.GpuIndexIVFFlat(res, D, nlist, faiss_amd.METRIC_L2))
            idx.train(xt)
            idx.add(xb)
            for nprobe in (8, 32, 128):
                idx.nprobe = nprobe
                for n in (512, 1024, 2048, 4096, 10000):
                    idx.set_scan_mode(1)
                    tq = timed(idx, n, xq_dev, Dd, Id)
                    idx.set_scan_mode(2)
                    tl = timed(idx, n, xq_dev, Dd, Id)
                    pick = idx.list_major_rule(n, nprobe, K)
                    best = tl < tq
                    loss = (tl / tq if pick else tq / tl) if pick != best else 1.0
                    total += 1
                    wrong += pick != best and loss > 1.1
                    print("%-7s nlist %5d nprobe %3d n %5d: query-major %7.3f ms  list-major %7.3f ms  rule picks %-11s %s"
                          % (kind, nlist, nprobe, n, tq, tl, "list-major" if pick else "query-major",
                             "" if pick == best else "(the other is %.2fx faster)" % loss), flush=True)
            del idx
    print("rule off by more than 10 %% in %d of %d cases" % (wrong, total))


if __name__ == "__main__":
    main()
