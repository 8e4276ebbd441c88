#!/usr/bin/env python
"""tools/ivfpq_sweep.py -- fused IVFPQ kernel time against nprobe (fixed per-query cost vs scan cost) and nb.
usage: ivfpq_sweep.py [nb ...]   (default 1000000)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import faiss_amd
from faiss_amd.datasets import synthetic_dataset, synthetic_more
nbs = [int(a) for a in sys.argv[1:]] or [1000000]
res = faiss_amd.StandardGpuResources(0)
xt, xb, xq, dmap = synthetic_dataset(128, 100000, 1000000, 10000, seed=1338, return_map=True)
idx = faiss_amd.GpuIndexIVFPQ(res, 128, 4096, 64, 8, faiss_amd.METRIC_L2)
idx.train(xt)
dev = torch.device("cuda", 0)
xq_dev = torch.from_numpy(xq).to(dev)
Dd = torch.empty((10000, 100), dtype=torch.float32, device=dev)
Id = torch.empty((10000, 100), dtype=torch.int64, device=dev)
done, chunk = 0, 0
for nb in sorted(nbs):
    t0 = time.time()
    while done < nb:
        xbc = xb if chunk == 0 else synthetic_more(dmap, min(1000000, nb - done), seed=1338 + chunk)
        t1 = time.time()
        idx.add(xbc)
        if chunk < 3:
            print("add of %d vectors: %.3f s" % (len(xbc), time.time() - t1), flush=True)
        done += len(xbc); chunk += 1
    print("nb=%d built in %.1fs; arena %s" % (nb, time.time() - t0, idx.arena_stats()), flush=True)
    for nprobe in (1, 4, 16, 32, 64):
        idx.nprobe = nprobe
        idx.search_ptr(10000, xq_dev.data_ptr(), 100, Dd.data_ptr(), Id.data_ptr())
        res.profile_enable(True); res.profile_reset()
        for _ in range(3):
            idx.search_ptr(10000, xq_dev.data_ptr(), 100, Dd.data_ptr(), Id.data_ptr())
        ms, n = res.profile_get("ivfpq_fused_kernel")
        res.profile_enable(False)
        ms /= max(n, 1)
        by = nprobe * nb / 4096.0 * 64 * 10000
        print("nb=%d nprobe=%d: fused kernel %.3f ms = %.0f GB/s algorithmic (%.1f%% of 8 TB/s)" % (
            nb, nprobe, ms, by / (ms * 1e-3) / 1e9, by / (ms * 1e-3) / 8e12 * 100), flush=True)
