#!/usr/bin/env python
"""tools/pq_dim_sweep.py -- IVFPQ beyond the LDS-codebook sweeps (round 5: the filter path over fp16 DECODED residuals): query-major
against list-major at d = 192 / 256 / 384 (nb = 500 000, nlist 1024, nprobe 16, k = 100), 10 000 and 1000 queries; results
asserted identical; what the automatic rule picks."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import faiss_amd
from faiss_amd.datasets import synthetic_dataset
res = faiss_amd.StandardGpuResources(0)
dev = torch.device("cuda", 0)
for d, M in ((128, 64), (192, 48), (256, 128), (256, 64), (384, 64)):
    xt, xb, xq = synthetic_dataset(d, 50000, 500000, 10000, seed=d)
    idx = faiss_amd.GpuIndexIVFPQ(res, d, 1024, M, 8, faiss_amd.METRIC_L2)
    idx.train(xt)
    idx.add(xb)
    idx.nprobe = 16
    xq_dev = torch.from_numpy(xq).to(dev)
    out = {}
    for nq in (10000, 1000):
        Dd = torch.empty((nq, 100), dtype=torch.float32, device=dev)
        Id = torch.empty((nq, 100), dtype=torch.int64, device=dev)
        ref = None
        for mode in (1, 2):
            idx.set_scan_mode(mode)
            for _ in range(2):
                idx.search_ptr(nq, xq_dev.data_ptr(), 100, Dd.data_ptr(), Id.data_ptr())
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5):
                idx.search_ptr(nq, xq_dev.data_ptr(), 100, Dd.data_ptr(), Id.data_ptr())
            torch.cuda.synchronize()
            out[(nq, mode)] = (time.perf_counter() - t0) / 5 * 1e3
            got = (Dd.cpu().numpy().copy(), Id.cpu().numpy().copy())
            if ref is None:
                ref = got
            assert np.array_equal(ref[0], got[0]) and np.array_equal(ref[1], got[1])
        idx.set_scan_mode(0)
        out[(nq, "auto")] = idx.list_major_rule(nq, 16, 100)
    print("d %3d M %3d: 10 000 queries query-major %.2f ms, list-major %.2f ms (rule: %s); 1000 queries %.2f / %.2f ms (rule: %s); copies %.0f MB" % (
        d, M, out[(10000, 1)], out[(10000, 2)], "list-major" if out[(10000, "auto")] else "query-major", out[(1000, 1)], out[(1000, 2)],
        "list-major" if out[(1000, "auto")] else "query-major", idx.resident_bytes()[1] / 1e6), flush=True)
    del idx
