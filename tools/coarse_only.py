#!/usr/bin/env python
"""tools/coarse_only.py -- the IVF coarse quantizer alone: a GpuIndexFlat over 4096 centroids (k-means of the bench's training set),
10 000 queries, k = 32 (flat_small_fused_kernel when the index takes that path).  FAISS_AMD_LIB_VARIANT=fstiming prints the phases."""
import os
import sys
import time

os.environ.setdefault("FAISS_AMD_EXPERIMENTS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402,F401
import torch  # noqa: E402

torch.cuda.init()
import faiss_amd  # noqa: E402
from faiss_amd.datasets import synthetic_dataset  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
res = faiss_amd.StandardGpuResources(0)
dev = torch.device("cuda", 0)
xt, xb, xq = synthetic_dataset(128, 100000, 100000, 10000, seed=1338)
ivf = faiss_amd.GpuIndexIVFFlat(res, 128, 4096, faiss_amd.METRIC_L2)
ivf.train(xt)
cent = ivf.get_centroids()
flat = faiss_amd.GpuIndexFlat(res, 128, faiss_amd.METRIC_L2)
flat.add(cent)
xq_dev = torch.from_numpy(xq[:nq]).to(dev)
D = torch.empty((nq, 32), dtype=torch.float32, device=dev)
I = torch.empty((nq, 32), dtype=torch.int64, device=dev)
for _ in range(3):
    flat.search_ptr(nq, xq_dev.data_ptr(), 32, D.data_ptr(), I.data_ptr())
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    flat.search_ptr(nq, xq_dev.data_ptr(), 32, D.data_ptr(), I.data_ptr())
torch.cuda.synchronize()
print("coarse quantizer, %d queries x 4096 centroids, k = 32: %.3f ms per search" % (nq, (time.perf_counter() - t0) / steps * 1e3))
res.profile_enable(True)
res.profile_reset()
flat.search_ptr(nq, xq_dev.data_ptr(), 32, D.data_ptr(), I.data_ptr())
for kn in ("flat_small_fused_kernel", "convert_f16_query", "flat_scan_kernel", "select_k_kernel", "prep_queries", "flat_filter_kernel_max", "flat_filter_kernel", "flat_tighten_kernel", "flat_rerank_kernel"):
    ms, n = res.profile_get(kn)
    if n:
        print("  %s %.3f ms (%d)" % (kn, ms, n))
print("checksum", float(D.sum()), int(I.sum()))
D2, I2 = torch.empty_like(D), torch.empty_like(I)
flat.set_small_fused(0)
flat.search_ptr(nq, xq_dev.data_ptr(), 32, D2.data_ptr(), I2.data_ptr())
torch.cuda.synchronize()
print("identical to the general path:", bool(torch.equal(D, D2) and torch.equal(I, I2)))
t0 = time.perf_counter()
for _ in range(steps):
    flat.search_ptr(nq, xq_dev.data_ptr(), 32, D2.data_ptr(), I2.data_ptr())
torch.cuda.synchronize()
print("general path: %.3f ms per search" % ((time.perf_counter() - t0) / steps * 1e3))
