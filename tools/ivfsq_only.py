#!/usr/bin/env python
"""tools/ivfsq_only.py -- GpuIndexIVFScalarQuantizer QT_8bit (nlist=4096, nprobe=32, residual encoding) search loop on SIFT-shaped synthetic data.
usage: ivfsq_only.py [steps] [nb]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import faiss_amd
from faiss_amd.datasets import synthetic_dataset, synthetic_more
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
res = faiss_amd.StandardGpuResources(0)
t0 = time.time()
xt, xb0, xq, dmap = synthetic_dataset(128, 100000, min(nb, 1000000), 10000, seed=1338, return_map=True)
idx = faiss_amd.GpuIndexIVFScalarQuantizer(res, 128, 4096, faiss_amd.ScalarQuantizer.QT_8bit, faiss_amd.METRIC_L2, True)
idx.train(xt)
idx.add(xb0)
done = len(xb0)
chunk = 0
while done < nb:  # further chunks of the same distribution (never 5 GB on the host at once)
    chunk += 1
    xbc = synthetic_more(dmap, min(1000000, nb - done), seed=1338 + chunk)
    idx.add(xbc)
    done += len(xbc)
print("train+add of %d vectors: %.1fs" % (nb, time.time() - t0), flush=True)
idx.nprobe = 32
dev = torch.device("cuda", 0)
xq_dev = torch.from_numpy(xq).to(dev)
Dd = torch.empty((10000, 100), dtype=torch.float32, device=dev)
Id = torch.empty((10000, 100), dtype=torch.int64, device=dev)
idx.search_ptr(10000, xq_dev.data_ptr(), 100, Dd.data_ptr(), Id.data_ptr())
torch.cuda.synchronize(); t0 = time.time()
for _ in range(steps):
    idx.search_ptr(10000, xq_dev.data_ptr(), 100, Dd.data_ptr(), Id.data_ptr())
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
res.profile_enable(True); res.profile_reset()
idx.search_ptr(10000, xq_dev.data_ptr(), 100, Dd.data_ptr(), Id.data_ptr())
print("scan_info %s" % (idx.scan_info(),))
for kn in ("ivf_lm_plan", "ivf_lm_scan_pass1", "ivf_lm_threshold", "ivf_lm_scan_pass2", "select_k_kernel", "ivf_finish_kernel"):
    ms_, n_ = res.profile_get(kn)
    if n_:
        print("  %s %.3f ms (%d)" % (kn, ms_, n_))
ms, n = res.profile_get("ivfsq_fused_kernel")
if not n:  # large batches take the list-major scan (ivf_listmajor.hip, kind 2): no fused launch to report
    print("ivfsq8 nb=%d: %.3f ms/step = %.0f QPS (list-major)" % (nb, dt * 1e3, 10000 / dt))
    sys.exit(0)
bytes_per_query = 32.0 * nb / 4096.0 * 128
print("ivfsq8 nb=%d: %.3f ms/step = %.0f QPS; fused kernel %.3f ms = %.0f GB/s algorithmic (%.1f%% of 8 TB/s)" % (
    nb, dt * 1e3, 10000 / dt, ms, bytes_per_query * 10000 / (ms * 1e-3) / 1e9, bytes_per_query * 10000 / (ms * 1e-3) / 8e12 * 100))
