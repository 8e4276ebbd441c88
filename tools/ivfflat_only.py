#!/usr/bin/env python
"""tools/ivfflat_only.py -- GpuIndexIVFFlat (nlist=4096, nprobe=32) search loop on SIFT-shaped synthetic data.
usage: ivfflat_only.py [steps] [nb]   (BASELINE.json configs[2]: nb = 10M)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import faiss_amd
from faiss_amd.datasets import synthetic_dataset, synthetic_more_device
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
res = faiss_amd.StandardGpuResources(0)
t0 = time.time()
xt, xb0, xq, dmap = synthetic_dataset(128, 100000, min(nb, 1000000), 10000, seed=1338, return_map=True)
idx = faiss_amd.GpuIndexIVFFlat(res, 128, 4096, faiss_amd.METRIC_L2)
idx.train(xt)
idx.add(xb0)
done = len(xb0)
chunk = 0
while done < nb:  # further chunks of the same distribution (never 5 GB on the host at once)
    chunk += 1
    n_c = min(1000000, nb - done)
    xbc = synthetic_more_device(dmap, n_c, 1338 + chunk, torch.device("cuda", 0))
    idx.add_ptr(n_c, xbc.data_ptr())
    done += n_c
    del xbc
print("train+add of %d vectors: %.1fs" % (nb, time.time() - t0), flush=True)
idx.nprobe = 32
if os.environ.get("LMF_PAIR") is not None:
    idx.set_lmf_pair(int(os.environ["LMF_PAIR"]))  # A/B: lock-step pair sweeps (round 6) on / off
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
print("ivfflat nb=%d: %.3f ms/step = %.0f QPS; scan_info %s" % (nb, dt * 1e3, 10000 / dt, idx.scan_info()))
for kn in ("ivfflat_fused_kernel", "ivf_finish_kernel", "ivf_lm_plan", "ivf_lm_scan_pass1", "ivf_lm_threshold", "ivf_lm_scan_pass2", "select_k_kernel"):
    ms, n = res.profile_get(kn)
    if n:
        print("  %s %.3f ms (%d)" % (kn, ms, n))
