#!/usr/bin/env python
"""tools/ivfpq_phases.py -- per-phase shader-clock ticks of the fused IVFPQ kernel (FAISS_AMD_IVF_PHASES diagnostics) and
kernel time for the workgroup sizes 512 / 1024, nb = 1M, nprobe = 32."""
import os, sys, time
os.environ["FAISS_AMD_EXPERIMENTS"] = "1"  # the library reads its FAISS_AMD_* knobs only behind this gate
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import faiss_amd
from faiss_amd.datasets import synthetic_dataset
res = faiss_amd.StandardGpuResources(0)
xt, xb, xq = synthetic_dataset(128, 100000, 1000000, 10000, seed=1338)
idx = faiss_amd.GpuIndexIVFPQ(res, 128, 4096, 64, 8, faiss_amd.METRIC_L2)
idx.train(xt); idx.add(xb); idx.nprobe = 32
dev = torch.device("cuda", 0)
xq_dev = torch.from_numpy(xq).to(dev)
Dd = torch.empty((10000, 100), dtype=torch.float32, device=dev)
Id = torch.empty((10000, 100), dtype=torch.int64, device=dev)
for fb in ("512", "1024"):
    os.environ["FAISS_AMD_IVFPQ_FB"] = fb
    for _ in range(2):
        idx.search_ptr(10000, xq_dev.data_ptr(), 100, Dd.data_ptr(), Id.data_ptr())
    res.profile_enable(True); res.profile_reset()
    for _ in range(3):
        idx.search_ptr(10000, xq_dev.data_ptr(), 100, Dd.data_ptr(), Id.data_ptr())
    ms, n = res.profile_get("ivfpq_fused_kernel")
    res.profile_enable(False)
    print("workgroup %s: fused kernel %.3f ms" % (fb, ms / max(n, 1)), flush=True)
