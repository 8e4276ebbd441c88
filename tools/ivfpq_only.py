#!/usr/bin/env python
"""tools/ivfpq_only.py -- IVF4096,PQ64 search loop on the bench data, nothing else (profiling target)."""
import os, sys, time
os.environ["FAISS_AMD_EXPERIMENTS"] = "1"  # the library reads its FAISS_AMD_* knobs only behind this gate
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import faiss_amd
from faiss_amd.datasets import synthetic_dataset, synthetic_more_device
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
metric = faiss_amd.METRIC_INNER_PRODUCT if len(sys.argv) > 3 and sys.argv[3] == "ip" else faiss_amd.METRIC_L2
res = faiss_amd.StandardGpuResources(0)
t0 = time.time()
xt, xb, xq, dmap = synthetic_dataset(128, 100000, min(nb, 1000000), 10000, seed=1338, return_map=True)
idx = faiss_amd.GpuIndexIVFPQ(res, 128, 4096, 64, 8, metric)
idx.train(xt); idx.add(xb); idx.nprobe = 32
done, chunk = len(xb), 0
while done < nb:  # BASELINE.json configs[3]: nb = 100M, generated and added chunk by chunk
    chunk += 1
    n_c = min(1000000, nb - done)
    xbc = synthetic_more_device(dmap, n_c, 1338 + chunk, torch.device("cuda", 0))
    idx.add_ptr(n_c, xbc.data_ptr())
    done += n_c
    del xbc
if os.environ.get("TWO_COPIES", "1") != "1":
    idx.set_lmf_two_copies(False)  # A/B: round 4's one-copy sweeps
print("train+add of %d vectors: %.1fs" % (nb, time.time() - t0), flush=True)
dev = torch.device("cuda", 0)
xq_dev = torch.from_numpy(xq).to(dev)
Dd = torch.empty((10000, 100), dtype=torch.float32, device=dev)
Id = torch.empty((10000, 100), dtype=torch.int64, device=dev)
for fb in os.environ.get("FB_LIST", "512").split(","):
    os.environ["FAISS_AMD_IVFPQ_FB"] = fb
    print("--- workgroup size", fb)
    idx.search_ptr(10000, xq_dev.data_ptr(), 100, Dd.data_ptr(), Id.data_ptr())
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(steps):
        idx.search_ptr(10000, xq_dev.data_ptr(), 100, Dd.data_ptr(), Id.data_ptr())
    torch.cuda.synchronize()
    dt_step = (time.time() - t0) / steps
    print("ivfpq search (%s): %.3f ms/step" % ("ip" if metric == 0 else "l2", (time.time() - t0) / steps * 1e3))
    res.profile_enable(True); res.profile_reset()
    idx.search_ptr(10000, xq_dev.data_ptr(), 100, Dd.data_ptr(), Id.data_ptr())
    print("scan_info", idx.scan_info())
    for kn in ("ivfpq_fused_kernel", "ivf_finish_kernel", "ivf_lm_plan", "ivf_lm_scan_pass1", "ivf_lm_threshold", "ivf_lm_scan_pass2", "flat_scan_kernel", "select_k_kernel", "flat_filter_kernel", "flat_filter_kernel_max", "flat_tighten_kernel", "flat_rerank_kernel", "convert_f16_query"):
        ms, n = res.profile_get(kn)
        if n:
            print("  %s %.3f ms (%d)" % (kn, ms, n))
