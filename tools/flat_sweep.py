#!/usr/bin/env python
"""tools/flat_sweep.py -- GpuIndexFlatL2 search time vs query batch size on the bench database (1M x 128), with the
planner's geometry / split count overridden through the timing knobs (FAISS_AMD_FILTER_GEOM / _NSPLIT)."""
import os, sys, time
os.environ["FAISS_AMD_EXPERIMENTS"] = "1"  # the library reads its FAISS_AMD_* knobs only behind this gate
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import faiss_amd
from faiss_amd.datasets import synthetic_dataset
res = faiss_amd.StandardGpuResources(0)
_, xb, xq = synthetic_dataset(128, 0, 1000000, 10000, seed=1338)
idx = faiss_amd.GpuIndexFlatL2(res, 128)
idx.add(xb)
dev = torch.device("cuda", 0)
xq_dev = torch.from_numpy(xq).to(dev)
Dd = torch.empty((10000, 100), dtype=torch.float32, device=dev)
Id = torch.empty((10000, 100), dtype=torch.int64, device=dev)
cases = sys.argv[1:] or ["10000", "5120", "2560", "1280", "1280:0", "1280:2", "1280:2:256", "640", "640:2"]
for c in cases:
    f = c.split(":")
    nq = int(f[0])
    for key, i in (("FAISS_AMD_FILTER_GEOM", 1), ("FAISS_AMD_FILTER_NSPLIT", 2)):
        if len(f) > i and f[i] != "":
            os.environ[key] = f[i]
        else:
            os.environ.pop(key, None)
    for _ in range(3):
        idx.search_ptr(nq, xq_dev.data_ptr(), 100, Dd.data_ptr(), Id.data_ptr())
    res.profile_enable(True); res.profile_reset()
    torch.cuda.synchronize(); t0 = time.time()
    steps = 20
    for _ in range(steps):
        idx.search_ptr(nq, xq_dev.data_ptr(), 100, Dd.data_ptr(), Id.data_ptr())
    torch.cuda.synchronize()
    ms = (time.time() - t0) / steps * 1e3
    prof = " ".join("%s=%.3f" % (nm.replace("flat_", "").replace("_kernel", ""), res.profile_get(nm)[0] / max(1, res.profile_get(nm)[1]))
                    for nm in ("flat_filter_kernel_max", "flat_tighten_kernel", "flat_filter_kernel", "flat_rerank_kernel", "convert_f16_query"))
    res.profile_enable(False)
    print("   kernels ms: " + prof)
    print("nq=%-6d geom=%-4s nsplit=%-4s %.3f ms/search  %.2f M QPS  (x%d ranks = %.1f M QPS)" % (
        nq, f[1] if len(f) > 1 else "auto", f[2] if len(f) > 2 else "auto", ms, nq / ms / 1e3, 10000 // nq if nq < 10000 else 1,
        (10000 // nq if nq < 10000 else 1) * nq / ms / 1e3), flush=True)

# PCIe-inclusive rate: queries and results in (pageable) host memory, as benchs/bench_gpu_sift1m.py measures
if "host" in os.environ.get("SWEEP_EXTRA", "host"):
    for key in ("FAISS_AMD_FILTER_GEOM", "FAISS_AMD_FILTER_NSPLIT"):
        os.environ.pop(key, None)
    for _ in range(2):
        idx.search(xq, 100)
    t0 = time.time()
    for _ in range(10):
        D, I = idx.search(xq, 100)
    ms = (time.time() - t0) / 10 * 1e3
    print("host buffers in/out: nq=10000 %.3f ms/search  %.2f M QPS" % (ms, 10000 / ms / 1e3), flush=True)
