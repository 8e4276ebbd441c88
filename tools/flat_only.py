#!/usr/bin/env python
"""tools/flat_only.py -- GpuIndexFlatL2 search loop on the bench data (profiling target for PMC passes)."""
import os, sys, time
os.environ["FAISS_AMD_EXPERIMENTS"] = "1"  # the library reads its FAISS_AMD_* knobs only behind this gate
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import faiss_amd
from faiss_amd.datasets import synthetic_dataset
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
res = faiss_amd.StandardGpuResources(0)
NQ = int(os.environ.get("NQ", "10000"))
_, xb, xq = synthetic_dataset(128, 0, 1000000, 10000, seed=1338)
xq = xq[:NQ]
idx = faiss_amd.GpuIndexFlatL2(res, 128)
idx.add(xb)
dev = torch.device("cuda", 0)
xq_dev = torch.from_numpy(xq).to(dev)
Dd = torch.empty((NQ, 100), dtype=torch.float32, device=dev)
Id = torch.empty((NQ, 100), dtype=torch.int64, device=dev)
# DBG_LIST: comma-separated FAISS_AMD_FILTER_DBG values, each optionally "dbg:nsplit" and / or "dbg/stagger"
# (FAISS_AMD_FILTER_STAGGER: 0 default schedule, 1 / 2 the out-of-phase second wave per SIMD, flat_filter.hip)
for dbg in os.environ.get("DBG_LIST", "0").split(","):
    stag = "0"
    if "/" in dbg:
        dbg, stag = dbg.split("/")
    os.environ["FAISS_AMD_FILTER_STAGGER"] = stag
    if ":" in dbg:
        dbg, ns = dbg.split(":")
        os.environ["FAISS_AMD_FILTER_NSPLIT"] = ns
    os.environ["FAISS_AMD_FILTER_DBG"] = dbg
    idx.search_ptr(NQ, xq_dev.data_ptr(), 100, Dd.data_ptr(), Id.data_ptr())
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(steps):
        idx.search_ptr(NQ, xq_dev.data_ptr(), 100, Dd.data_ptr(), Id.data_ptr())
    torch.cuda.synchronize()
    print("flat search nq=%d (dbg %s stagger %s nsplit %s geom %s): %.3f ms/step" % (NQ, dbg, stag, os.environ.get("FAISS_AMD_FILTER_NSPLIT"), os.environ.get("FAISS_AMD_FILTER_GEOM"), (time.time() - t0) / steps * 1e3), flush=True)
    res.profile_enable(True); res.profile_reset()
    idx.search_ptr(NQ, xq_dev.data_ptr(), 100, Dd.data_ptr(), Id.data_ptr())
    print("   per kernel (ms, launches):", {kn: tuple(round(v, 4) for v in res.profile_get(kn)) for kn in (
        "convert_f16_query", "flat_filter_kernel_max", "flat_tighten_kernel", "flat_filter_kernel", "flat_rerank_kernel")})
    res.profile_enable(False)

