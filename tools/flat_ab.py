#!/usr/bin/env python
"""tools/flat_ab.py -- same-process A/B of the flat search (BASELINE.json configs[1]: nb = 1M, 10 000 queries, k = 100) over the
two planner knobs that trade the maxima pass against the collect pass: the sampling stride of the maxima pass
(FAISS_AMD_FILTER_TSTRIDE: every n-th tile of a split) and the number of database splits (FAISS_AMD_FILTER_NSPLIT).
Two rounds in ABAB order (box drift shows as the difference between the rounds); results are asserted identical.
VERDICT r4 item 7: "whole-search >= 0.40 or a committed A/B that says why not"."""
import os, sys, time
os.environ["FAISS_AMD_EXPERIMENTS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
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
NAMES = ("flat_filter_kernel_max", "flat_tighten_kernel", "flat_filter_kernel", "flat_rerank_kernel", "convert_f16_query")
cases = [("auto", "auto")] + [(t, "auto") for t in ("2", "4", "8", "16", "32")] + [("auto", n) for n in ("64", "96", "128", "192", "256")]
base = None
peak = 2.0 * 10000 * 1e6 * 128 / 2.5e15 * 1e3  # ms of the f16 matrix pipe at its dense peak
for rnd in range(2):
    for ts, ns in cases:
        for key, v in (("FAISS_AMD_FILTER_TSTRIDE", ts), ("FAISS_AMD_FILTER_NSPLIT", ns)):
            if v == "auto":
                os.environ.pop(key, None)
            else:
                os.environ[key] = v
        for _ in range(3):
            idx.search_ptr(10000, xq_dev.data_ptr(), 100, Dd.data_ptr(), Id.data_ptr())
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            idx.search_ptr(10000, xq_dev.data_ptr(), 100, Dd.data_ptr(), Id.data_ptr())
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 20 * 1e3
        res.profile_enable(True); res.profile_reset()
        for _ in range(3):
            idx.search_ptr(10000, xq_dev.data_ptr(), 100, Dd.data_ptr(), Id.data_ptr())
        torch.cuda.synchronize()
        prof = " ".join("%s=%.3f" % (nm.replace("flat_", "").replace("_kernel", ""), res.profile_get(nm)[0] / max(1, res.profile_get(nm)[1])) for nm in NAMES)
        res.profile_enable(False)
        got = (Dd.cpu().numpy().copy(), Id.cpu().numpy().copy())
        if base is None:
            base = got
        same = np.array_equal(base[0], got[0]) and np.array_equal(base[1], got[1])
        print("round %d tstride %-4s nsplit %-4s: %.3f ms/search = %.2f M QPS, whole-search share of the f16 peak %.3f, same results %s | %s"
              % (rnd, ts, ns, ms, 10.0 / ms, peak / ms, same, prof), flush=True)
