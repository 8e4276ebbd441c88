#!/usr/bin/env python
"""tools/coarse_ab.py -- A/B of the one-launch coarse quantizer (flat_small_fused_kernel, round 6) on the bench's nb = 1M IVF legs:
IVF4096,PQ64 / IVF4096,Flat / IVF4096,SQ8, nprobe 32, 10 000 queries, k = 100; search time and the coarse quantizer's spans."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402,F401
import torch  # noqa: E402

torch.cuda.init()
import faiss_amd  # noqa: E402
from faiss_amd.datasets import synthetic_dataset  # noqa: E402

res = faiss_amd.StandardGpuResources(0)
dev = torch.device("cuda", 0)
xt, xb, xq = synthetic_dataset(128, 100000, 1000000, 10000, seed=1338)
xq_dev = torch.from_numpy(xq).to(dev)
D = [torch.empty((10000, 100), dtype=torch.float32, device=dev) for _ in range(2)]
I = [torch.empty((10000, 100), dtype=torch.int64, device=dev) for _ in range(2)]
legs = [("IVF4096,PQ64", lambda: faiss_amd.GpuIndexIVFPQ(res, 128, 4096, 64, 8, faiss_amd.METRIC_L2)),
        ("IVF4096,Flat", lambda: faiss_amd.GpuIndexIVFFlat(res, 128, 4096, faiss_amd.METRIC_L2)),
        ("IVF4096,SQ8", lambda: faiss_amd.GpuIndexIVFScalarQuantizer(res, 128, 4096, 0, faiss_amd.METRIC_L2, True))]
cent = None
for name, make in legs:
    idx = make()
    if cent is None:
        idx.train(xt)
        cent = idx.get_centroids()
    else:
        idx.copy_centroids(cent)
        idx.train(xt)
    idx.add(xb)
    idx.nprobe = 32
    print(name, flush=True)
    for rep in range(3):
        for on in (0, 1):
            idx.set_small_fused(on)
            for _ in range(3):
                idx.search_ptr(10000, xq_dev.data_ptr(), 100, D[on].data_ptr(), I[on].data_ptr())
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                idx.search_ptr(10000, xq_dev.data_ptr(), 100, D[on].data_ptr(), I[on].data_ptr())
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 10 * 1e3
            res.profile_enable(True)
            res.profile_reset()
            idx.search_ptr(10000, xq_dev.data_ptr(), 100, D[on].data_ptr(), I[on].data_ptr())
            names = ("convert_f16_query", "flat_small_fused_kernel", "flat_filter_kernel_max", "flat_tighten_kernel", "flat_filter_kernel",
                     "flat_rerank_kernel")
            sp = {k: res.profile_get(k)[0] for k in names}
            res.profile_enable(False)
            print("   run %d one-launch %d: search %.3f ms   coarse spans: %s" % (
                rep, on, ms, ", ".join("%s %.3f" % (k.replace("flat_", "").replace("_kernel", ""), v) for k, v in sp.items() if v > 0)), flush=True)
        same = bool(torch.equal(D[0], D[1]) and torch.equal(I[0], I[1]))
        print("   identical results: %s" % same, flush=True)
        assert same
    del idx
