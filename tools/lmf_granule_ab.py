#!/usr/bin/env python
"""tools/lmf_granule_ab.py -- granule size of the filter sweeps' bound (set_lmf_tuning gran_blocks: 32 G rows per granule) at nb = 1M,
IVF4096,PQ64 / IVF4096,Flat, nprobe 32, 10 000 queries, k = 100; results must be identical."""
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
D0 = torch.empty((10000, 100), dtype=torch.float32, device=dev)
I0 = torch.empty((10000, 100), dtype=torch.int64, device=dev)
D = torch.empty_like(D0)
I = torch.empty_like(I0)
for kind in ("pq", "flat"):
    idx = faiss_amd.GpuIndexIVFPQ(res, 128, 4096, 64, 8, faiss_amd.METRIC_L2) if kind == "pq" else faiss_amd.GpuIndexIVFFlat(res, 128, 4096, faiss_amd.METRIC_L2)
    idx.train(xt)
    idx.add(xb)
    idx.nprobe = 32
    print(kind, flush=True)
    for rep in range(2):
        for G in (0, 1, 2, 4):
            idx.set_lmf_tuning(0, G, 0, 0)
            out = (D0, I0) if G == 0 else (D, I)
            for _ in range(3):
                idx.search_ptr(10000, xq_dev.data_ptr(), 100, out[0].data_ptr(), out[1].data_ptr())
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                idx.search_ptr(10000, xq_dev.data_ptr(), 100, out[0].data_ptr(), out[1].data_ptr())
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 20 * 1e3
            res.profile_enable(True)
            res.profile_reset()
            idx.search_ptr(10000, xq_dev.data_ptr(), 100, out[0].data_ptr(), out[1].data_ptr())
            sp = {k: res.profile_get(k)[0] for k in ("ivf_lmf_sweep_min", "ivf_lmf_bound", "ivf_lmf_sweep_collect", "ivf_lmf_tighten", "ivf_lmf_rerank")}
            res.profile_enable(False)
            same = G == 0 or bool(torch.equal(D, D0) and torch.equal(I, I0))
            print("   run %d G %d: search %.3f ms   sweep 1 %.3f bound %.3f sweep 2 %.3f tighten %.3f rerank %.3f  same %s" % (
                rep, G, ms, sp["ivf_lmf_sweep_min"], sp["ivf_lmf_bound"], sp["ivf_lmf_sweep_collect"], sp["ivf_lmf_tighten"], sp["ivf_lmf_rerank"], same), flush=True)
            assert same
    idx.set_lmf_tuning(0, 0, 0, 0)
    del idx
