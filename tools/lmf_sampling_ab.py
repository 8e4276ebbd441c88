#!/usr/bin/env python
"""tools/lmf_sampling_ab.py -- share of the rows sweep 1 of the filter path looks at (set_lmf_sampling: prefix of an item's rows >> shift),
same process, results must be identical.  IVF4096,PQ64 and IVF4096,Flat, nprobe 32, 10 000 queries, k = 100; sizes in millions."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402,F401
import torch  # noqa: E402

torch.cuda.init()
import faiss_amd  # noqa: E402
from faiss_amd.datasets import synthetic_dataset, synthetic_more_device  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "pq"
sizes = [int(a) for a in sys.argv[2:]] or [10]
res = faiss_amd.StandardGpuResources(0)
dev = torch.device("cuda", 0)
xt, xb, xq, dmap = synthetic_dataset(128, 100000, 1000000, 10000, seed=1338, return_map=True)
xq_dev = torch.from_numpy(xq).to(dev)
D0 = torch.empty((10000, 100), dtype=torch.float32, device=dev)
I0 = torch.empty((10000, 100), dtype=torch.int64, device=dev)
D = torch.empty_like(D0)
I = torch.empty_like(I0)
idx = faiss_amd.GpuIndexIVFPQ(res, 128, 4096, 64, 8, faiss_amd.METRIC_L2) if kind == "pq" else faiss_amd.GpuIndexIVFFlat(res, 128, 4096, faiss_amd.METRIC_L2)
idx.train(xt)
idx.add(xb)
idx.nprobe = 32
done, chunk = len(xb), 0
for mb in sizes:
    nb = mb * 1000000
    while done < nb:
        chunk += 1
        n_c = min(1000000, nb - done)
        xbc = synthetic_more_device(dmap, n_c, 1338 + chunk, dev)
        idx.add_ptr(n_c, xbc.data_ptr())
        done += n_c
        del xbc
    print("%s nb = %dM" % (kind, mb), flush=True)
    for rep in range(2):
        for shift in (0, 1, 2, 3, 4, -1):
            idx.set_lmf_sampling(shift)
            out = (D0, I0) if shift == 0 else (D, I)
            for _ in range(2):
                idx.search_ptr(10000, xq_dev.data_ptr(), 100, out[0].data_ptr(), out[1].data_ptr())
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(8):
                idx.search_ptr(10000, xq_dev.data_ptr(), 100, out[0].data_ptr(), out[1].data_ptr())
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 8 * 1e3
            res.profile_enable(True)
            res.profile_reset()
            idx.search_ptr(10000, xq_dev.data_ptr(), 100, out[0].data_ptr(), out[1].data_ptr())
            sp = {k: res.profile_get(k)[0] for k in ("ivf_lmf_sweep_min", "ivf_lmf_bound", "ivf_lmf_sweep_collect", "ivf_lmf_tighten", "ivf_lmf_rerank")}
            res.profile_enable(False)
            same = shift == 0 or bool(torch.equal(D, D0) and torch.equal(I, I0))
            print("   run %d sampling %2d: search %.3f ms   sweep 1 %.3f bound %.3f sweep 2 %.3f tighten %.3f rerank %.3f  redo %d  same %s" % (
                rep, shift, ms, sp["ivf_lmf_sweep_min"], sp["ivf_lmf_bound"], sp["ivf_lmf_sweep_collect"], sp["ivf_lmf_tighten"],
                sp["ivf_lmf_rerank"], idx.scan_info()[2], same), flush=True)
            assert same
idx.set_lmf_sampling(0)
