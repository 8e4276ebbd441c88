#!/usr/bin/env python
"""tools/flat_pair_ab.py -- same-process A/B of the lock-step pair sweeps (ivf_lm_filter.hip PAIR, round 6) for IVF4096,Flat (and
IVF4096,SQ8 at 1M): nprobe 32, 10 000 queries, k = 100 at nb = 1M and 10M.  Identical results required; sweep times from the spans."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402,F401
import torch  # noqa: E402

torch.cuda.init()
import faiss_amd  # noqa: E402
from faiss_amd.datasets import synthetic_dataset, synthetic_more_device  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [1, 10]
res = faiss_amd.StandardGpuResources(0)
dev = torch.device("cuda", 0)
xt, xb, xq, dmap = synthetic_dataset(128, 100000, 1000000, 10000, seed=1338, return_map=True)
xq_dev = torch.from_numpy(xq).to(dev)
D = [torch.empty((10000, 100), dtype=torch.float32, device=dev) for _ in range(2)]
I = [torch.empty((10000, 100), dtype=torch.int64, device=dev) for _ in range(2)]


def ab(idx, label):
    print(label, flush=True)
    for rep in range(3):
        for on in (0, 1):
            idx.set_lmf_pair(2 * on)  # both sweeps
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
            sp = {k: res.profile_get(k)[0] for k in ("ivf_lmf_sweep_min", "ivf_lmf_sweep_collect", "ivf_lmf_rerank")}
            res.profile_enable(False)
            print("   run %d pair %d: search %.3f ms   sweep 1 %.3f  sweep 2 %.3f  rerank %.3f  scan %s" % (
                rep, on, ms, sp["ivf_lmf_sweep_min"], sp["ivf_lmf_sweep_collect"], sp["ivf_lmf_rerank"], idx.scan_info()), flush=True)
        same = bool(torch.equal(D[0], D[1]) and torch.equal(I[0], I[1]))
        print("   identical results: %s" % same, flush=True)
        assert same


idx = faiss_amd.GpuIndexIVFFlat(res, 128, 4096, faiss_amd.METRIC_L2)
idx.train(xt)
idx.add(xb)
idx.nprobe = 32
cent = idx.get_centroids()
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
    ab(idx, "IVF4096,Flat nb = %dM" % mb)
del idx
sq = faiss_amd.GpuIndexIVFScalarQuantizer(res, 128, 4096, 0, faiss_amd.METRIC_L2, True)
sq.copy_centroids(cent)
sq.train(xt)
sq.add(xb)
sq.nprobe = 32
ab(sq, "IVF4096,SQ8 nb = 1M")
