#!/usr/bin/env python
"""tools/selector_timing.py -- cost of an IDSelector at the bench shapes (queries / results resident in HBM): Flat 1M x
10k and IVF4096,PQ64 / IVF4096,Flat nprobe 32, each without a selector, with a selector that admits everything (the
overhead of the mask pass and the bit tests alone), with half of the labels, and with 1 % of them."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import faiss_amd
from faiss_amd.datasets import synthetic_dataset
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
res = faiss_amd.StandardGpuResources(0)
nb, nq, k = 1000000, 10000, 100
xt, xb, xq = synthetic_dataset(128, 100000, nb, nq, seed=1338)
dev = torch.device("cuda", 0)
xq_dev = torch.from_numpy(xq).to(dev)
Dd = torch.empty((nq, k), dtype=torch.float32, device=dev)
Id = torch.empty((nq, k), dtype=torch.int64, device=dev)
rs = np.random.RandomState(0)
sels = [("none", None), ("all", faiss_amd.IDSelectorAll()), ("range 50%", faiss_amd.IDSelectorRange(nb // 4, 3 * nb // 4)),
        ("batch 50%", faiss_amd.IDSelectorBatch(rs.permutation(nb)[: nb // 2])),
        ("bitmap 50%", faiss_amd.IDSelectorBitmap(rs.randint(0, 256, nb // 8).astype(np.uint8))),
        ("batch 1%", faiss_amd.IDSelectorBatch(rs.permutation(nb)[: nb // 100]))]


def run(name, idx, mk):
    for sname, sel in sels:
        params = None if sel is None else mk(sel)
        idx.search_ptr(nq, xq_dev.data_ptr(), k, Dd.data_ptr(), Id.data_ptr(), params)
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(steps):
            idx.search_ptr(nq, xq_dev.data_ptr(), k, Dd.data_ptr(), Id.data_ptr(), params)
        torch.cuda.synchronize()
        ms = (time.time() - t0) / steps * 1e3
        extra = ""
        if hasattr(idx, "filter_stats"):
            extra = " filter path %s, overflow queries %d" % idx.filter_stats()
        valid = int((Id >= 0).sum().item())
        print("%-8s selector %-11s %.3f ms/step, %d of %d results filled%s" % (name, sname, ms, valid, nq * k, extra), flush=True)


flat = faiss_amd.GpuIndexFlatL2(res, 128)
flat.add(xb)
run("flat", flat, lambda s: faiss_amd.SearchParameters(sel=s))
del flat
pq = faiss_amd.GpuIndexIVFPQ(res, 128, 4096, 64, 8, faiss_amd.METRIC_L2)
pq.train(xt); pq.add(xb); pq.nprobe = 32
run("ivfpq", pq, lambda s: faiss_amd.SearchParametersIVF(sel=s))
ivf = faiss_amd.GpuIndexIVFFlat(res, 128, 4096, faiss_amd.METRIC_L2)
ivf.copy_centroids(pq.get_centroids()); ivf.add(xb); ivf.nprobe = 32
del pq
run("ivfflat", ivf, lambda s: faiss_amd.SearchParametersIVF(sel=s))
