#!/usr/bin/env python
"""tools/host_out_ab.py -- Flat nb = 1M, 10 000 queries, k = 100: device-resident search against the search on pageable host buffers
(SURVEY 8d's metric); results of the two must be identical."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.cuda.init()
import faiss_amd  # noqa: E402
from faiss_amd.datasets import synthetic_dataset  # noqa: E402

D, NB, NQ, K = 128, 1000000, 10000, 100
res = faiss_amd.StandardGpuResources(0)
xt, xb, xq = synthetic_dataset(D, 100000, NB, NQ, seed=1338)
dev = torch.device("cuda", 0)
xq_dev = torch.from_numpy(xq).to(dev)
Dd = torch.empty((NQ, K), dtype=torch.float32, device=dev)
Id = torch.empty((NQ, K), dtype=torch.int64, device=dev)
idx = faiss_amd.GpuIndexFlat(res, D, faiss_amd.METRIC_L2)
idx.add(xb)


def timeit(fn, reps=15):
    fn()
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


Dh = np.zeros((NQ, K), dtype=np.float32)
Ih = np.zeros((NQ, K), dtype=np.int64)
for rep in range(3):
    dev_ms = timeit(lambda: idx.search_ptr(NQ, xq_dev.data_ptr(), K, Dd.data_ptr(), Id.data_ptr()))
    host_ms = timeit(lambda: idx.search_ptr(NQ, xq.ctypes.data, K, Dh.ctypes.data, Ih.ctypes.data))
    same = bool(np.array_equal(Dh, Dd.cpu().numpy()) and np.array_equal(Ih, Id.cpu().numpy()))
    print("run %d: device-resident %.3f ms, host buffers %.3f ms (%.3f of the device-resident rate), identical %s" % (rep, dev_ms, host_ms, dev_ms / host_ms, same), flush=True)
    assert same
for nq in (1000, 3000, 5000):
    Dh, Ih = idx.search(xq[:nq], K)
    idx.search_ptr(nq, xq_dev.data_ptr(), K, Dd.data_ptr(), Id.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(Dh, Dd.cpu().numpy()[:nq]) and np.array_equal(Ih, Id.cpu().numpy()[:nq]), nq
print("subsets identical")
