"""tools/host_path_sweep.py -- host-resident queries / results (the metric as SURVEY 8(d) defines it) against the device-resident
search, for the paged path's knobs (StandardGpuResources.setPagedSearch: minimum bytes, queries per page).  Flat 1M and IVF4096,PQ64
1M, 10 000 queries, k = 100."""
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
Dh = np.empty((NQ, K), dtype=np.float32)
Ih = np.empty((NQ, K), dtype=np.int64)


def timeit(fn, reps=15):
    fn()
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def leg(name, idx):
    dev_ms = timeit(lambda: idx.search_ptr(NQ, xq_dev.data_ptr(), K, Dd.data_ptr(), Id.data_ptr()))
    ref = (Dd.cpu().numpy().copy(), Id.cpu().numpy().copy())
    print("%s: device-resident %.3f ms" % (name, dev_ms), flush=True)
    for min_bytes, page in ((64 << 20, 0), (1 << 20, 5120), (1 << 20, 4096), (1 << 20, 3456), (1 << 20, 2560), (1 << 20, 2048), (1 << 20, 1280)):
        res.setPagedSearch(min_bytes, page)
        ms = timeit(lambda: idx.search_ptr(NQ, xq.ctypes.data, K, Dh.ctypes.data, Ih.ctypes.data))  # pageable host buffers
        ok = np.array_equal(Dh, ref[0]) and np.array_equal(Ih, ref[1])
        print("   paged_min %3d MiB page %5d: host buffers %.3f ms = %.3f of device-resident  (same results: %s)" % (
            min_bytes >> 20, page, ms, dev_ms / ms, ok), flush=True)
    res.setPagedSearch(64 << 20, 0)


flat = faiss_amd.GpuIndexFlatL2(res, D)
flat.add(xb)
leg("Flat 1M", flat)
del flat
pq = faiss_amd.GpuIndexIVFPQ(res, D, 4096, 64, 8, faiss_amd.METRIC_L2)
pq.train(xt)
pq.add(xb)
pq.nprobe = 32
leg("IVF4096,PQ64 1M", pq)
