"""Timing of the device-resident k-means at the shapes IVF / PQ training use (GPU box)."""
import os, sys, time
import numpy as np
import torch
torch.cuda.init()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import faiss_amd

res = faiss_amd.StandardGpuResources()
rs = np.random.RandomState(0)
for (n, d, k, it) in [(65536, 2, 256, 25), (262144, 128, 4096, 10), (1000000, 64, 16384, 5)]:
    x = rs.rand(n, d).astype(np.float32)
    for mode in ("device", "host-loop"):
        if mode == "device":
            ix = faiss_amd.GpuIndexFlatL2(res, d)
        else:
            ix = faiss_amd.IndexReplicas(d, threaded=False)
            ix.add_replica(faiss_amd.GpuIndexFlatL2(res, d))
        c = faiss_amd.Clustering(d, k, niter=it, seed=1)
        c.train(x[:2 * k], ix) if False else None
        t0 = time.time()
        c.train(x, ix)
        dt = time.time() - t0
        print(f"n={n} d={d} k={k} niter={it} {mode}: {dt:.3f} s ({dt / it * 1e3:.2f} ms / iteration) on_device={c.on_device}", flush=True)
xt = rs.rand(100000, 128).astype(np.float32)
for rep in range(2):
    pq = faiss_amd.GpuIndexIVFPQ(res, 128, 4096, 64, 8, 1)
    t0 = time.time()
    pq.train(xt)
    print(f"IVF4096,PQ64 train on 100k x 128: {time.time() - t0:.3f} s", flush=True)
