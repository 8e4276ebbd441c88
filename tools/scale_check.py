#!/usr/bin/env python
"""tools/scale_check.py -- the larger BASELINE.json configurations on one MI355X: speed AND parity.
usage: scale_check.py ivfpq|ivfflat NB [nsample]
  configs[2]: scale_check.py ivfflat 10000000      configs[3]: scale_check.py ivfpq 100000000
Builds the index chunk by chunk (synthetic SIFT-shaped data, never more than 1M rows on the host), times the search of
all 10 000 queries (k = 100, nprobe = 32, queries / results in HBM) and checks a sample of the queries BIT-EXACTLY
against the CPU oracle restatement run on the probed lists read back from the device; also sortedness / label validity
of all results.  (The oracle is the checker here, as in tests/.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import faiss_amd
from faiss_amd.datasets import synthetic_dataset, synthetic_more
from oracle.pyoracle import METRIC_L2, Oracle

kind, nb = sys.argv[1], int(sys.argv[2])
nsample = int(sys.argv[3]) if len(sys.argv) > 3 else 8
D_, NLIST, NPROBE, K, NQ = 128, 4096, 32, 100, 10000
res = faiss_amd.StandardGpuResources(0)
xt, xb, xq, dmap = synthetic_dataset(D_, 100000, min(nb, 1000000), NQ, seed=1338, return_map=True)
pq = kind == "ivfpq"
idx = faiss_amd.GpuIndexIVFPQ(res, D_, NLIST, 64, 8, METRIC_L2) if pq else faiss_amd.GpuIndexIVFFlat(res, D_, NLIST, METRIC_L2)
t0 = time.time()
idx.train(xt)
t_add, t_gen, done, chunk = 0.0, 0.0, 0, 0
while done < nb:
    t1 = time.time()
    xbc = xb if chunk == 0 else synthetic_more(dmap, min(1000000, nb - done), seed=1338 + chunk)
    t_gen += time.time() - t1
    t1 = time.time()
    idx.add(xbc)
    t_add += time.time() - t1
    done += len(xbc); chunk += 1
print("%s nb=%d: train+build %.1fs (add calls %.1fs = %.2f M vectors/s from host memory, data generation %.1fs); arena rows used/holes/allocated %s"
      % (kind, nb, time.time() - t0, t_add, nb / t_add / 1e6, t_gen, idx.arena_stats()), flush=True)
idx.nprobe = NPROBE
dev = torch.device("cuda", 0)
xq_dev = torch.from_numpy(xq).to(dev)
Dd = torch.empty((NQ, K), dtype=torch.float32, device=dev)
Id = torch.empty((NQ, K), dtype=torch.int64, device=dev)
idx.search_ptr(NQ, xq_dev.data_ptr(), K, Dd.data_ptr(), Id.data_ptr())
torch.cuda.synchronize(); t0 = time.time()
steps = 3
for _ in range(steps):
    idx.search_ptr(NQ, xq_dev.data_ptr(), K, Dd.data_ptr(), Id.data_ptr())
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
kname = "ivfpq_fused_kernel" if pq else "ivfflat_fused_kernel"
res.profile_enable(True); res.profile_reset()
idx.search_ptr(NQ, xq_dev.data_ptr(), K, Dd.data_ptr(), Id.data_ptr())
ms, n = res.profile_get(kname)
sel_ms, _ = res.profile_get("select_k_kernel")
res.profile_enable(False)
by = NPROBE * nb / float(NLIST) * (64 if pq else D_ * 4) * NQ
print("%s nb=%d: %.3f ms/step = %.0f QPS; %s %.3f ms = %.0f GB/s algorithmic (%.1f%% of 8 TB/s); select_k_kernel %.3f ms"
      % (kind, nb, dt * 1e3, NQ / dt, kname, ms, by / (ms * 1e-3) / 1e9, by / (ms * 1e-3) / 8e12 * 100, sel_ms), flush=True)
D, I = Dd.cpu().numpy(), Id.cpu().numpy()
assert (np.diff(D, axis=1) >= 0).all() and (I >= 0).all() and (I < nb).all()
assert all(len(set(r)) == K for r in I[:200])
# ---- sample parity against the oracle on the probed lists
sel = np.random.RandomState(3).choice(NQ, nsample, replace=False)
cent = idx.get_centroids()
pqc = idx.get_pq_centroids() if pq else None
Dq, Iq = idx.quantizer_search(xq[sel], NPROBE)
sizes = np.zeros(NLIST, dtype=np.uint32)
codes, ids = [], []
t0 = time.time()
for l in np.unique(Iq):
    sizes[l] = idx.get_list_size(int(l))
    codes.append(idx.get_list_codes(int(l)))
    ids.append(idx.get_list_ids(int(l)))
codes, ids = np.concatenate(codes), np.concatenate(ids)
Do, Io, cD, cI = Oracle.ivf_search(1 if pq else 0, METRIC_L2, cent, sizes, codes, ids, xq[sel], NPROBE, K, M=64 if pq else 0, pq=pqc)
ok = np.array_equal(cI, Iq) and np.array_equal(cD, Dq) and np.array_equal(Io, I[sel]) and np.array_equal(Do, D[sel])
print("%s nb=%d: %d sampled queries vs the oracle on their %d probed lists (%.1f M entries read back, %.1fs): %s"
      % (kind, nb, nsample, len(np.unique(Iq)), len(ids) / 1e6, time.time() - t0, "BIT-EXACT distances and labels" if ok else "MISMATCH"),
      flush=True)
assert ok
