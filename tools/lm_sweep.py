#!/usr/bin/env python
"""tools/lm_sweep.py -- round 3's f32 list-major scan (scan_mode 3), timing experiments on the bench data: FAISS_AMD_LM_P1 x
FAISS_AMD_LM_DBG.  The library reads these knobs ONCE per process and only under FAISS_AMD_EXPERIMENTS=1 (round 4): run one
combination per process, e.g.
    FAISS_AMD_EXPERIMENTS=1 FAISS_AMD_LM_P1=4 FAISS_AMD_LM_DBG=0 python tools/lm_sweep.py ivfflat 4/0 10000000
(the combination argument only labels the output; the filter path of round 4 is swept by tools/lmf_sweep.py, whose knobs
are API calls).
usage: lm_sweep.py kind "p1/dbg" [nb] [nq]"""
import os, sys, time
os.environ["FAISS_AMD_EXPERIMENTS"] = "1"  # the library reads its FAISS_AMD_* knobs only behind this gate
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import faiss_amd
from faiss_amd.datasets import synthetic_dataset, synthetic_more
kind = sys.argv[1]
combos = sys.argv[2].split(",")
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 1000000
nq = int(sys.argv[4]) if len(sys.argv) > 4 else 10000
K, NLIST, NPROBE = 100, 4096, 32
res = faiss_amd.StandardGpuResources(0)
xt, xb, xq, dmap = synthetic_dataset(128, 100000, min(nb, 1000000), 10000, seed=1338, return_map=True)
xq = xq[:nq]
dev = torch.device("cuda", 0)
xq_dev = torch.from_numpy(xq).to(dev)
idx = (faiss_amd.GpuIndexIVFPQ(res, 128, NLIST, 64, 8, faiss_amd.METRIC_L2) if kind == "ivfpq"
       else faiss_amd.GpuIndexIVFScalarQuantizer(res, 128, NLIST, faiss_amd.ScalarQuantizer.QT_8bit, faiss_amd.METRIC_L2, True) if kind == "ivfsq"
       else faiss_amd.GpuIndexIVFFlat(res, 128, NLIST, faiss_amd.METRIC_L2))
idx.train(xt); idx.add(xb)
done, chunk = len(xb), 0
proj_d = torch.from_numpy(np.ascontiguousarray(dmap[0], dtype=np.float64)).to(dev)
scale_d = torch.from_numpy(np.ascontiguousarray(dmap[1], dtype=np.float64)).to(dev)
while done < nb:  # further chunks drawn on the device (bench.py scale_leg): the host recipe costs 2 s per million rows
    chunk += 1
    n_c = min(1000000, nb - done)
    g = torch.Generator(device=dev); g.manual_seed(1338 + chunk)
    lat = torch.randn((n_c, proj_d.shape[0]), generator=g, device=dev, dtype=torch.float64)
    xbc = torch.sin(torch.matmul(lat, proj_d) * scale_d).to(torch.float32).contiguous()
    torch.cuda.synchronize()
    idx.add_ptr(n_c, xbc.data_ptr()); done += n_c
    del xbc, lat
idx.nprobe = NPROBE
idx.set_scan_mode(3)  # the f32 list-major scan these knobs belong to
Dd = torch.empty((nq, K), dtype=torch.float32, device=dev)
Id = torch.empty((nq, K), dtype=torch.int64, device=dev)
SPANS = ("ivf_lm_plan", "ivf_lm_scan_pass1", "ivf_lm_threshold", "ivf_lm_scan_pass2", "select_k_kernel")
for c in combos:
    p1, dbg = c.split("/")
    os.environ["FAISS_AMD_LM_P1"] = p1
    os.environ["FAISS_AMD_LM_DBG"] = dbg
    ov0 = idx.scan_info()[2]
    idx.search_ptr(nq, xq_dev.data_ptr(), K, Dd.data_ptr(), Id.data_ptr())
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(3):
        idx.search_ptr(nq, xq_dev.data_ptr(), K, Dd.data_ptr(), Id.data_ptr())
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 3
    ov = (idx.scan_info()[2] - ov0) / 4
    res.profile_enable(True); res.profile_reset()
    idx.search_ptr(nq, xq_dev.data_ptr(), K, Dd.data_ptr(), Id.data_ptr())
    spans = {s: res.profile_get(s) for s in SPANS}
    res.profile_enable(False)
    print("%s nb=%d nq=%d P1=%s dbg=%s: %.3f ms/step, overflow queries/search %.0f | " % (kind, nb, nq, p1, dbg, dt * 1e3, ov) +
          ", ".join("%s %.3f(%d)" % (s.replace("ivf_lm_", ""), v[0], v[1]) for s, v in spans.items()), flush=True)
