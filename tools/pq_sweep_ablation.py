#!/usr/bin/env python
"""tools/pq_sweep_ablation.py -- which unit of the IVFPQ filter sweeps do the others wait for?  ("or a committed A/B
that shows which unit refused to overlap".)  Runs lib/variants/libfaiss_amd_lmf_ablate.so (`make -C faiss_amd/csrc variants`), whose
`ivf_lmf_pq_kernel<..., FG, ABL>` instantiations take units out of the sweep (ABL = a mask, see BITS below and ivf_lm_filter.hip),
chosen per launch by FAISS_AMD_LMF_ABLATE.  Results of
an ablated search are WRONG by construction; only the spans of the two sweeps are read.  IVF4096,PQ64, nprobe 32, 10 000 queries,
k = 100; sizes in millions as arguments (default 10)."""
import os
import sys

os.environ["FAISS_AMD_EXPERIMENTS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402,F401
import torch  # noqa: E402

torch.cuda.init()
import faiss_amd  # noqa: E402

faiss_amd.LIB_PATH = os.path.join(os.path.dirname(faiss_amd.LIB_PATH), "variants", "libfaiss_amd_lmf_ablate.so")
from faiss_amd.datasets import synthetic_dataset, synthetic_more_device  # noqa: E402

BITS = {1: "no gathers", 2: "3 of 24 MFMAs", 4: "no epilogue", 8: "conflict-free gathers", 16: "no code loads", 32: "NaN thresholds (no hits)",
        64: "no row-norm term", 128: "no flush", 256: "no dense pass", 512: "uncontended flush atomics"}
MASKS = [0, 1, 2, 4, 8, 16, 32, 64, 128, 256, 4 + 1, 4 + 2, 4 + 16, 4 + 1 + 2, 4 + 1 + 16, 4 + 2 + 16, 4 + 1 + 2 + 16, 32 + 1, 32 + 2, 32 + 16, 32 + 64, 512]
NAMES = {m: " + ".join(BITS[b] for b in BITS if m & b) or "the sweep as shipped" for m in MASKS}
sizes = [int(a) for a in sys.argv[1:]] or [10]
res = faiss_amd.StandardGpuResources(0)
dev = torch.device("cuda", 0)
xt, xb, xq, dmap = synthetic_dataset(128, 100000, 1000000, 10000, seed=1338, return_map=True)
xq_dev = torch.from_numpy(xq).to(dev)
D = torch.empty((10000, 100), dtype=torch.float32, device=dev)
I = torch.empty((10000, 100), dtype=torch.int64, device=dev)
idx = faiss_amd.GpuIndexIVFPQ(res, 128, 4096, 64, 8, faiss_amd.METRIC_L2)
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
    print("nb = %dM" % mb, flush=True)
    for rep in range(2):
        for abl in MASKS:
            os.environ["FAISS_AMD_LMF_ABLATE"] = str(abl)
            s1, s2 = [], []
            for it in range(4):
                res.profile_enable(True)
                res.profile_reset()
                idx.search_ptr(10000, xq_dev.data_ptr(), 100, D.data_ptr(), I.data_ptr())
                a, b = res.profile_get("ivf_lmf_sweep_min")[0], res.profile_get("ivf_lmf_sweep_collect")[0]
                res.profile_enable(False)
                if it:
                    s1.append(a)
                    s2.append(b)
            print("   run %d  %3d %-58s sweep 1 %.3f ms   sweep 2 %.3f ms   scan %s" % (rep, abl, NAMES[abl], min(s1), min(s2), idx.scan_info()),
                  flush=True)
os.environ["FAISS_AMD_LMF_ABLATE"] = "0"
