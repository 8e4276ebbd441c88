import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
os.environ["FAISS_AMD_EXPERIMENTS"] = "1"
import numpy as np, torch
torch.cuda.init()
import faiss_amd
from oracle.pyoracle import METRIC_INNER_PRODUCT, METRIC_L2, synthetic_dataset
from test_gpu_listmajor import _build
res = faiss_amd.StandardGpuResources(0)
for (kind, metric, d, M, nlist, nb, nq, nprobe, k) in [(1, METRIC_INNER_PRODUCT, 128, 64, 64, 40000, 1100, 8, 10), (1, METRIC_L2, 128, 64, 64, 40000, 1500, 8, 100), (1, METRIC_INNER_PRODUCT, 32, 16, 128, 20000, 600, 100, 10)]:
    xt, xb, xq = synthetic_dataset(d, 4000, nb, nq, seed=nb + 10)
    idx, cent, pq = _build(res, kind, metric, d, M, nlist, xt, xb)
    idx.nprobe = nprobe
    idx.set_scan_mode(1)
    D0, I0 = idx.search(xq, k)
    idx.set_scan_mode(2)
    idx.set_lmf_sampling(-1)
    tot = 0
    for rep in range(6):
        D, I = idx.search(xq, k)
        bad = np.where((I != I0).any(1) | (D != D0).any(1))[0]
        tot += len(bad)
    print("dbg", sys.argv[1:], "kind", kind, "metric", metric, "d", d, "k", k, "bad queries over 6 searches", tot, "redo", idx.scan_info()[2], flush=True)
