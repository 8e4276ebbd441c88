#!/usr/bin/env python
"""tools/lm_check.py -- list-major vs query-major IVF scan on the bench data (IVF4096, nprobe 32, 10k queries, k 100):
per-kernel times of both, agreement of the results, a sample against the oracle's list-major restatement.
usage: lm_check.py [kinds=ivfflat,ivfpq] [steps=5] [nb=1000000] [nq=10000]      (kinds: ivfflat, ivfpq, ivfsq = QT_8bit)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import faiss_amd
from faiss_amd.datasets import synthetic_dataset, synthetic_more

kinds = (sys.argv[1] if len(sys.argv) > 1 else "ivfflat,ivfpq").split(",")
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 1000000
nq = int(sys.argv[4]) if len(sys.argv) > 4 else 10000
K, NLIST, NPROBE = 100, 4096, 32
res = faiss_amd.StandardGpuResources(0)
xt, xb, xq, dmap = synthetic_dataset(128, 100000, min(nb, 1000000), 10000, seed=1338, return_map=True)
xq = xq[:nq]
dev = torch.device("cuda", 0)
xq_dev = torch.from_numpy(xq).to(dev)
SPANS = ("ivf_lm_plan", "ivf_lm_scan_pass1", "ivf_lm_threshold", "ivf_lm_scan_pass2", "select_k_kernel", "ivfflat_fused_kernel",
         "ivfpq_fused_kernel", "ivfsq_fused_kernel", "ivf_finish_kernel", "flat_filter_kernel", "flat_filter_kernel_max", "flat_tighten_kernel",
         "flat_rerank_kernel", "convert_f16_query")
for kind in kinds:
    t0 = time.time()
    if kind == "ivfpq":
        idx = faiss_amd.GpuIndexIVFPQ(res, 128, NLIST, 64, 8, faiss_amd.METRIC_L2)
    elif kind == "ivfsq":
        idx = faiss_amd.GpuIndexIVFScalarQuantizer(res, 128, NLIST, faiss_amd.ScalarQuantizer.QT_8bit, faiss_amd.METRIC_L2, True)
    else:
        idx = faiss_amd.GpuIndexIVFFlat(res, 128, NLIST, faiss_amd.METRIC_L2)
    idx.train(xt)
    idx.add(xb)
    done, chunk = len(xb), 0
    while done < nb:
        chunk += 1
        xbc = synthetic_more(dmap, min(1000000, nb - done), seed=1338 + chunk)
        idx.add(xbc)
        done += len(xbc)
    idx.nprobe = NPROBE
    print("%s: train+add of %d vectors %.1fs" % (kind, nb, time.time() - t0), flush=True)
    out = {}
    for mode, name in ((1, "query-major"), (2, "list-major")):
        idx.set_scan_mode(mode)
        Dd = torch.empty((nq, K), dtype=torch.float32, device=dev)
        Id = torch.empty((nq, K), dtype=torch.int64, device=dev)
        idx.search_ptr(nq, xq_dev.data_ptr(), K, Dd.data_ptr(), Id.data_ptr())
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(steps):
            idx.search_ptr(nq, xq_dev.data_ptr(), K, Dd.data_ptr(), Id.data_ptr())
        torch.cuda.synchronize()
        dt = (time.time() - t0) / steps
        res.profile_enable(True); res.profile_reset()
        idx.search_ptr(nq, xq_dev.data_ptr(), K, Dd.data_ptr(), Id.data_ptr())
        spans = {s: res.profile_get(s) for s in SPANS}
        res.profile_enable(False)
        print("%s %s nb=%d nq=%d: %.3f ms/step = %.0f QPS; scan_info %s" % (kind, name, nb, nq, dt * 1e3, nq / dt, idx.scan_info()), flush=True)
        print("    " + ", ".join("%s %.3f" % (s, v[0]) for s, v in spans.items() if v[1]), flush=True)
        out[mode] = (Dd.cpu().numpy(), Id.cpu().numpy())
    from compare import check_knn
    st = check_knn(out[2][0], out[2][1], out[1][0], out[1][1], rtol=1e-4, max_tie_frac=1.0, name="list-major vs query-major")
    print("    list-major vs query-major on all %d queries: %s" % (nq, st), flush=True)
    if nb <= 1000000:
        # a sample against the oracle (the device lists read back)
        from oracle.pyoracle import METRIC_L2, Oracle
        sizes = np.array([idx.get_list_size(l) for l in range(NLIST)], dtype=np.uint32)
        codes = np.concatenate([idx.get_list_codes(l).reshape(int(sizes[l]), -1) for l in range(NLIST) if sizes[l]])
        ids = np.concatenate([idx.get_list_ids(l) for l in range(NLIST) if sizes[l]])
        sel = np.r_[0:8, nq // 2: nq // 2 + 8]
        pq = idx.get_pq_centroids() if kind == "ivfpq" else None
        if kind == "ivfsq":
            vmin, vdiff = Oracle.sq_unpack(0, 128, idx.get_trained())
            Do, Io = Oracle.ivfsq_search(0, True, METRIC_L2, idx.get_centroids(), sizes, codes, ids, vmin, vdiff, xq[sel], NPROBE, K,
                                         arith=1)
        else:
            Do, Io, _, _ = Oracle.ivf_search(1 if kind == "ivfpq" else 0, METRIC_L2, idx.get_centroids(), sizes, codes, ids, xq[sel],
                                             NPROBE, K, M=64 if kind == "ivfpq" else 0, pq=pq, arith=1)
        check_knn(out[2][0][sel], out[2][1][sel], Do, Io, exact=True, name="list-major vs oracle")
        print("    list-major == oracle (arith 1) bit for bit on %d sampled queries" % len(sel), flush=True)
    del idx
