#!/usr/bin/env python
"""tools/hip_error_probe.py -- does any library call leave a HIP error code behind (hipGetLastError / hipPeekAtLastError)?"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import faiss_amd
from faiss_amd.datasets import synthetic_dataset
hip = ctypes.CDLL("libamdhip64.so")
hip.hipGetErrorString.restype = ctypes.c_char_p
def peek(tag):
    e = hip.hipGetLastError()
    print("%-40s hipGetLastError = %d (%s)" % (tag, e, hip.hipGetErrorString(e).decode()), flush=True)
peek("start")
res = faiss_amd.StandardGpuResources(0); peek("resources")
_, xb, xq = synthetic_dataset(64, 0, 20000, 500, seed=3)
idx = faiss_amd.GpuIndexFlatL2(res, 64); peek("index new")
idx.add(xb); peek("add")
D, I = idx.search(xq, 10); peek("search (filter path)")
idx.set_use_filter_kernel(False); idx.search(xq, 10); peek("search (exact path)")
res.setPagedSearch(min_bytes=1, page_queries=100); idx.search(xq, 10); peek("paged search")
ivf = faiss_amd.GpuIndexIVFPQ(res, 64, 16, 8, 8, 1); ivf.train(xb[:3000]); peek("ivfpq train")
ivf.add(xb); peek("ivfpq add"); ivf.nprobe = 4; ivf.search(xq, 5); peek("ivfpq search")
faiss_amd.knn_gpu(res, xq, xb, 5); peek("bfKnn")
del ivf, idx, res; peek("freed")
import torch
s = torch.cuda.Stream(); print("torch stream ok", s)
