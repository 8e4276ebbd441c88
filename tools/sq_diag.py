"""diagnostic: GpuIndexIVFScalarQuantizer vs oracle vs reference on one configuration (GPU box)"""
import os, sys
import numpy as np
import torch
torch.cuda.init()
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import faiss_amd
from oracle.pyoracle import Oracle, Ref, synthetic_dataset
qt, metric = 5, 0
res = faiss_amd.StandardGpuResources(0)
for byres in (False, True):
    d, nlist, nb, nq, nprobe, k = 72, 32, 15000, 200, 8, 20
    xt, xb, xq = synthetic_dataset(d, 5000, nb, nq, seed=77 + qt)
    sc = 255.0 / max(xt.max(), xb.max(), xq.max())
    xt, xb, xq = (np.floor(np.abs(v) * sc).astype(np.float32) for v in (xt, xb, xq))
    idx = faiss_amd.GpuIndexIVFScalarQuantizer(res, d, nlist, qt, metric, byres)
    idx.train(xt); idx.add(xb); idx.nprobe = nprobe
    D, I = idx.search(xq, k)
    cent = idx.get_centroids()
    sizes = np.array([idx.get_list_size(l) for l in range(nlist)], dtype=np.uint32)
    codes = np.concatenate([idx.get_list_codes(l) for l in range(nlist)], axis=0)
    ids = np.concatenate([idx.get_list_ids(l) for l in range(nlist)])
    z = np.zeros(d, np.float32)
    Do, Io = Oracle.ivfsq_search(qt, byres, metric, cent, sizes, codes, ids, z, z, xq, nprobe, k)
    print("byres", byres, "gpu vs oracle: D equal", np.array_equal(D, Do), "I equal", np.array_equal(I, Io), "maxabs", np.abs(D - Do).max())
    ref = Ref.ivfsq(d, nlist, qt, metric, byres)
    ref.set_sq_trained(cent, idx.get_trained())
    ref.add(xb); ref.set_nprobe(nprobe)
    Dr, Ir = ref.search(xq, k)
    rel = np.abs(D - Dr) / np.maximum(np.abs(Dr), 1e-30)
    r, c = np.unravel_index(np.argmax(rel), rel.shape)
    print("  gpu vs ref: max rel", rel.max(), "at", r, c, D[r, :4], Dr[r, :4], I[r, :4], Ir[r, :4], "I equal frac", (I == Ir).mean())
    relo = np.abs(Do - Dr) / np.maximum(np.abs(Dr), 1e-30)
    print("  oracle vs ref: max rel", relo.max(), "I equal frac", (Io == Ir).mean())
    cs = np.sort(np.linalg.norm(cent, axis=1))
    print("  centroid norms min/max", cs[0], cs[-1], "list sizes min/max", sizes.min(), sizes.max())
