"""Randomised shapes for the list-major scan behind the f16 filter (faiss_amd/csrc/ivf_lm_filter.hip): for every drawn
(index type, metric, d, M, nlist, nb, nq, nprobe, k, selector) the list-major search must return, on ALL queries, the very
bits of the query-major search (the filter only selects, the survivors are re-derived with the query-major
arithmetic) -- and a sample of it the bits of the oracle (orc_ivf_search_ex, arith 0).  The draws cover what the
hand-picked cases of test_gpu_listmajor.py do not combine: ragged list lengths (clustered data, empty lists), one to
several row chunks and query groups per list, 1 ... 3 occupied query blocks per work item, fewer work items than XCD
queues, d on both sides of every kernel instantiation (16 ... 512), k from 1 to 600.
Reference being replaced: faiss/gpu/impl/IVFInterleaved.cuh:33-224, PQScanMultiPassNoPrecomputed-inl.cuh:173-270."""
import numpy as np
import pytest

import faiss_amd
from compare import check_knn
from oracle.pyoracle import METRIC_INNER_PRODUCT, METRIC_L2, Oracle

pytestmark = pytest.mark.gpu


def _draw(seed):
    r = np.random.RandomState(1000 + seed)
    kind = int(r.randint(0, 2))
    metric = METRIC_L2 if r.rand() < 0.7 else METRIC_INNER_PRODUCT
    if kind == 0:
        d = int(r.choice([16, 24, 40, 64, 96, 100, 128, 130, 192, 256, 300, 384, 448, 512]))
        M = 0
    else:
        dsub = int(r.choice([1, 2, 4, 8, 16]))
        M = int(r.choice([m for m in (4, 8, 12, 16, 24, 32, 48, 64) if m * dsub <= 128 and (m * dsub) % 16 == 0]))
        d = M * dsub
    nlist = int(r.choice([1, 3, 8, 17, 64, 200]))
    nb = int(r.choice([300, 2000, 9000, 30000, 120000]))
    nq = int(r.choice([1, 31, 33, 97, 300, 1500, 4000]))
    nprobe = int(min(nlist, r.choice([1, 2, 5, 16, 64])))
    k = int(r.choice([1, 7, 100, 600]))
    sel = int(r.randint(0, 3)) if r.rand() < 0.3 else 0
    return kind, metric, d, M, nlist, nb, nq, nprobe, k, sel


@pytest.mark.parametrize("seed", range(160))
def test_list_major_filter_equals_query_major_on_random_shapes(res, seed):
    kind, metric, d, M, nlist, nb, nq, nprobe, k, sel = _draw(seed)
    r = np.random.RandomState(seed)
    # clustered data: list lengths from empty to several row chunks
    nc = max(1, nlist // 2)
    centers = r.randn(nc, d).astype("float32") * 2
    xb = (centers[r.randint(0, nc, nb)] + r.randn(nb, d).astype("float32") * 0.7).astype("float32")
    xq = (centers[r.randint(0, nc, nq)] + r.randn(nq, d).astype("float32") * 0.7).astype("float32")
    cent = (centers[r.randint(0, nc, nlist)] + r.randn(nlist, d).astype("float32") * 0.5).astype("float32")
    pq = None
    if kind == 0:
        idx = faiss_amd.GpuIndexIVFFlat(res, d, nlist, metric)
    else:
        idx = faiss_amd.GpuIndexIVFPQ(res, d, nlist, M, 8, metric)
        pq = (r.rand(M, 256, d // M).astype("float32") - 0.5) * 1.5
        idx.copy_pq_centroids(pq)
    idx.copy_centroids(cent)
    ids = r.permutation(10 * nb)[:nb].astype("int64")
    idx.add_with_ids(xb, ids)
    idx.nprobe = nprobe
    params = None
    if sel == 1:
        params = faiss_amd.SearchParametersIVF(nprobe=nprobe, sel=faiss_amd.IDSelectorRange(int(2 * nb), int(7 * nb)))
    elif sel == 2:
        params = faiss_amd.SearchParametersIVF(nprobe=nprobe, sel=faiss_amd.IDSelectorBatch(ids[::3]))
    idx.set_scan_mode(idx.SCAN_QUERY_MAJOR)
    D0, I0 = idx.search(xq, k, params=params)
    assert idx.scan_info()[1] == 1
    idx.set_scan_mode(idx.SCAN_LIST_MAJOR)
    D1, I1 = idx.search(xq, k, params=params)
    assert idx.scan_info()[1] == 2 and idx.last_scan_arith() == 0
    desc = "kind %d metric %d d %d M %d nlist %d nb %d nq %d nprobe %d k %d sel %d" % (kind, metric, d, M, nlist, nb, nq, nprobe, k, sel)
    assert np.array_equal(I1, I0), desc
    assert np.array_equal(D1, D0), desc
    if sel == 0:
        s = np.r_[0:min(nq, 24)]
        sizes, codes, lids, _ = Oracle.build_ivf_lists(kind, metric, cent, xb, pq=pq, ids=ids)
        Do, Io, _, _ = Oracle.ivf_search(kind, metric, cent, sizes, codes, lids, xq[s], nprobe, k, M=M, pq=pq, arith=0)
        check_knn(D1[s], I1[s], Do, Io, exact=True, name="fuzz " + desc)
