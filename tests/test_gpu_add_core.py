"""GPU tests of GpuIndexIVF::add_core -- add with a caller-supplied inverted-list assignment (faiss/gpu/GpuIndexIVF.cu:
321-356; contrib/ivf_tools.py add_preassigned; reference test faiss/gpu/test/test_gpu_index.py:23-90, where the
assignment deliberately comes from a different quantizer).  Lists must hold exactly what IndexIVF::add_core puts there:
entries in insertion order, codes encoded against the GIVEN list's centroid, out-of-range assignments left out but counted
in ntotal."""
import numpy as np
import pytest

import faiss_amd
from compare import check_knn
from faiss_amd import ScalarQuantizer as SQ
from oracle.pyoracle import METRIC_INNER_PRODUCT, METRIC_L2, Oracle, Ref, synthetic_dataset

pytestmark = pytest.mark.gpu


def _alt_assign(nb, nlist, seed, out_of_range=True):
    rs = np.random.RandomState(seed)
    a = rs.randint(0, nlist, nb).astype(np.int64)
    a[rs.permutation(nb)[:7]] = -1      # "no list": left out, still counted (IndexIVF::add_core does the same)
    if out_of_range:
        a[rs.permutation(nb)[:3]] = nlist   # beyond the lists: the same here (the reference's CPU index aborts on it)
    return a


@pytest.mark.parametrize("kind,metric", [(0, METRIC_L2), (1, METRIC_L2), (1, METRIC_INNER_PRODUCT)])
def test_add_core_builds_the_assigned_lists(res, kind, metric):
    d, nlist, nb, M = 32, 64, 9000, 8
    xt, xb, xq = synthetic_dataset(d, 3000, nb, 50, seed=12)
    cent, _ = faiss_amd.kmeans(res, xt, nlist, niter=4, seed=3)
    pq = None
    if kind == 0:
        idx = faiss_amd.GpuIndexIVFFlat(res, d, nlist, metric)
    else:
        idx = faiss_amd.GpuIndexIVFPQ(res, d, nlist, M, 8, metric)
        pq = (np.random.RandomState(7).rand(M, 256, d // M).astype("float32") - 0.5) * 0.4
        idx.copy_pq_centroids(pq)
    idx.copy_centroids(cent)
    assign = _alt_assign(nb, nlist, 5)
    ids = np.random.RandomState(1).permutation(nb).astype(np.int64) + 50
    # two calls (the second without ids: sequential from ntotal), like incremental adds
    half = nb // 2
    idx.add_core(xb[:half], assign[:half], ids[:half])
    faiss_amd.add_preassigned(idx, xb[half:], assign[half:])
    want_ids = np.concatenate([ids[:half], np.arange(half, nb, dtype=np.int64)])
    ok = (assign >= 0) & (assign < nlist)
    assert idx.ntotal == nb and idx.stored_vectors == int(ok.sum())
    codes = Oracle.pq_encode(pq, cent, xb, np.where(ok, assign, 0)) if kind else None
    for l in range(nlist):
        rows = np.nonzero(assign == l)[0]
        assert idx.get_list_size(l) == len(rows)
        assert np.array_equal(idx.get_list_ids(l), want_ids[rows])
        got = idx.get_list_codes(l)
        if kind == 0:
            assert np.array_equal(got.view(np.float32).reshape(len(rows), d), xb[rows])
        else:
            assert np.array_equal(got.reshape(len(rows), M), codes[rows])
    # searching the hand-assigned lists: the oracle on the same lists
    idx.nprobe = 8
    D, I = idx.search(xq, 10)
    order = np.argsort(np.where(ok, assign, nlist), kind="stable")[: int(ok.sum())]
    sizes = np.bincount(assign[ok], minlength=nlist).astype(np.uint32)
    lc = np.ascontiguousarray(xb[order]).view(np.uint8).reshape(len(order), -1) if kind == 0 else codes[order]
    Do, Io, _, _ = Oracle.ivf_search(kind, metric, cent, sizes, lc, want_ids[order], xq, 8, 10, M=M if kind else 0, pq=pq)
    check_knn(D, I, Do, Io, exact=True, name="search after add_core")


@pytest.mark.skipif(not Ref.available(), reason="oracle/_ref not shipped")
@pytest.mark.parametrize("desc,metric", [("IVF64,Flat", METRIC_L2), ("IVF64,PQ8", METRIC_L2), ("IVF64,SQ8", METRIC_INNER_PRODUCT)])
def test_add_core_through_the_bridge_matches_reference(desc, metric):
    """the reference's own IndexIVF::add_core against the bridge index's, same assignment: byte-identical lists"""
    d, nb, nq, k = 32, 6000, 100, 10
    xt, xb, xq = synthetic_dataset(d, 3000, nb, nq, seed=13)
    cpu = Ref.index_factory(d, desc, metric)
    cpu.set_train_niter(4, 4)
    cpu.train(xt)
    bres = Ref.amd_resources(0)
    try:
        gpu = Ref.index_cpu_to_gpu(bres, cpu)  # trained, still empty
        assign = _alt_assign(nb, 64, 9, out_of_range=False)
        ids = np.random.RandomState(2).permutation(nb).astype(np.int64) * 3
        cpu.add_core(xb, assign, ids)
        gpu.add_core(xb, assign, ids)
        assert gpu.ntotal == cpu.ntotal == nb
        back = Ref.index_gpu_to_cpu(gpu)
        s0, c0, i0 = cpu.lists()
        s1, c1, i1 = back.lists()
        assert np.array_equal(s0, s1) and np.array_equal(i0, i1) and np.array_equal(c0, c1)
        cpu.set_nprobe(8)
        Dr, Ir = cpu.search(xq, k)
        D, I = gpu.search_nprobe(xq, k, 8)
        check_knn(D, I, Dr, Ir, rtol=1e-4, name=desc + " after add_core")
        del gpu, back
    finally:
        Ref.amd_resources_free(bres)
