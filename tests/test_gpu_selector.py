"""GPU parity tests of IDSelector searches (pytest -m gpu), through the C ABI.

The rule under test (pinned on the REAL reference by tests/test_selector_cpu.py: IndexFlat::search /
IndexIVF::search with SearchParameters::sel): a search with a selector returns what a search of the selected subset
returns.  Here the device result is compared BIT-EXACTLY (distances and labels) with the oracle run on that subset --
the selected rows of a flat index, the inverted lists restricted to the selected entries -- for every selector type
(faiss/impl/IDSelector.h: Range, Array / Batch, Bitmap, All, Not, And, Or, XOr), both metrics, every kernel path a
selector touches: fp16 filter + re-rank (both geometries), exact fp32 scan, the fused IVFFlat / IVFPQ / IVFSQ scans with
the in-kernel finish, the probe-split merge and the deferred finish, the unfused IVF scans, search_preassigned,
IndexShards; and, through the reference-side bridge, the reference's own selector objects against its CPU indexes."""
import numpy as np
import pytest

import faiss_amd
from compare import check_knn
from faiss_amd import ScalarQuantizer as SQ
from oracle.pyoracle import METRIC_INNER_PRODUCT, METRIC_L2, Oracle, Ref, synthetic_dataset
from selector_cases import filter_lists, selector_cases
from test_selector_cpu import subset_flat

pytestmark = pytest.mark.gpu
SP, SPI = faiss_amd.SearchParameters, faiss_amd.SearchParametersIVF


# ------------------------------------------------------------------------------------------ flat
@pytest.mark.parametrize("metric", [METRIC_L2, METRIC_INNER_PRODUCT])
@pytest.mark.parametrize("d,nb,nq,k,f16", [
    (64, 20000, 300, 100, False),   # fp16 filter + re-rank, 4-wave geometry
    (128, 24000, 2100, 10, False),  # 8-wave geometry (d = 128, batch >= 2048)
    (40, 3000, 70, 100, False),     # exact fp32 scan (small database)
    (96, 20000, 100, 1500, False),  # k > 1024: exact scan on a large database
    (64, 20000, 200, 20, True),     # fp16 storage
])
def test_flat_selector_equals_subset_search(res, metric, d, nb, nq, k, f16):
    _, xb, xq = synthetic_dataset(d, 0, nb, nq, seed=nb + k)
    idx = faiss_amd.GpuIndexFlat(res, d, metric, config=faiss_amd.GpuIndexFlatConfig(useFloat16=True)) if f16 \
        else faiss_amd.GpuIndexFlat(res, d, metric)
    idx.add(xb)
    if f16:  # the index holds (and the oracle is fed) the fp16-rounded values
        xb, xq = xb.astype(np.float16).astype(np.float32), xq.astype(np.float16).astype(np.float32)
    labels = np.arange(nb, dtype=np.int64)
    D0, I0 = idx.search(xq, k)
    sub = np.r_[0:min(nq, 48)]
    for case in selector_cases(0, nb, seed=d):
        D, I = idx.search(xq, k, params=SP(sel=case["sel"]))
        keep = case["member"](labels)
        Do, Io = subset_flat(metric, xb, xq[sub], k, keep)
        check_knn(D[sub], I[sub], Do, Io, exact=True, name="flat selector " + case["name"])
        # every query: only selected labels, best first, padded exactly when the selection is smaller than k
        valid = I >= 0
        assert keep[I[valid]].all(), case["name"]
        assert (valid.sum(axis=1) == min(k, int(keep.sum()))).all(), case["name"]
        if case["name"] == "all":
            assert np.array_equal(D, D0) and np.array_equal(I, I0)
    # the selector leaves no trace: the next plain search is the first one again
    D1, I1 = idx.search(xq, k)
    assert np.array_equal(D1, D0) and np.array_equal(I1, I0)


def test_flat_selector_takes_the_filter_path(res):
    d, nb = 64, 30000
    _, xb, xq = synthetic_dataset(d, 0, nb, 500, seed=77)
    idx = faiss_amd.GpuIndexFlatL2(res, d)
    idx.add(xb)
    sel = faiss_amd.IDSelectorBatch(np.random.RandomState(3).permutation(nb)[: nb // 2])
    D, I = idx.search(xq, 50, params=SP(sel=sel))
    used, novf = idx.filter_stats()
    assert used and novf == 0, "a half-selected database must stay on the fp16 filter path without overflow queries"
    keep = np.array([sel.is_member(i) for i in range(nb)])
    Do, Io = subset_flat(METRIC_L2, xb, xq[:32], 50, keep)
    check_knn(D[:32], I[:32], Do, Io, exact=True, name="filter path with selector")


# ------------------------------------------------------------------------------------------ IVF
def _ivf_index(res, kind, metric, d, nlist, xt, M=8, qtype=SQ.QT_8bit):
    cent, _ = faiss_amd.kmeans(res, xt, nlist, niter=4, seed=3)
    pq = None
    if kind == 0:
        idx = faiss_amd.GpuIndexIVFFlat(res, d, nlist, metric)
        idx.copy_centroids(cent)
    elif kind == 1:
        idx = faiss_amd.GpuIndexIVFPQ(res, d, nlist, M, 8, metric)
        pq = (np.random.RandomState(7).rand(M, 256, d // M).astype("float32") - 0.5) * 0.4
        idx.copy_pq_centroids(pq)
        idx.copy_centroids(cent)
    else:
        idx = faiss_amd.GpuIndexIVFScalarQuantizer(res, d, nlist, qtype, metric, True)
        idx.train(xt)
        cent = idx.get_centroids()
    return idx, cent, pq


def _ivf_oracle(kind, metric, idx, cent, pq, M, qtype, sizes, codes, ids, xq, nprobe, k):
    if kind == 2:
        vmin, vdiff = Oracle.sq_unpack(qtype, idx.d, idx.get_trained())
        return Oracle.ivfsq_search(qtype, True, metric, cent, sizes, codes, ids, vmin, vdiff, xq, nprobe, k)
    Do, Io, _, _ = Oracle.ivf_search(kind, metric, cent, sizes, codes, ids, xq, nprobe, k, M=M if kind else 0, pq=pq)
    return Do, Io


def _gpu_lists(idx):
    sizes = np.array([idx.get_list_size(l) for l in range(idx.nlist)], dtype=np.uint32)
    codes = np.concatenate([idx.get_list_codes(l) for l in range(idx.nlist)], axis=0)
    ids = np.concatenate([idx.get_list_ids(l) for l in range(idx.nlist)])
    return sizes, codes, ids


@pytest.mark.parametrize("kind,metric,d,nq,k", [
    (0, METRIC_L2, 64, 1100, 100),           # IVFFlat, deferred finish (batch fills the chip)
    (0, METRIC_INNER_PRODUCT, 40, 60, 10),   # IVFFlat, probes split over workgroups + merge
    (1, METRIC_L2, 64, 1100, 100),           # IVFPQ (generic M), deferred finish
    (1, METRIC_INNER_PRODUCT, 64, 300, 10),  # IVFPQ, in-kernel finish / split
    (2, METRIC_L2, 64, 1100, 50),            # IVFSQ 8 bit
    (2, METRIC_INNER_PRODUCT, 48, 200, 20),
])
def test_ivf_selector_equals_filtered_lists(res, kind, metric, d, nq, k):
    nlist, nb, nprobe, M, qtype = 32, 20000, 8, 8, SQ.QT_8bit
    xt, xb, xq = synthetic_dataset(d, 4000, nb, nq, seed=nb + k + kind)
    ids = np.random.RandomState(5).permutation(nb).astype(np.int64) * 7 + 3
    idx, cent, pq = _ivf_index(res, kind, metric, d, nlist, xt, M, qtype)
    idx.add_with_ids(xb, ids)
    idx.nprobe = nprobe
    sizes, codes, lids = _gpu_lists(idx)
    D0, I0 = idx.search(xq, k)
    sub = np.r_[0:min(nq, 40)]
    for case in selector_cases(3, 3 + 7 * nb, seed=kind):
        D, I = idx.search(xq, k, params=SPI(sel=case["sel"]))
        keep = case["member"](lids)
        s2, c2, i2 = filter_lists(sizes, codes, lids, keep)
        Do, Io = _ivf_oracle(kind, metric, idx, cent, pq, M, qtype, s2, c2, i2, xq[sub], nprobe, k)
        check_knn(D[sub], I[sub], Do, Io, exact=True, name="ivf kind %d selector %s" % (kind, case["name"]))
        valid = I >= 0
        assert case["member"](I[valid]).all(), case["name"]
        if case["name"] == "all":
            assert np.array_equal(D, D0) and np.array_equal(I, I0)
        if kind != 2 and case["name"] in ("batch30", "range_and_not_batch"):
            # the unfused path (every distance as a key in HBM + select kernel) honours the selector too
            idx.set_use_fused_scan(False)
            Du, Iu = idx.search(xq[:64], k, params=SPI(sel=case["sel"]))
            idx.set_use_fused_scan(True)
            assert np.array_equal(Du, D[:64]) and np.array_equal(Iu, I[:64]), case["name"]
    # nprobe override and selector in one SearchParametersIVF; a plain SearchParameters works on an IVF index too
    case = selector_cases(3, 3 + 7 * nb, seed=kind)[2]
    D, I = idx.search(xq[:50], k, params=SPI(nprobe=3, sel=case["sel"]))
    s2, c2, i2 = filter_lists(sizes, codes, lids, case["member"](lids))
    Do, Io = _ivf_oracle(kind, metric, idx, cent, pq, M, qtype, s2, c2, i2, xq[:50], 3, k)
    check_knn(D, I, Do, Io, exact=True, name="nprobe + selector")
    D2, I2 = idx.search(xq[:50], k, params=SP(sel=case["sel"]))
    D3, I3 = idx.search(xq[:50], k, params=SPI(sel=case["sel"]))
    assert np.array_equal(D2, D3) and np.array_equal(I2, I3)
    # search_preassigned with the same parameters == search
    Dq, Iq = idx.quantizer_search(xq[:50], nprobe)
    D4, I4 = idx.search_preassigned(xq[:50], k, Iq, Dq, params=SPI(sel=case["sel"]))
    assert np.array_equal(D4, D3) and np.array_equal(I4, I3)
    # and no trace afterwards
    D1, I1 = idx.search(xq, k)
    assert np.array_equal(D1, D0) and np.array_equal(I1, I0)


@pytest.mark.parametrize("kind,metric,d,M", [(0, METRIC_L2, 128, 0), (0, METRIC_INNER_PRODUCT, 40, 0), (1, METRIC_L2, 128, 64),
                                             (1, METRIC_INNER_PRODUCT, 64, 16)])
def test_list_major_filter_scan_honours_the_selector(res, kind, metric, d, M):
    """Round 4: the list-major scan behind the f16 filter (ivf_lm_filter.hip) tests the selector's row bits in both sweeps
    (rows it excludes take no part in the bound nor in the collection), so IDSelector searches of large batches no longer
    fall back to the query-major scan.  Every selector: bit-identical to the query-major search with the same selector
    (all queries) and to the oracle on the lists restricted to the selected entries (a sample); sparse selections that
    leave fewer than k granules admit everything and still come out exact."""
    nlist, nb, nq, nprobe, k = 32, 40000, 2100, 8, 50
    xt, xb, xq = synthetic_dataset(d, 4000, nb, nq, seed=nb + d)
    ids = np.random.RandomState(5).permutation(nb).astype(np.int64) * 7 + 3
    idx, cent, pq = _ivf_index(res, kind, metric, d, nlist, xt, M)
    idx.add_with_ids(xb, ids)
    idx.nprobe = nprobe
    sizes, codes, lids = _gpu_lists(idx)
    sub = np.r_[0:32]
    for case in selector_cases(3, 3 + 7 * nb, seed=kind):
        idx.set_scan_mode(idx.SCAN_QUERY_MAJOR)
        D0, I0 = idx.search(xq, k, params=SPI(sel=case["sel"]))
        idx.set_scan_mode(idx.SCAN_LIST_MAJOR)
        D, I = idx.search(xq, k, params=SPI(sel=case["sel"]))
        assert idx.scan_info()[1] == 2 and idx.last_scan_arith() == 0, case["name"]
        assert np.array_equal(D, D0) and np.array_equal(I, I0), case["name"]
        keep = case["member"](lids)
        s2, c2, i2 = filter_lists(sizes, codes, lids, keep)
        Do, Io = _ivf_oracle(kind, metric, idx, cent, pq, M, SQ.QT_8bit, s2, c2, i2, xq[sub], nprobe, k)
        check_knn(D[sub], I[sub], Do, Io, exact=True, name="list-major filter, selector %s" % case["name"])
        assert case["member"](I[I >= 0]).all(), case["name"]
    # automatic mode: the batch is large enough for the list-major scan, with or without a selector
    idx.set_scan_mode(idx.SCAN_AUTO)
    case = selector_cases(3, 3 + 7 * nb, seed=kind)[2]
    if idx.list_major_rule(nq, nprobe, k):
        idx.search(xq, k, params=SPI(sel=case["sel"]))
        assert idx.scan_info()[1] == 2
    # the f32 list-major scan still refuses selectors
    idx.set_scan_mode(idx.SCAN_LIST_MAJOR_F32)
    with pytest.raises(faiss_amd.FaissAmdError, match="IDSelector"):
        idx.search(xq[:10], 5, params=SPI(sel=case["sel"]))


def test_ivfpq64_selector_bench_shape_kernel(res):
    """the M = 64 instantiation of the fused IVFPQ scan (the bench kernel: one v_perm per gather, rotated blocks)"""
    d, nlist, nb, nq, nprobe, k, M = 128, 64, 40000, 1100, 8, 100, 64
    xt, xb, xq = synthetic_dataset(d, 6000, nb, nq, seed=5)
    idx, cent, pq = _ivf_index(res, 1, METRIC_L2, d, nlist, xt, M)
    idx.add(xb)
    idx.nprobe = nprobe
    sizes, codes, lids = _gpu_lists(idx)
    for case in selector_cases(0, nb, seed=1)[1:7]:
        D, I = idx.search(xq, k, params=SPI(sel=case["sel"]))
        s2, c2, i2 = filter_lists(sizes, codes, lids, case["member"](lids))
        Do, Io, _, _ = Oracle.ivf_search(1, METRIC_L2, cent, s2, c2, i2, xq[:32], nprobe, k, M=M, pq=pq)
        check_knn(D[:32], I[:32], Do, Io, exact=True, name="ivfpq64 selector " + case["name"])
        assert case["member"](I[I >= 0]).all()


def test_shards_forward_search_parameters(res):
    """IndexShards hands the parameters to every shard (faiss/IndexShards.cpp:196-265): with explicit ids the sharded
    selector search equals the unsharded one; IndexReplicas refuses parameters like the reference."""
    d, nlist, nb, nq, k = 32, 16, 12000, 100, 20
    xt, xb, xq = synthetic_dataset(d, 3000, nb, nq, seed=8)
    ids = np.random.RandomState(2).permutation(nb).astype(np.int64) + 100
    cent, _ = faiss_amd.kmeans(res, xt, nlist, niter=4, seed=3)

    def make(r=res):
        i = faiss_amd.GpuIndexIVFFlat(r, d, nlist, METRIC_L2)
        i.copy_centroids(cent)
        i.nprobe = 4
        return i

    whole = make()
    whole.add_with_ids(xb, ids)
    # one resources object (stream) per shard, one host thread per shard: the shards compile and upload the SAME selector
    # objects concurrently
    shards = faiss_amd.IndexShards(d, threaded=True, successive_ids=False)
    for _ in range(3):
        shards.add_shard(make(faiss_amd.StandardGpuResources(0)))
    shards.add_with_ids(xb, ids)
    sel = faiss_amd.IDSelectorRange(2000, 9000) & ~faiss_amd.IDSelectorBatch(ids[::3])
    Dw, Iw = whole.search(xq, k, params=SPI(sel=sel))
    for _ in range(3):
        fresh = faiss_amd.IDSelectorRange(2000, 9000) & ~faiss_amd.IDSelectorBatch(ids[::3])  # (first use: upload race)
        Ds, Is = shards.search(xq, k, params=SPI(sel=fresh))
        assert np.array_equal(Iw, Is) and np.array_equal(Dw, Ds)
    rep = faiss_amd.IndexReplicas(d, threaded=False)
    rep.add_replica(make())
    with pytest.raises(faiss_amd.FaissAmdError, match="search params not supported"):
        rep.search(xq, k, params=SPI(sel=sel))


# ------------------------------------------------------------------------------------------ through the bridge
@pytest.mark.skipif(not Ref.available(), reason="oracle/_ref not shipped")
@pytest.mark.parametrize("desc,metric", [("Flat", METRIC_L2), ("Flat", METRIC_INNER_PRODUCT), ("IVF64,Flat", METRIC_L2),
                                         ("IVF64,PQ16", METRIC_L2), ("IVF64,SQ8", METRIC_INNER_PRODUCT)])
def test_reference_selectors_through_the_bridge(desc, metric):
    """faiss::IDSelector objects of the reference, handed to faiss::Index::search of a bridge index (cloned from the
    reference's CPU index), against the same call on the CPU index itself."""
    d, nb, nq, k, nprobe = 64, 20000, 200, 50, 8
    xt, xb, xq = synthetic_dataset(d, 4000, nb, nq, seed=41)
    cpu = Ref.index_factory(d, desc, metric)
    ivf = "IVF" in desc
    if ivf:
        cpu.set_train_niter(5, 6)
        cpu.train(xt)
        cpu.set_nprobe(nprobe)
    cpu.add(xb)
    bres = Ref.amd_resources(0)
    try:
        gpu = Ref.index_cpu_to_gpu(bres, cpu)
        for case in selector_cases(0, nb, seed=6):
            if case["ref"] is None:
                continue
            # (IVF indexes of the reference want SearchParametersIVF: nprobe > 0 selects that type in the shim)
            np_arg = nprobe if ivf else 0
            Dr, Ir = cpu.search_sel(xq, k, nprobe=np_arg, **case["ref"])
            D, I = gpu.search_sel(xq, k, nprobe=np_arg, **case["ref"])
            check_knn(D, I, Dr, Ir, rtol=1e-4, name="%s bridge selector %s" % (desc, case["name"]))
        if not ivf:
            # a selector type the bridge cannot translate structurally: tabulated over the row numbers
            Dr, Ir = cpu.search_sel(xq, k, kind=5, a=7, b=3)
            D, I = gpu.search_sel(xq, k, kind=5, a=7, b=3)
            check_knn(D, I, Dr, Ir, rtol=1e-4, name="bridge custom selector")
            assert (I[I >= 0] % 7 == 3).all()
        Dr, Ir = cpu.search_sel(xq, k, kind=6, a=5000, b=15000, data=np.zeros(100, np.uint8), nprobe=nprobe if ivf else 0)
        D, I = gpu.search_sel(xq, k, kind=6, a=5000, b=15000, data=np.zeros(100, np.uint8), nprobe=nprobe if ivf else 0)
        check_knn(D, I, Dr, Ir, rtol=1e-4, name="bridge xor / or / all")
        assert ((I[I >= 0] < 5000) | (I[I >= 0] >= 15000)).all()
        del gpu
    finally:
        Ref.amd_resources_free(bres)
