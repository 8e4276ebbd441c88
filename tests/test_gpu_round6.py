"""GPU tests of round 6: ADVICE r5's findings (stale decoded copies after a codebook change, transactional patching of the
sweeps' copy), and the boundary additions of VERDICT r5 (caller-owned / fp16 coarse quantizer, indices options, memory
report, scalar-quantizer range statistics).  Every comparison is bit-exact unless a tolerance is written next to it.
Reference behaviour: faiss/gpu/GpuIndexIVF.cu:41-110, faiss/gpu/GpuIndexIVFPQ.cu:170-217, faiss/gpu/impl/IVFBase.cu:509-593."""
import numpy as np
import pytest

import faiss_amd
from oracle.pyoracle import METRIC_INNER_PRODUCT, METRIC_L2, synthetic_dataset

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("d,M", [(96, 32), (192, 48)])
def test_new_pq_codebook_invalidates_the_decoded_copy_of_the_lists(res, d, M):
    """the decoded-residual mode of the IVFPQ filter sweeps keeps an fp16 copy of the residuals DECODED WITH
    the codebook.  copy_pq_centroids() on an index that holds rows must drop that copy: otherwise the sweeps estimate with
    the old codebook's values while the band and the rerank use the new one, and the superset guarantee is gone."""
    nlist, k = 16, 30
    xt, xb, xq = synthetic_dataset(d, 6000, 30000, 700, seed=d + 3)
    idx = faiss_amd.GpuIndexIVFPQ(res, d, nlist, M, 8, METRIC_L2)
    idx.train(xt)
    idx.add(xb)
    idx.nprobe = 6
    idx.set_scan_mode(idx.SCAN_LIST_MAJOR)
    D0, I0 = idx.search(xq, k)  # builds the decoded copy
    assert idx.scan_info()[1] == 2 and idx.last_scan_arith() == 0
    pq = idx.get_pq_centroids()
    rs = np.random.RandomState(7)
    # a codebook that differs a lot: entries permuted inside every sub-quantizer and scaled
    pq2 = np.stack([pq[m][rs.permutation(256)] * (0.6 + 0.05 * (m % 5)) for m in range(M)]).astype(np.float32)
    idx.copy_pq_centroids(pq2)
    D1, I1 = idx.search(xq, k)
    assert idx.scan_info()[1] == 2
    idx.set_scan_mode(idx.SCAN_QUERY_MAJOR)
    Dq, Iq = idx.search(xq, k)
    assert np.array_equal(I1, Iq) and np.array_equal(D1, Dq)
    assert not np.array_equal(I0, I1)  # (the codebook really changed the answer)


# ------------------------------------------------------------------ caller-owned / fp16 coarse quantizer, indices options
def _trained_pair(res, d, nlist, seed):
    xt, xb, xq = synthetic_dataset(d, 6000, 30000, 500, seed=seed)
    cent, _ = faiss_amd.kmeans(res, xt, nlist, niter=6, seed=3)
    return xt, xb, xq, cent


@pytest.mark.parametrize("metric", [METRIC_L2, METRIC_INNER_PRODUCT])
def test_caller_owned_coarse_quantizer_shared_by_two_indexes(res, metric):
    """GpuIndexIVFFlat / GpuIndexIVFPQ(resources, coarseQuantizer, ...) (faiss/gpu/GpuIndexIVF.cu:41-70): the quantizer is the
    caller's GpuIndexFlat, not owned, may already hold its centroids (no coarse training then) and may serve several indexes."""
    d, nlist, M, k = 64, 32, 16, 25
    xt, xb, xq, cent = _trained_pair(res, d, nlist, 17)
    q = faiss_amd.GpuIndexFlat(res, d, metric)
    q.add(cent)
    a = faiss_amd.GpuIndexIVFFlat(res, d, nlist, metric, quantizer=q)
    assert a.is_trained and a.quantizer_info() == (False, False, faiss_amd.INDICES_64_BIT)
    b = faiss_amd.GpuIndexIVFPQ(res, d, nlist, M, 8, metric, quantizer=q)
    assert not b.is_trained  # the product quantizer is still missing
    b.train(xt)
    assert b.is_trained and q.ntotal == nlist and np.array_equal(q.reconstruct_n(0, nlist), cent)  # coarse level untouched
    own_a = faiss_amd.GpuIndexIVFFlat(res, d, nlist, metric)
    own_a.copy_centroids(cent)
    own_b = faiss_amd.GpuIndexIVFPQ(res, d, nlist, M, 8, metric)
    own_b.copy_centroids(cent)
    own_b.copy_pq_centroids(b.get_pq_centroids())
    for idx in (a, b, own_a, own_b):
        idx.add(xb)
        idx.nprobe = 7
    for mine, ref in ((a, own_a), (b, own_b)):
        for nq in (500, 9):  # list-major / query-major
            D, I = mine.search(xq[:nq], k)
            Dr, Ir = ref.search(xq[:nq], k)
            assert np.array_equal(I, Ir) and np.array_equal(D, Dr)
    # an EMPTY caller-owned quantizer is trained in place by the index (trainQuantizer_, GpuIndexIVF.cu:508-538)
    q2 = faiss_amd.GpuIndexFlat(res, d, metric)
    c = faiss_amd.GpuIndexIVFFlat(res, d, nlist, metric, quantizer=q2)
    assert not c.is_trained
    c.train(xt)
    assert c.is_trained and q2.ntotal == nlist
    # the indexes go first, the quantizer survives them
    del a, b, c
    Dq, Iq = q.search(xq[:5], 3)
    assert Iq.shape == (5, 3)
    # wrong shapes are refused like the reference's verifyIVFSettings_
    with pytest.raises(faiss_amd.FaissAmdError):
        faiss_amd.GpuIndexIVFFlat(res, d + 8, nlist, metric, quantizer=q)
    with pytest.raises(faiss_amd.FaissAmdError):
        faiss_amd.GpuIndexIVFFlat(res, d, nlist + 1, metric, quantizer=q)


@pytest.mark.parametrize("kind", ["flat", "pq", "sq"])
def test_fp16_coarse_quantizer(res, kind):
    """GpuIndexIVFConfig::flatConfig.useFloat16 (faiss/gpu/GpuIndexIVF.h:23-35; TestGpuIndexIVFPQ.cpp Float16Coarse): the coarse
    quantizer stores fp16 centroids.  Coarse search = the search of a GpuIndexFlat with useFloat16 over the centroids; residuals
    are taken against the ROUNDED centroids.  Checked bit for bit against an fp32-quantizer index that is handed the rounded
    centroids and that coarse search's output (add_core / search_preassigned)."""
    d, nlist, M, k, nprobe = 64, 32, 16, 20, 6
    xt, xb, xq, cent = _trained_pair(res, d, nlist, 23)
    cfg = faiss_amd.GpuIndexIVFConfig(flat_useFloat16=True)
    if kind == "flat":
        mk = lambda c: faiss_amd.GpuIndexIVFFlat(res, d, nlist, METRIC_L2, config=c)  # noqa: E731
    elif kind == "pq":
        mk = lambda c: faiss_amd.GpuIndexIVFPQ(res, d, nlist, M, 8, METRIC_L2,  # noqa: E731
                                               config=faiss_amd.GpuIndexIVFPQConfig(flat_useFloat16=True) if c else None)
    else:
        mk = lambda c: faiss_amd.GpuIndexIVFScalarQuantizer(res, d, nlist, faiss_amd.ScalarQuantizer.QT_8bit, METRIC_L2, True,  # noqa: E731
                                                            config=c)
    h = mk(cfg)
    assert h.quantizer_info()[1] is True
    h.copy_centroids(cent)
    f16 = faiss_amd.GpuIndexFlat(res, d, METRIC_L2, config=faiss_amd.GpuIndexFlatConfig(useFloat16=True))
    f16.add(cent)
    rounded = f16.reconstruct_n(0, nlist)
    assert np.array_equal(rounded, cent.astype(np.float16).astype(np.float32)) and not np.array_equal(rounded, cent)
    assert np.array_equal(h.get_centroids(), rounded)
    r = mk(None)
    r.copy_centroids(rounded)
    if kind == "pq":
        h.train(xt)
        r.copy_pq_centroids(h.get_pq_centroids())
    elif kind == "sq":
        h.train(xt)
        r.copy_trained(h.get_trained())
    h.add(xb)
    _, a1 = f16.search(xb, 1)
    r.add_core(xb, a1[:, 0])
    for l in range(nlist):
        assert np.array_equal(h.get_list_ids(l), r.get_list_ids(l)) and np.array_equal(h.get_list_codes(l), r.get_list_codes(l))
    h.nprobe = r.nprobe = nprobe
    Dc, Ic = f16.search(xq, nprobe)
    for nq in (500, 11):
        D, I = h.search(xq[:nq], k)
        Dr, Ir = r.search_preassigned(xq[:nq], k, Ic[:nq], Dc[:nq])
        assert np.array_equal(I, Ir) and np.array_equal(D, Dr)


@pytest.mark.parametrize("kind", [0, 1])
def test_indices_options(res, kind):
    """faiss/gpu/GpuIndicesOptions.h: INDICES_CPU returns the user ids (as INDICES_64_BIT does); INDICES_IVF keeps no ids at all and
    labels a result (inverted list << 32 | offset in the list) -- impl/IVFUtilsSelect2.cu:148."""
    d, nlist, M, k = 32, 16, 8, 15
    xt, xb, xq, cent = _trained_pair(res, d, nlist, 29)
    ids = np.arange(len(xb), dtype=np.int64) * 3 + 7

    def mk(opt):
        c = faiss_amd.GpuIndexIVFConfig(indicesOptions=opt)
        idx = (faiss_amd.GpuIndexIVFFlat(res, d, nlist, METRIC_L2, config=c) if kind == 0 else
               faiss_amd.GpuIndexIVFPQ(res, d, nlist, M, 8, METRIC_L2, config=faiss_amd.GpuIndexIVFPQConfig(indicesOptions=opt)))
        idx.copy_centroids(cent)
        if kind == 1:
            idx.copy_pq_centroids(pq)
        idx.add_with_ids(xb[:20000], ids[:20000])
        idx.add_with_ids(xb[20000:], ids[20000:])  # (a second add: offsets continue, relocated lists keep theirs)
        idx.nprobe = 5
        return idx

    pq = None
    if kind == 1:
        t = faiss_amd.GpuIndexIVFPQ(res, d, nlist, M, 8, METRIC_L2)
        t.train(xt)
        pq, cent = t.get_pq_centroids(), t.get_centroids()
    base = mk(faiss_amd.INDICES_64_BIT)
    D0, I0 = base.search(xq, k)
    for opt in (faiss_amd.INDICES_CPU, faiss_amd.INDICES_32_BIT):
        D, I = mk(opt).search(xq, k)
        assert np.array_equal(D, D0) and np.array_equal(I, I0)
    ivf = mk(faiss_amd.INDICES_IVF)
    assert ivf.quantizer_info()[2] == faiss_amd.INDICES_IVF
    D, I = ivf.search(xq, k)
    assert np.array_equal(D, D0)
    lists = [base.get_list_ids(l) for l in range(nlist)]
    ok = I0 >= 0
    assert np.array_equal(I >= 0, ok)
    back = np.array([lists[int(v) >> 32][int(v) & 0xffffffff] for v in I[ok]])
    # (equal distances: the order of tied results follows the label, which differs between the two id schemes: compare as sets per query)
    assert np.array_equal(np.sort(back), np.sort(I0[ok])) and (back == I0[ok]).mean() > 0.99


def test_memory_info_follows_the_indexes(res):
    """StandardGpuResources::getMemoryInfo (faiss/gpu/StandardGpuResources.cpp:676): the library's own device allocations"""
    import gc
    gc.collect()
    m0 = res.getMemoryInfo()
    d, nb = 64, 200000
    _, xb, _ = synthetic_dataset(d, 0, nb, 1, seed=3)
    idx = faiss_amd.GpuIndexFlatL2(res, d)
    idx.add(xb)
    m1 = res.getMemoryInfo()
    assert m1["bytes"] - m0["bytes"] >= nb * d * 4 and m1["allocations"] > m0["allocations"]
    assert m1["peak_bytes"] >= m1["bytes"] and m1["device_total"] > 200e9 and m1["temp_memory"] >= (64 << 20)
    assert m1["device_free"] < m1["device_total"]
    del idx
    gc.collect()
    m2 = res.getMemoryInfo()
    assert m2["bytes"] <= m0["bytes"] + (1 << 20)


@pytest.mark.parametrize("rangestat,arg", [(1, 2.5), (2, 0.02), (3, 0.0)])
@pytest.mark.parametrize("by_residual", [False, True])
def test_ivfsq_trains_every_range_statistic(res, rangestat, arg, by_residual):
    """GpuIndexIVFScalarQuantizer.train with RS_meanstd / RS_quantiles / RS_optim: `trained` is what
    faiss::ScalarQuantizer::train gives on the (residual) training vectors -- the host restatement is pinned byte for byte on the
    compiled reference by tests/test_oracle_cpu.py; here the index must feed it the right rows and encode / search with the result."""
    from faiss_amd import ScalarQuantizer as SQ
    d, nlist = 32, 16
    xt, xb, xq = synthetic_dataset(d, 5000, 20000, 200, seed=61)
    for qtype in (SQ.QT_8bit, SQ.QT_4bit_uniform, SQ.QT_6bit):
        idx = faiss_amd.GpuIndexIVFScalarQuantizer(res, d, nlist, qtype, METRIC_L2, by_residual)
        idx.set_rangestat(rangestat, arg)
        idx.train(xt)
        rows = xt
        if by_residual:
            cent = idx.get_centroids()
            _, lab = idx.quantizer_search(xt, 1)
            rows = (xt - cent[lab[:, 0]]).astype(np.float32)
        want = faiss_amd.sq_train_rangestat(qtype, rangestat, arg, rows)
        assert np.array_equal(idx.get_trained().view(np.uint32), want.view(np.uint32))
        idx.add(xb)
        idx.nprobe = 4
        D, I = idx.search(xq, 10)
        assert (I[:, 0] >= 0).all() and np.isfinite(D[:, 0]).all()
        # the codes are ScalarQuantizer::compute_codes with that range (oracle restatement, pinned on the reference elsewhere)
        from oracle.pyoracle import Oracle
        vmin, vdiff = Oracle.sq_unpack(qtype, d, idx.get_trained())
        cent = idx.get_centroids()
        lab = Oracle.ivf_assign(METRIC_L2, cent, xb)
        codes = Oracle.sq_encode(qtype, xb, vmin, vdiff, lab if by_residual else None, cent if by_residual else None)
        got = np.empty_like(codes)
        for l in range(nlist):
            ids = idx.get_list_ids(l)
            if len(ids):
                got[ids] = idx.get_list_codes(l)
        assert np.array_equal(got, codes)


def test_parameter_space_initialize_and_explore(res):
    """GpuParameterSpace.initialize / explore of the Python mirror (faiss/gpu/GpuAutoTune.cpp:33-77): nprobe = powers of two below nlist;
    the optimal operating points reach the exhaustive probe's recall and are sorted by it."""
    d, nlist = 32, 64
    xt, xb, xq = synthetic_dataset(d, 4000, 30000, 300, seed=67)
    idx = faiss_amd.GpuIndexIVFFlat(res, d, nlist, METRIC_L2)
    idx.train(xt)
    idx.add(xb)
    flat = faiss_amd.GpuIndexFlatL2(res, d)
    flat.add(xb)
    _, gt = flat.search(xq, 1)
    ps = faiss_amd.GpuParameterSpace()
    assert ps.initialize(idx) == {"nprobe": [1, 2, 4, 8, 16, 32]}
    pts = ps.explore(idx, xq, 10, gt[:, 0])
    perfs = [p[0] for p in pts]
    assert perfs == sorted(perfs) and perfs[-1] > 0.97 and len(pts) >= 3


# ------------------------------------------------------------------ small databases in one launch (the IVF coarse quantizer)
@pytest.mark.parametrize("metric", [METRIC_L2, METRIC_INNER_PRODUCT])
@pytest.mark.parametrize("d,nb,k,nq", [(128, 4096, 32, 1500), (128, 4096, 64, 333), (64, 4000, 32, 700), (100, 2100, 16, 40),
                                       (128, 3000, 1, 129), (32, 4096, 8, 2500)])
def test_small_database_in_one_launch_equals_the_general_path(res, metric, d, nb, k, nq):
    """flat_small_fused_kernel: a flat search over <= 4096 rows with k <= 64 -- what the coarse quantizer of
    an IVF4096 index runs -- in one launch: maxima pass, threshold, collect pass, exact re-rank, ordering.  Bit-identical to the
    general launches and to the oracle (ties by label included), also for rows / queries that are not multiples of 32."""
    from oracle.pyoracle import Oracle, integer_dataset
    _, xb, xq = synthetic_dataset(d, 0, nb, nq, seed=d + k)
    idx = faiss_amd.GpuIndexFlat(res, d, metric)
    idx.add(xb)
    idx.set_use_filter_kernel(True, 2048)  # (what GpuIndexIVF sets on its quantizer)
    idx.set_small_fused(True)
    D1, I1 = idx.search(xq, k)
    used, novf = idx.filter_stats()
    assert used and novf == 0
    idx.set_small_fused(False)
    D0, I0 = idx.search(xq, k)
    assert np.array_equal(I1, I0) and np.array_equal(D1, D0)
    Do, Io = Oracle.flat_search(metric, xb, xq[:64], k)
    assert np.array_equal(I1[:64], Io) and np.array_equal(D1[:64], Do)
    # many exact ties (integer data): more rows inside the band than a candidate list holds -> those queries take the exact scan
    xbi, xqi = integer_dataset(d, nb, 200, seed=3, hi=3)
    ti = faiss_amd.GpuIndexFlat(res, d, metric)
    ti.add(xbi)
    ti.set_use_filter_kernel(True, 2048)
    Dt, It = ti.search(xqi, k)
    Dto, Ito = Oracle.flat_search(metric, xbi, xqi, k)
    assert np.array_equal(It, Ito) and np.array_equal(Dt, Dto)


def test_ivf_search_with_the_one_launch_coarse_quantizer(res):
    """the IVF legs of the bench shape in small: coarse quantization through the one-launch kernel or the general launches --
    the same lists probed, the same results"""
    d, nlist, M, k = 128, 4096, 64, 20
    xt, xb, xq = synthetic_dataset(d, 20000, 60000, 2500, seed=71)
    cent, _ = faiss_amd.kmeans(res, xt, nlist, niter=2, seed=3)
    idx = faiss_amd.GpuIndexIVFFlat(res, d, nlist, METRIC_L2)
    idx.copy_centroids(cent)
    idx.add(xb)
    idx.nprobe = 32
    idx.set_small_fused(True)
    D1, I1 = idx.search(xq, k)
    Dc1, Ic1 = idx.quantizer_search(xq, 32)
    idx.set_small_fused(False)
    D0, I0 = idx.search(xq, k)
    Dc0, Ic0 = idx.quantizer_search(xq, 32)
    assert np.array_equal(Ic1, Ic0) and np.array_equal(Dc1, Dc0)
    assert np.array_equal(I1, I0) and np.array_equal(D1, D0)


@pytest.mark.parametrize("metric", [METRIC_L2, METRIC_INNER_PRODUCT])
def test_coarse_quantizer_overflow_joins_the_redo_set_of_the_search(res, metric):
    """Behind the filter sweeps the one-launch coarse quantizer does not read its overflow count back (a host round trip in the
    middle of every search): a query it cannot serve -- more centroids inside the band of its nprobe-th best than a candidate list
    holds, a query outside the fp16 range -- gets labels of -1 and a flag, the bound kernel puts flagged queries into the search's
    redo set, and the redo runs the coarse quantizer's general path first.  300 identical centroids (every query near them
    overflows), a few queries beyond the fp16 range; results equal the query-major scan's (which quantizes at once) bit for bit."""
    d, nlist, nb, nq, k, nprobe = 128, 2048, 50000, 3000, 25, 16
    xt, xb, xq = synthetic_dataset(d, 12000, nb, nq, seed=91)
    cent, _ = faiss_amd.kmeans(res, xt, nlist, niter=2, seed=5)
    cent = cent.copy()
    cent[100:400] = cent[100]  # ties: the band of the 16th best centroid of a query near cent[100] holds 300 rows
    xq = xq.copy()
    xq[:40] = cent[100] + 0.01 * xq[:40]   # queries that overflow the coarse quantizer's candidate lists
    xq[40:44] *= 3.0e4                      # beyond the fp16 range (flagged by both stages)
    idx = faiss_amd.GpuIndexIVFFlat(res, d, nlist, metric)
    idx.copy_centroids(cent)
    idx.add(xb)
    idx.nprobe = nprobe
    idx.set_scan_mode(idx.SCAN_QUERY_MAJOR)
    Dq, Iq = idx.search(xq, k)
    Dcq, Icq = idx.quantizer_search(xq, nprobe)
    idx.set_scan_mode(idx.SCAN_LIST_MAJOR)
    for fused in (True, False):
        idx.set_small_fused(fused)
        D, I = idx.search(xq, k)
        assert idx.scan_info()[1] == 2
        assert np.array_equal(I, Iq) and np.array_equal(D, Dq), fused
        if fused:
            assert idx.scan_info()[2] >= 40  # the handed-back queries went through the redo
    Dc, Ic = idx.quantizer_search(xq, nprobe)  # (the stand-alone search of the quantizer redoes its overflow at once)
    assert np.array_equal(Ic, Icq) and np.array_equal(Dc, Dcq)


@pytest.mark.parametrize("kind,metric", [(0, METRIC_L2), (0, METRIC_INNER_PRODUCT), (2, METRIC_L2)])
def test_lock_step_pair_sweeps_return_the_same_bits(res, kind, metric):
    """ivf_lm_filter.hip PAIR: two-wave workgroups walk the query groups of a (list, row chunk) in lock-step, or
    split an item's rows when it has no sibling.  Few lists + many queries: every list is probed by several groups of 96 queries;
    uneven list lengths and a batch that is not a multiple of anything.  Same results as a free-running wavefront per item and as
    the query-major scan; also with sampling of sweep 1 forced on (split rows at granule boundaries)."""
    d, nlist, nb, nq, k = 128, 24, 90000, 3001, 40
    xt, xb, xq = synthetic_dataset(d, 6000, nb, nq, seed=73)
    if kind == 0:
        idx = faiss_amd.GpuIndexIVFFlat(res, d, nlist, metric)
    else:
        idx = faiss_amd.GpuIndexIVFScalarQuantizer(res, d, nlist, faiss_amd.ScalarQuantizer.QT_8bit, metric, True)
    idx.train(xt)
    idx.add(xb)
    idx.nprobe = 6
    idx.set_scan_mode(idx.SCAN_QUERY_MAJOR)
    Dq, Iq = idx.search(xq, k)
    idx.set_scan_mode(idx.SCAN_LIST_MAJOR)
    for sampling in (0, 2, -1):
        idx.set_lmf_sampling(sampling)
        for on in (2, 1, 0):
            idx.set_lmf_pair(on)
            D, I = idx.search(xq, k)
            assert idx.scan_info()[1] == 2 and idx.last_scan_arith() == 0
            assert np.array_equal(I, Iq) and np.array_equal(D, Dq), (sampling, on)


# ------------------------------------------------------------------ randomized differential test of the rebuilt candidate path
def _fuzz_cases():
    rng = np.random.default_rng(20260930)
    cases = []
    for i in range(14):
        kind = int(rng.integers(0, 3))  # 0 IVFFlat, 1 IVFPQ, 2 IVFSQ8
        metric = METRIC_L2 if rng.random() < 0.6 else METRIC_INNER_PRODUCT
        d = int(rng.choice([32, 64, 96, 128])) if kind != 0 else int(rng.choice([24, 64, 100, 128, 200]))
        nlist = int(rng.choice([8, 20, 64, 200]))
        nb = int(rng.choice([20000, 60000, 130000]))
        nq = int(rng.choice([2049, 2500, 4100]))
        k = int(rng.choice([1, 10, 100, 200]))
        nprobe = int(min(nlist, rng.choice([1, 4, 9, 32])))
        cap = int(rng.choice([0, 0, 300, 1200]))  # a small candidate room forces flush overflow + the redo path
        cases.append((i, kind, metric, d, nlist, nb, nq, k, nprobe, cap))
    return cases


@pytest.mark.parametrize("case", _fuzz_cases(), ids=lambda c: "case%d-kind%d" % (c[0], c[1]))
def test_filter_path_equals_the_query_major_scan_on_random_shapes(res, case):
    """The dense pass / flush of sweep 2 were rebuilt in round 6 (a record per lane, atomics issued in batches): random index
    types, metrics, dimensions, list counts, batch sizes, k, nprobe and candidate rooms (small rooms: parked candidates beyond a
    query's segment are dropped, the query is redone) -- the list-major search behind the f16 filter returns the bits of the
    query-major scan every time, whatever the sampling of sweep 1."""
    i, kind, metric, d, nlist, nb, nq, k, nprobe, cap = case
    xt, xb, xq = synthetic_dataset(d, 8000, nb, nq, seed=200 + i)
    if kind == 0:
        idx = faiss_amd.GpuIndexIVFFlat(res, d, nlist, metric)
    elif kind == 1:
        idx = faiss_amd.GpuIndexIVFPQ(res, d, nlist, d // 2 if d % 2 == 0 and d <= 128 else d // 4, 8, metric)
    else:
        idx = faiss_amd.GpuIndexIVFScalarQuantizer(res, d, nlist, faiss_amd.ScalarQuantizer.QT_8bit, metric, True)
    idx.train(xt)
    idx.add(xb)
    idx.nprobe = nprobe
    idx.set_scan_mode(idx.SCAN_QUERY_MAJOR)
    Dq, Iq = idx.search(xq, k)
    idx.set_scan_mode(idx.SCAN_LIST_MAJOR)
    if cap:
        idx.set_lmf_tuning(0, 0, cap, 0)
    for sampling in (0, -1):
        idx.set_lmf_sampling(sampling)
        D, I = idx.search(xq, k)
        assert idx.scan_info()[1] == 2
        assert np.array_equal(I, Iq) and np.array_equal(D, Dq), (case, sampling)
