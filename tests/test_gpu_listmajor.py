"""GPU parity tests of the list-major IVF scans: large batches visit every inverted list once per group of the queries
probing it.  Two scans, both run by every test here (fixture lm_mode):
  * SCAN_LIST_MAJOR (faiss_amd/csrc/ivf_lm_filter.hip, round 4; IVFFlat and IVFPQ shapes it serves): f16 MFMA estimates
    with a rigorous error band select a superset of the answer, the survivors are re-derived with the arithmetic of the
    query-major scan -- results BIT-IDENTICAL to the query-major scan and to its restatement (orc_ivf_search_ex, arith 0);
  * SCAN_LIST_MAJOR_F32 (faiss_amd/csrc/ivf_listmajor.hip, round 3; also what SCAN_LIST_MAJOR falls back to for the
    shapes the filter does not serve): every distance on the f32 matrix pipe, BIT-IDENTICAL to its own restatement
    (arith 1 -- pinned on the reference's golden outputs by tests/test_oracle_cpu.py::test_ivf_list_major_restatement_
    vs_golden*), within the north-star tolerance of the query-major scan (labels identical outside near-tie groups,
    distances <= 1e-4 relative).
idx.last_scan_arith() says which restatement applies to the last search.
Reference being replaced: faiss/gpu/impl/IVFInterleaved.cuh:33-224, PQScanMultiPassNoPrecomputed-inl.cuh:173-270."""
import numpy as np
import pytest

import faiss_amd
from compare import check_knn
from oracle.pyoracle import METRIC_INNER_PRODUCT, METRIC_L2, Oracle, synthetic_dataset

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[2, 3], ids=["filter", "f32"])
def lm_mode(request):
    return request.param


def _build(res, kind, metric, d, M, nlist, xt, xb, seed=7):
    cent, _ = faiss_amd.kmeans(res, xt, nlist, niter=4, seed=3)
    pq = None
    if kind == 0:
        idx = faiss_amd.GpuIndexIVFFlat(res, d, nlist, metric)
    else:
        idx = faiss_amd.GpuIndexIVFPQ(res, d, nlist, M, 8, metric)
        pq = (np.random.RandomState(seed).rand(M, 256, d // M).astype("float32") - 0.5) * 0.4
        idx.copy_pq_centroids(pq)
    idx.copy_centroids(cent)
    idx.add(xb)
    return idx, cent, pq


@pytest.mark.parametrize("kind,metric,d,M,nlist,nb,nq,nprobe,k", [
    (0, METRIC_L2, 128, 0, 64, 40000, 1500, 8, 100),            # dpad = 128: the fully unrolled instantiation
    (0, METRIC_INNER_PRODUCT, 128, 0, 64, 40000, 700, 8, 10),
    (0, METRIC_L2, 40, 0, 16, 9000, 300, 16, 7),                # dpad < 128, every list probed, ragged tiles
    (0, METRIC_L2, 72, 0, 8, 20000, 1100, 2, 1000),             # lists of ~2500 rows: several row chunks per list, big k
    (0, METRIC_INNER_PRODUCT, 64, 0, 32, 5000, 130, 5, 2048),   # k above the rows many queries see
    (0, METRIC_L2, 64, 0, 8, 90000, 600, 3, 50),                # lists of ~11 000 rows: row chunks of 2816 rows
    (1, METRIC_L2, 64, 32, 8, 90000, 600, 3, 50),               # the same for codes
    (0, METRIC_L2, 32, 0, 128, 20000, 600, 100, 10),            # more than 64 probes: two rounds of the plan's scans
    (1, METRIC_INNER_PRODUCT, 32, 16, 128, 20000, 600, 100, 10),
    (1, METRIC_L2, 128, 64, 64, 40000, 1500, 8, 100),           # bench shape: M = 64, dsub = 2
    (1, METRIC_INNER_PRODUCT, 128, 64, 64, 40000, 1100, 8, 10),
    (1, METRIC_L2, 64, 16, 32, 30000, 400, 32, 600),            # dsub = 4, every list probed
    (1, METRIC_L2, 96, 12, 16, 5000, 1200, 5, 2048),            # dsub = 8, code chunks of 4 bytes
    (1, METRIC_L2, 32, 32, 8, 3000, 1030, 3, 5),                # dsub = 1
    (1, METRIC_INNER_PRODUCT, 36, 4, 16, 6000, 257, 4, 20),     # d = 36 -> dpad = 40: zero padding of the decoded tile, dsub = 9
    (1, METRIC_L2, 96, 48, 16, 12000, 700, 6, 30),              # three 16-byte pieces per stored row (lane pairs 2 + 1)
    (1, METRIC_INNER_PRODUCT, 80, 80, 8, 9000, 300, 4, 40),     # five pieces: staged byte by byte; dsub = 1
    (1, METRIC_L2, 120, 60, 16, 8000, 520, 5, 64),              # dpad = 120 < 128, M % 16 != 0 (4-byte pieces), dsub = 2
    # d > 128 (round 4, filter scan only): 16 / 24 / 32 k-steps per row, two or one query block per work item
    (0, METRIC_L2, 256, 0, 32, 20000, 900, 8, 50),
    (0, METRIC_INNER_PRODUCT, 200, 0, 16, 8000, 300, 4, 20),    # shadow rows padded to 256 halfs
    (0, METRIC_L2, 136, 0, 16, 9000, 700, 5, 100),
    (0, METRIC_L2, 384, 0, 16, 8000, 500, 6, 100),
    (0, METRIC_INNER_PRODUCT, 300, 0, 8, 6000, 200, 3, 1000),   # padded to 384
    (0, METRIC_L2, 512, 0, 16, 6000, 300, 4, 10),
    (0, METRIC_L2, 420, 0, 8, 12000, 640, 2, 30),               # padded to 512, lists of ~1500 rows
])
def test_list_major_scan_matches_oracle_and_query_major(res, lm_mode, kind, metric, d, M, nlist, nb, nq, nprobe, k):
    if d > 128 and lm_mode == 3:
        pytest.skip("the f32 list-major scan serves d <= 128")
    xt, xb, xq = synthetic_dataset(d, 4000, nb, nq, seed=nb + k)
    idx, cent, pq = _build(res, kind, metric, d, M, nlist, xt, xb)
    idx.nprobe = nprobe
    idx.set_scan_mode(idx.SCAN_QUERY_MAJOR)
    D0, I0 = idx.search(xq, k)
    assert idx.scan_info()[1] == 1
    idx.set_scan_mode(lm_mode)
    D, I = idx.search(xq, k)
    arith = idx.last_scan_arith()
    assert idx.scan_info()[1] == 2 and (arith == 1 or lm_mode == idx.SCAN_LIST_MAJOR)
    if arith == 0:
        # behind the f16 filter: the very bits of the query-major scan, for every query
        assert np.array_equal(I, I0) and np.array_equal(D, D0)
    # the two scans agree to rounding
    check_knn(D, I, D0, I0, rtol=1e-4, name="list-major vs query-major")
    sel = np.r_[0:min(nq, 48)]
    sizes, codes, ids, _ = Oracle.build_ivf_lists(kind, metric, cent, xb, pq=pq)
    Do, Io, _, _ = Oracle.ivf_search(kind, metric, cent, sizes, codes, ids, xq[sel], nprobe, k, M=M, pq=pq, arith=arith)
    check_knn(D[sel], I[sel], Do, Io, exact=True, name="list-major vs oracle")
    # run to run: the order in which workgroups append candidates never shows
    D2, I2 = idx.search(xq, k)
    assert np.array_equal(D, D2) and np.array_equal(I, I2)


def test_list_major_independent_of_batch_composition(res, lm_mode):
    """A query's result does not depend on which other queries share its batch (tiles, item order, pass-1 / pass-2 split
    of the lists all change with the batch)."""
    d, nlist, nb, k = 64, 32, 20000, 30
    xt, xb, xq = synthetic_dataset(d, 4000, nb, 900, seed=11)
    idx, _, _ = _build(res, 0, METRIC_L2, d, 0, nlist, xt, xb)
    idx.nprobe = 6
    idx.set_scan_mode(lm_mode)
    D, I = idx.search(xq, k)
    D1, I1 = idx.search(xq[100:137], k)
    assert np.array_equal(D1, D[100:137]) and np.array_equal(I1, I[100:137])
    perm = np.random.RandomState(0).permutation(len(xq))
    D2, I2 = idx.search(xq[perm], k)
    assert np.array_equal(D2, D[perm]) and np.array_equal(I2, I[perm])


@pytest.mark.parametrize("kind,nnear,levels", [(0, 6000, 1), (1, 6000, 1), (0, 40000, 2), (1, 40000, 2)])
def test_list_major_overflow_rerun_is_exact(res, lm_mode, kind, nnear, levels):
    """Adversarial layout for the pass-1 bound: the nearest list of every query holds exactly k rows that are FAR away,
    the next two lists hold `nnear` rows each that are all closer -- more candidates than a segment has room for (4096).
    Those queries are redone with 16 x the room (enough for 2 x 6000 rows), and where that overflows too (2 x 40 000
    rows) with every probe in pass 1; the answer is still the oracle's bit for bit."""
    d, k, nlist = 32, 4, 4
    rs = np.random.RandomState(3)
    cent = np.zeros((nlist, d), "float32")
    cent[:, 0] = [0.0, 10.0, 20.0, 30.0]
    xq = np.zeros((40, d), "float32")       # queries sit on centroid 0
    xq[:, 1] = rs.rand(40) * 0.01
    far = np.zeros((k, d), "float32")       # list 0: |far - q|^2 ~ 900
    far[:, 2] = 30.0 + rs.rand(k)
    near1 = np.zeros((nnear, d), "float32")  # list 1 (4.95 from centroid 1, 5.05 from centroid 0): |near1 - q|^2 ~ 25.5
    near1[:, 0] = 5.05 + rs.rand(nnear) * 0.01
    near2 = np.zeros((nnear, d), "float32")  # list 2: |near2 - q|^2 ~ 226
    near2[:, 0] = 15.05 + rs.rand(nnear) * 0.01
    xb = np.concatenate([far, near1, near2]).astype("float32")
    M = 8
    if kind == 0:
        idx = faiss_amd.GpuIndexIVFFlat(res, d, nlist, METRIC_L2)
        pq = None
    else:
        idx = faiss_amd.GpuIndexIVFPQ(res, d, nlist, M, 8, METRIC_L2)
        pq = (np.random.RandomState(7).rand(M, 256, d // M).astype("float32") - 0.5) * 0.02
        idx.copy_pq_centroids(pq)
    idx.copy_centroids(cent)
    idx.add(xb)
    sizes = [idx.get_list_size(l) for l in range(nlist)]
    assert sizes[0] == k and sizes[1] == nnear and sizes[2] == nnear
    idx.nprobe = 3
    idx.set_scan_mode(lm_mode)
    before = idx.scan_info()[2]
    D, I = idx.search(xq, k)
    if kind == 0 and lm_mode == idx.SCAN_LIST_MAJOR_F32:
        assert idx.scan_info()[2] - before == levels * len(xq), "the overflow path was not exercised"
    sz, codes, ids, _ = Oracle.build_ivf_lists(kind, METRIC_L2, cent, xb, pq=pq)
    Do, Io, _, _ = Oracle.ivf_search(kind, METRIC_L2, cent, sz, codes, ids, xq, 3, k, M=M if kind else 0, pq=pq,
                                     arith=idx.last_scan_arith())
    check_knn(D, I, Do, Io, exact=True, name="overflow rerun vs oracle")


def test_list_major_edge_cases(res, lm_mode):
    """empty lists, probes without a list (-1), a NaN query, fewer than k rows in all probed lists, search_preassigned"""
    d, nlist, k = 24, 16, 12
    xt, xb, xq = synthetic_dataset(d, 2000, 3000, 64, seed=2)
    idx, cent, _ = _build(res, 0, METRIC_L2, d, 0, nlist, xt, xb[:1])  # one stored row: 15 of 16 lists empty
    idx.nprobe = 4
    idx.set_scan_mode(lm_mode)
    D, I = idx.search(xq, k)
    sizes, codes, ids, _ = Oracle.build_ivf_lists(0, METRIC_L2, cent, xb[:1])
    Do, Io, _, _ = Oracle.ivf_search(0, METRIC_L2, cent, sizes, codes, ids, xq, 4, k, arith=idx.last_scan_arith())
    check_knn(D, I, Do, Io, exact=True, name="nearly empty index")
    idx.add(xb[1:])
    xqn = xq.copy()
    xqn[5, 3] = np.nan
    D, I = idx.search(xqn, k)
    sizes, codes, ids, _ = Oracle.build_ivf_lists(0, METRIC_L2, cent, xb)
    Do, Io, cD, cI = Oracle.ivf_search(0, METRIC_L2, cent, sizes, codes, ids, xqn, 4, k, arith=idx.last_scan_arith())
    keep = np.r_[0:5, 6:len(xq)]
    check_knn(D[keep], I[keep], Do[keep], Io[keep], exact=True, name="batch with a NaN query")
    assert (I[5] == -1).all()
    # preassigned: the caller's lists, some of them -1
    assign = cI.copy()
    assign[5] = 0
    assign[::3, 1] = -1
    cdis = cD.copy()
    cdis[5] = 0
    Dp, Ip = idx.search_preassigned(xq, k, assign, cdis)
    idx.set_scan_mode(idx.SCAN_QUERY_MAJOR)
    Dq, Iq = idx.search_preassigned(xq, k, assign, cdis)
    check_knn(Dp, Ip, Dq, Iq, rtol=1e-4, name="preassigned, list-major vs query-major")


def test_scan_mode_rule_and_refusals(res):
    d, nlist = 32, 64
    xt, xb, xq = synthetic_dataset(d, 3000, 8000, 2100, seed=4)
    idx, _, _ = _build(res, 0, METRIC_L2, d, 0, nlist, xt, xb)
    idx.nprobe = 8
    assert idx.scan_info()[0] == 0
    # the rule (GpuIndexIVF::list_major_rule: two cost models fitted to side-by-side timings, profiles/r04_i / r04_k / r04_y):
    # the time the query-major scan needs for the bytes it streams (queries x probes x rows per list x bytes per row) against
    # the fixed + per-query + per-touched-list cost of the list-major scan: a small index like this one (8000 rows of 128
    # bytes) stays query-major for any batch but a huge one.  A query
    # must also probe >= ~1.1 k granules of 16 rows, or the k-th best granule estimate bounds nothing (k = 100 of 1000 rows)
    assert not idx.list_major_rule(2100, 8, 10) and idx.list_major_rule(100000, 64, 10)
    assert idx.list_major_rule(100000, 64, 100) and not idx.list_major_rule(100000, 8, 100)
    D, I = idx.search(xq, 10)
    assert idx.scan_info()[1] == 1
    idx.set_scan_mode(idx.SCAN_LIST_MAJOR)
    D1, I1 = idx.search(xq, 10)
    assert idx.scan_info()[1] == 2 and np.array_equal(D, D1) and np.array_equal(I, I1)
    idx.set_scan_mode(idx.SCAN_AUTO)
    # a larger one: 64 lists of 3125 rows of 256 bytes; 2100 queries x 16 probes stream 27 GB
    d2 = 64
    xt2, xb2, xq2 = synthetic_dataset(d2, 3000, 200000, 2100, seed=5)
    big, _, _ = _build(res, 0, METRIC_L2, d2, 0, nlist, xt2, xb2)
    big.nprobe = 16
    assert big.list_major_rule(2100, 16, 10) and big.list_major_rule(300, 16, 10) and not big.list_major_rule(20, 16, 10)
    Db, Ib = big.search(xq2, 10)
    assert big.scan_info()[1] == 2 and big.last_scan_arith() == 0
    Ds, Is = big.search(xq2[:20], 10)
    assert big.scan_info()[1] == 1 and np.array_equal(Ds, Db[:20]) and np.array_equal(Is, Ib[:20])
    # IVFPQ: lists of >= 128 rows, and the code bytes the query-major scan would stream must outweigh the fixed cost
    pq, _, _ = _build(res, 1, METRIC_L2, d2, 16, nlist, xt2, xb2)
    pq.nprobe = 16
    assert not pq.list_major_rule(100, 4, 10) and pq.list_major_rule(2100, 16, 10) and pq.list_major_rule(2100, 64, 10)
    Dp, Ip = pq.search(xq2, 10, params=faiss_amd.SearchParametersIVF(nprobe=64))
    assert pq.scan_info()[1] == 2 and pq.last_scan_arith() == 0
    pq.set_scan_mode(pq.SCAN_QUERY_MAJOR)
    Dq, Iq = pq.search(xq2, 10, params=faiss_amd.SearchParametersIVF(nprobe=64))
    assert np.array_equal(Dp, Dq) and np.array_equal(Ip, Iq)
    with pytest.raises(faiss_amd.FaissAmdError):
        idx.set_scan_mode(4)
    idx.set_scan_mode(idx.SCAN_LIST_MAJOR_F32)
    with pytest.raises(faiss_amd.FaissAmdError, match="IDSelector"):
        idx.search(xq[:10], 5, params=faiss_amd.SearchParameters(sel=faiss_amd.IDSelectorRange(0, 100)))
    idx.set_scan_mode(idx.SCAN_LIST_MAJOR)  # (behind the filter the list-major scan serves selector searches)
    idx.search(xq[:10], 5, params=faiss_amd.SearchParameters(sel=faiss_amd.IDSelectorRange(0, 100)))
    assert idx.scan_info()[1] == 2
    # d > 128: IVFFlat behind the filter up to d = 512, the f32 scan not at all
    for dbig, modes_refused in ((136, (idx.SCAN_LIST_MAJOR_F32,)), (520, (idx.SCAN_LIST_MAJOR, idx.SCAN_LIST_MAJOR_F32))):
        big = faiss_amd.GpuIndexIVFFlat(res, dbig, 8, METRIC_L2)
        big.copy_centroids(np.random.RandomState(0).rand(8, dbig).astype("float32"))
        big.add(np.random.RandomState(1).rand(100, dbig).astype("float32"))
        assert not big.list_major_rule(10, 8, 2)
        for m in modes_refused:
            big.set_scan_mode(m)
            with pytest.raises(faiss_amd.FaissAmdError, match="not supported"):
                big.search(np.zeros((3, dbig), "float32"), 2)
    # IVFPQ: the codebook sweeps serve d <= 128; beyond that (round 5) the filter runs over the decoded residuals, up to d = 512;
    # the f32 scan still refuses
    pq_big = faiss_amd.GpuIndexIVFPQ(res, 256, 8, 32, 8, METRIC_L2)
    xs = np.random.RandomState(2).rand(3000, 256).astype("float32")
    pq_big.train(xs)
    pq_big.add(xs[:500])
    pq_big.set_scan_mode(pq_big.SCAN_QUERY_MAJOR)
    Dq, Iq = pq_big.search(xs[:3], 2)
    pq_big.set_scan_mode(pq_big.SCAN_LIST_MAJOR)
    Dl, Il = pq_big.search(xs[:3], 2)
    assert np.array_equal(Dq, Dl) and np.array_equal(Iq, Il)
    pq_big.set_scan_mode(pq_big.SCAN_LIST_MAJOR_F32)
    with pytest.raises(faiss_amd.FaissAmdError, match="not supported"):
        pq_big.search(xs[:3], 2)


@pytest.mark.parametrize("kind", [0, 1])
def test_list_major_after_incremental_adds_and_copied_lists(res, lm_mode, kind):
    """the per-row norms the scan needs follow the rows through list growth / relocation / compaction and bulk loads"""
    d, nlist, nb, nq, k, M = 64, 32, 30000, 300, 20, 16
    xt, xb, xq = synthetic_dataset(d, 4000, nb, nq, seed=9)
    idx, cent, pq = _build(res, kind, METRIC_L2, d, M, nlist, xt, xb[:100])
    for a, b in ((100, 150), (150, 4000), (4000, 4100), (4100, 30000)):
        idx.add(xb[a:b])
    idx.nprobe = 7
    idx.set_scan_mode(lm_mode)
    D, I = idx.search(xq, k)
    sizes, codes, ids, _ = Oracle.build_ivf_lists(kind, METRIC_L2, cent, xb, pq=pq)
    Do, Io, _, _ = Oracle.ivf_search(kind, METRIC_L2, cent, sizes, codes, ids, xq, 7, k, M=M if kind else 0, pq=pq, arith=idx.last_scan_arith())
    check_knn(D, I, Do, Io, exact=True, name="after incremental adds")
    # bulk load of the same lists into a fresh index (copyFrom path)
    if kind == 0:
        idx2 = faiss_amd.GpuIndexIVFFlat(res, d, nlist, METRIC_L2)
    else:
        idx2 = faiss_amd.GpuIndexIVFPQ(res, d, nlist, M, 8, METRIC_L2)
        idx2.copy_pq_centroids(pq)
    idx2.copy_centroids(cent)
    idx2.copy_lists(sizes, codes, ids)
    idx2.nprobe = 7
    idx2.set_scan_mode(lm_mode)
    D2, I2 = idx2.search(xq, k)
    assert np.array_equal(D, D2) and np.array_equal(I, I2)
    freed = idx.reclaimMemory()  # compaction moves every list
    D3, I3 = idx.search(xq, k)
    assert freed >= 0 and np.array_equal(D, D3) and np.array_equal(I, I3)


@pytest.mark.gpu
@pytest.mark.parametrize("k", [1, 50, 256, 300])
def test_list_major_many_equal_distances(res, lm_mode, k):
    """Integer-valued vectors: thousands of rows at exactly the same distance from a query.  The bound and the final
    selection must pick the k smallest (distance, scan position) keys and order them by (distance, label) -- the
    wavefront-per-query kernel resolves the ties at the k-th distance by a second bisection on the position (k <= 256), the
    radix kernel serves k = 300; both against the oracle bit for bit."""
    d, nlist, nb, nq, nprobe = 16, 16, 30000, 500, 5
    rs = np.random.RandomState(k)
    xb = rs.randint(0, 3, size=(nb, d)).astype("float32")
    xq = rs.randint(0, 3, size=(nq, d)).astype("float32")
    cent = rs.rand(nlist, d).astype("float32") * 2
    idx = faiss_amd.GpuIndexIVFFlat(res, d, nlist, METRIC_L2)
    idx.copy_centroids(cent)
    idx.add(xb)
    idx.nprobe = nprobe
    idx.set_scan_mode(lm_mode)
    D, I = idx.search(xq, k)
    assert idx.scan_info()[1] == 2
    sizes, codes, ids, _ = Oracle.build_ivf_lists(0, METRIC_L2, cent, xb)
    Do, Io, _, _ = Oracle.ivf_search(0, METRIC_L2, cent, sizes, codes, ids, xq, nprobe, k, arith=idx.last_scan_arith())
    check_knn(D, I, Do, Io, exact=True, name="tie-heavy list-major")
    assert len(np.unique(D[0])) < max(2, k // 2) or k == 1  # the data really ties


@pytest.mark.parametrize("kind,d", [(0, 64), (1, 64), (0, 256)])
def test_results_do_not_depend_on_the_batch_size(res, kind, d):
    """the automatic rule sends large batches through the list-major scan.  Behind the f16 filter that
    scan returns the bits of the query-major scan, so a query's distances and labels are the same whatever batch it
    arrives in (and therefore under IndexShards / IndexReplicas query splits and the paged host path)."""
    nlist, M, k = 64, 32, 40
    xt, xb, xq = synthetic_dataset(d, 4000, 60000, 2048, seed=31)
    idx, _, _ = _build(res, kind, METRIC_L2, d, M, nlist, xt, xb)
    idx.nprobe = 16
    if kind == 1:
        idx.set_scan_mode(idx.SCAN_LIST_MAJOR)  # (the IVFPQ rule asks for more code bytes; force the scan)
    else:
        assert idx.list_major_rule(2048, 16, k) and not idx.list_major_rule(30, 16, k)
    D, I = idx.search(xq, k)
    assert idx.scan_info()[1] == 2 and idx.last_scan_arith() == 0
    D2, I2 = idx.search(xq[:2047], k)
    assert idx.scan_info()[1] == 2 and np.array_equal(D2, D[:2047]) and np.array_equal(I2, I[:2047])
    idx.set_scan_mode(idx.SCAN_AUTO)
    D1, I1 = idx.search(xq[:30], k)
    assert idx.scan_info()[1] == 1
    assert np.array_equal(D1, D[:30]) and np.array_equal(I1, I[:30])


def test_list_major_near_duplicates_and_self_search(res, lm_mode):
    """rows that are (near-)duplicates of the queries.  The f32 list-major scan computes L2 by the norm
    expansion |q|^2 + |y|^2 - 2 <q, y> (clamped at 0): near distance 0 its relative error is unbounded (cancellation),
    its absolute error stays at the rounding of the norms.  Behind the filter the list-major scan re-derives the
    survivors with the direct sum of (q - y)^2: a stored copy of the query comes back with distance exactly 0 and the
    near-duplicates keep full relative precision -- the bits of the query-major scan."""
    d, nlist, k = 64, 32, 10
    xt, xb, xq = synthetic_dataset(d, 4000, 30000, 2100, seed=17)
    rs = np.random.RandomState(5)
    xb = xb.copy()
    xb[:2100] = xq                                                     # exact copies
    xb[2100:4200] = xq + (rs.rand(2100, d).astype("float32") - 0.5) * 1e-3   # near-duplicates
    idx, cent, _ = _build(res, 0, METRIC_L2, d, 0, nlist, xt, xb)
    idx.nprobe = 8
    idx.set_scan_mode(idx.SCAN_QUERY_MAJOR)
    D0, I0 = idx.search(xq, k)
    idx.set_scan_mode(lm_mode)
    D, I = idx.search(xq, k)
    assert idx.scan_info()[1] == 2
    if idx.last_scan_arith() == 0:
        assert np.array_equal(D, D0) and np.array_equal(I, I0)
        assert (D[:, 0] == 0).all() and (I[:, 0] == np.arange(2100)).all()
        exact = ((xq.astype(np.float64) - xb[I[:, 1]].astype(np.float64)) ** 2).sum(-1)
        assert np.allclose(D[:, 1], exact, rtol=1e-5, atol=0)         # ~1e-5 distances, relative precision kept
    else:
        # the norm expansion: absolute error at the scale of the rounding of |q|^2 + |y|^2, whatever the distance
        scale = (xq.astype(np.float64) ** 2).sum(-1)
        assert (np.abs(D[:, 0]) <= 1e-5 * scale).all()
        exact = ((xq.astype(np.float64)[:, None, :] - xb[I].astype(np.float64)) ** 2).sum(-1)
        assert (np.abs(D - exact) <= 1e-5 * scale[:, None] + 1e-4 * exact).all()


@pytest.mark.parametrize("kind,metric,d,M,scale", [
    (0, METRIC_L2, 128, 0, 1.0), (0, METRIC_INNER_PRODUCT, 128, 0, 1.0), (0, METRIC_L2, 40, 0, 1.0),
    (0, METRIC_L2, 128, 0, 200.0),           # large values: the band scales with |q| |y|
    (0, METRIC_L2, 64, 0, 1e-3),             # small values (fp16 denormals in play)
    (0, METRIC_L2, 256, 0, 1.0), (0, METRIC_INNER_PRODUCT, 300, 0, 1.0), (0, METRIC_L2, 512, 0, 1.0),  # d > 128
    (1, METRIC_L2, 128, 64, 1.0), (1, METRIC_INNER_PRODUCT, 128, 64, 1.0), (1, METRIC_L2, 64, 16, 1.0),
    (1, METRIC_L2, 96, 12, 1.0), (1, METRIC_L2, 32, 32, 1.0), (1, METRIC_L2, 128, 64, 50.0),
    # few sub-quantizers of many coordinates (dsub = 32) with the data far from the origin -- the fp32 chains of the
    # exact path's per-row term |r^|^2 + 2 <c, r^> then carry an error of ~ d 2^-24 |c| |r^| that the M-proportional term of
    # the band does not cover (scale < 0 marks the shifted case: every coordinate + 20)
    (1, METRIC_L2, 128, 4, -1.0), (1, METRIC_INNER_PRODUCT, 128, 4, -1.0), (1, METRIC_L2, 64, 4, -1.0),
])
def test_list_filter_error_bound_holds(res, kind, metric, d, M, scale):
    """The superset argument of the f16 filter (ivf_lm_filter.hip) rests on |estimate - exact| <= E_q for EVERY row a query
    probes (kernels.h ivf_filter_err_bound + the IVFPQ table-grid term).  Checked directly: the estimates of every probed
    row (test hook, sweep mode 3) against the exact distances of the query-major arithmetic (the oracle, k = all rows),
    with the band the bound kernel grants the query.  The measured worst |estimate - exact| / E_q is printed: how much
    of the band real data uses."""
    nlist, nb, nq, nprobe = 16, 12000, 96, 4
    xt, xb, xq = synthetic_dataset(d, 3000, nb, nq, seed=d + M)
    if scale < 0:
        xt, xb, xq = xt + np.float32(20.0), xb + np.float32(20.0), xq + np.float32(20.0)
        scale = 1.0
    xt, xb, xq = xt * np.float32(scale), xb * np.float32(scale), xq * np.float32(scale)
    idx, cent, pq = _build(res, kind, metric, d, M, nlist, xt, xb)
    if pq is not None and scale != 1.0:
        pq = (pq * np.float32(scale)).astype(np.float32)
        idx.copy_pq_centroids(pq)
        idx.reset()
        idx.add(xb)
    idx.nprobe = nprobe
    sizes, codes, ids, _ = Oracle.build_ivf_lists(kind, metric, cent, xb, pq=pq)
    Dq, Iq = idx.quantizer_search(xq, nprobe)
    rows = np.array([int(sum(sizes[l] for l in Iq[q] if l >= 0)) for q in range(nq)])
    stride = int(nprobe * sizes.max())
    est, band = idx.filter_dump(xq, 10, stride)
    assert np.isfinite(band).all() and (band > 0).all()
    # exact distances of all probed rows: the oracle with k = the rows the query probes, mapped back to scan positions
    kall = int(rows.max())
    Do, Io, _, _ = Oracle.ivf_search(kind, metric, cent, sizes, codes, ids, xq, nprobe, kall, M=M, pq=pq)
    starts = np.concatenate([[0], np.cumsum(sizes)])[:-1]
    worst = 0.0
    for q in range(nq):
        pos_ids = np.concatenate([ids[int(starts[l]):int(starts[l]) + int(sizes[l])] for l in Iq[q] if l >= 0])
        assert len(pos_ids) == rows[q] and not np.isnan(est[q, :rows[q]]).any() and np.isnan(est[q, rows[q]:]).all()
        exact = np.empty(rows[q], dtype=np.float64)
        order = {int(i): r for r, i in enumerate(Io[q, :rows[q]])}
        for p_, i in enumerate(pos_ids):
            exact[p_] = Do[q, order[int(i)]]
        err = np.abs(est[q, :rows[q]].astype(np.float64) - exact)
        assert (err <= band[q]).all(), (q, float(err.max()), float(band[q]))
        worst = max(worst, float(err.max() / band[q]))
    print("kind %d metric %d d %d scale %g: worst |estimate - exact| / band = %.3f" % (kind, metric, d, scale, worst))
    assert worst > 1e-4  # (the estimates are estimates: a zero error would mean the test compares a thing with itself)
