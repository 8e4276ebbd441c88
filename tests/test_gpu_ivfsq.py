"""GPU parity tests of GpuIndexIVFScalarQuantizer (pytest -m gpu), through the C ABI:
codes byte-identical to faiss::ScalarQuantizer::compute_codes (oracle restatement, pinned on the real reference by
tests/test_oracle_cpu.py::test_ivfsq_* and the committed golden fixture), search results BIT-EXACT against the oracle's
restatement of the IVFSQ scanners in the kernel's summation order, and within 1e-4 relative of the live reference
(faiss/gpu/test/TestGpuIndexIVFScalarQuantizer.cpp runs the same matrix: every supported qtype, L2 / IP, copyFrom /
copyTo)."""
import numpy as np
import pytest

import faiss_amd
from compare import check_knn
from faiss_amd import ScalarQuantizer as SQ
from oracle.pyoracle import METRIC_INNER_PRODUCT, METRIC_L2, Oracle, Ref, synthetic_dataset

pytestmark = pytest.mark.gpu

QTYPES = [SQ.QT_8bit, SQ.QT_4bit, SQ.QT_8bit_uniform, SQ.QT_4bit_uniform, SQ.QT_fp16, SQ.QT_8bit_direct, SQ.QT_6bit]
QNAMES = {SQ.QT_8bit: "8bit", SQ.QT_4bit: "4bit", SQ.QT_8bit_uniform: "8bit_uniform", SQ.QT_4bit_uniform: "4bit_uniform",
          SQ.QT_fp16: "fp16", SQ.QT_8bit_direct: "8bit_direct", SQ.QT_6bit: "6bit"}


def _data(qtype, d, nt, nb, nq, seed):
    xt, xb, xq = synthetic_dataset(d, nt, nb, nq, seed=seed)
    if qtype == SQ.QT_8bit_direct:
        # "fast indexing of uint8s": byte-valued data
        sc = 255.0 / max(xt.max(), xb.max(), xq.max())
        xt, xb, xq = (np.floor(np.abs(v) * sc).astype(np.float32) for v in (xt, xb, xq))
    return xt, xb, xq


def _gpu_lists(idx):
    sizes = np.array([idx.get_list_size(l) for l in range(idx.nlist)], dtype=np.uint32)
    codes = np.concatenate([idx.get_list_codes(l) for l in range(idx.nlist)], axis=0)
    ids = np.concatenate([idx.get_list_ids(l) for l in range(idx.nlist)])
    return sizes, codes, ids


@pytest.mark.parametrize("by_residual", [True, False])
@pytest.mark.parametrize("metric", [METRIC_L2, METRIC_INNER_PRODUCT])
@pytest.mark.parametrize("qtype", QTYPES, ids=[QNAMES[q] for q in QTYPES])
def test_ivfsq_codes_and_search_match_oracle(res, qtype, metric, by_residual):
    d, nlist, nb, nq, nprobe, k = 40, 32, 20000, 700, 6, 50  # d = 40: partly filled chunks, padded rows, partial blocks
    xt, xb, xq = _data(qtype, d, 6000, nb, nq, seed=31 + qtype)
    idx = faiss_amd.GpuIndexIVFScalarQuantizer(res, d, nlist, qtype, metric, by_residual)
    assert not idx.is_trained
    idx.train(xt)
    assert idx.is_trained and idx.qtype == qtype and idx.by_residual == by_residual
    assert idx.code_size == Oracle.sq_code_size(qtype, d)
    idx.add(xb)
    idx.nprobe = nprobe
    cent = idx.get_centroids()
    vmin, vdiff = Oracle.sq_unpack(qtype, d, idx.get_trained())
    # ---- codes: ScalarQuantizer::compute_codes on the (residual) vectors, byte for byte
    sizes, codes, ids = _gpu_lists(idx)
    lab = Oracle.ivf_assign(metric, cent, xb)
    want = Oracle.sq_encode(qtype, xb, vmin, vdiff, lab if by_residual else None, cent if by_residual else None)
    got = np.empty_like(want)
    got[ids] = codes
    assert np.array_equal(got, want)
    assert np.array_equal(np.bincount(lab, minlength=nlist).astype(np.uint32), sizes)
    # ---- search: bit-exact against the oracle's restatement of the scanner
    D, I = idx.search(xq, k)
    sel = np.r_[0:60]
    Do, Io = Oracle.ivfsq_search(qtype, by_residual, metric, cent, sizes, codes, ids, vmin, vdiff, xq[sel], nprobe, k)
    check_knn(D[sel], I[sel], Do, Io, exact=True, name="ivfsq vs oracle")
    # small batch: the probes of a query are split over several workgroups and merged by the select kernel
    D2, I2 = idx.search(xq[:7], k)
    assert np.array_equal(D2, D[:7]) and np.array_equal(I2, I[:7])


@pytest.mark.parametrize("qtype,metric,by_residual,d,nlist,nb,nq,nprobe,k", [
    (SQ.QT_8bit, METRIC_L2, True, 128, 64, 40000, 1500, 8, 100),        # bench shape: one chunk group per block
    (SQ.QT_8bit, METRIC_INNER_PRODUCT, True, 128, 64, 40000, 1100, 8, 10),
    (SQ.QT_8bit, METRIC_L2, True, 200, 16, 8000, 1100, 4, 20),           # two chunk groups
    (SQ.QT_4bit, METRIC_L2, True, 300, 16, 6000, 40, 16, 600),           # three chunk groups, probes split, big k
    (SQ.QT_fp16, METRIC_L2, False, 600, 8, 3000, 1030, 3, 5),            # ten chunk groups of fp16
    (SQ.QT_6bit, METRIC_INNER_PRODUCT, False, 1024, 8, 2000, 20, 8, 7),  # d at the limit
    (SQ.QT_8bit_uniform, METRIC_L2, True, 16, 8, 5000, 1200, 5, 2048),   # k at the limit, one chunk in all
    (SQ.QT_8bit, METRIC_L2, True, 128, 512, 60000, 1100, 300, 10),       # nprobe x d beyond the table rows: groups
])
def test_ivfsq_shapes_match_oracle(res, qtype, metric, by_residual, d, nlist, nb, nq, nprobe, k):
    xt, xb, xq = _data(qtype, d, max(4000, 40 * nlist), nb, nq, seed=nb + k)
    idx = faiss_amd.GpuIndexIVFScalarQuantizer(res, d, nlist, qtype, metric, by_residual)
    idx.train(xt)
    idx.add(xb)
    idx.nprobe = nprobe
    D, I = idx.search(xq, k)
    cent = idx.get_centroids()
    vmin, vdiff = Oracle.sq_unpack(qtype, d, idx.get_trained())
    sizes, codes, ids = _gpu_lists(idx)
    sel = np.r_[0:min(nq, 24)]
    Do, Io = Oracle.ivfsq_search(qtype, by_residual, metric, cent, sizes, codes, ids, vmin, vdiff, xq[sel], nprobe, k)
    check_knn(D[sel], I[sel], Do, Io, exact=True, name="ivfsq vs oracle")


def test_ivfsq_trained_range_is_minmax_of_the_residuals(res):
    """ScalarQuantizer::train with RS_minmax (training.cpp:209-232, 333-365): per-dimension (or global) minimum and
    range of the training residuals, widened by rangestat_arg."""
    d, nlist = 24, 16
    xt, _, _ = synthetic_dataset(d, 5000, 0, 0, seed=4)
    for qtype, arg in [(SQ.QT_8bit, 0.0), (SQ.QT_4bit, 0.1), (SQ.QT_8bit_uniform, 0.0), (SQ.QT_6bit, 0.25)]:
        idx = faiss_amd.GpuIndexIVFScalarQuantizer(res, d, nlist, qtype, METRIC_L2, True)
        idx.set_rangestat(SQ.RS_minmax, arg)
        idx.train(xt)
        cent = idx.get_centroids()
        r = xt - cent[Oracle.ivf_assign(METRIC_L2, cent, xt)]
        t = idx.get_trained()
        if qtype == SQ.QT_8bit_uniform:
            vmin, vmax = np.float32(r.min()), np.float32(r.max())
            vexp = np.float32(vmax - vmin) * np.float32(arg)
            assert t.shape == (2,) and t[0] == vmin - vexp and t[1] == (vmax + vexp) - (vmin - vexp)
        else:
            vmin, vmax = r.min(axis=0), r.max(axis=0)
            vexp = (vmax - vmin) * np.float32(arg)
            assert np.array_equal(t[:d], vmin - vexp) and np.array_equal(t[d:], (vmax + vexp) - (vmin - vexp))
    idx = faiss_amd.GpuIndexIVFScalarQuantizer(res, d, nlist, SQ.QT_8bit, METRIC_L2, True)
    idx.set_rangestat(SQ.RS_meanstd, 2.0)  # (served since round 6: tests/test_gpu_round6.py::test_ivfsq_trains_every_range_statistic)
    idx.train(xt)
    assert idx.is_trained
    idx.set_rangestat(7, 0.0)  # not a ScalarQuantizer::RangeStat
    idx2 = faiss_amd.GpuIndexIVFScalarQuantizer(res, d, nlist, SQ.QT_8bit, METRIC_L2, True)
    idx2.set_rangestat(7, 0.0)
    with pytest.raises(RuntimeError):
        idx2.train(xt)
    # the types without a range are trained as soon as the coarse quantizer is
    f = faiss_amd.GpuIndexIVFScalarQuantizer(res, d, nlist, SQ.QT_fp16, METRIC_L2, True)
    f.train(xt)
    assert f.is_trained and f.get_trained().size == 0
    with pytest.raises(RuntimeError):
        faiss_amd.GpuIndexIVFScalarQuantizer(res, d, nlist, 7, METRIC_L2, True)  # QT_bf16: not a GPU type


@pytest.mark.skipif(not Ref.available(), reason="oracle/_ref not shipped")
@pytest.mark.parametrize("by_residual", [True, False])
@pytest.mark.parametrize("metric", [METRIC_L2, METRIC_INNER_PRODUCT])
@pytest.mark.parametrize("qtype", QTYPES, ids=[QNAMES[q] for q in QTYPES])
def test_ivfsq_vs_live_reference(res, qtype, metric, by_residual):
    """The same quantizers in faiss::IndexIVFScalarQuantizer (CPU): identical lists (codes and ids, byte for byte),
    distances within 1e-4 relative, labels equal outside near-tie groups; then the other direction: the reference's
    lists loaded into the backend (copyFrom) give the same results as its own add()."""
    d, nlist, nb, nq, nprobe, k = 64, 32, 15000, 200, 8, 20
    if qtype == SQ.QT_8bit_direct:
        # for d % 16 == 0 the CPU reference scans QT_8bit_direct with DistanceComputerByte, which truncates the QUERY to
        # bytes as well (impl/scalar_quantizer/distance_computers.h; residual queries are not byte valued); the
        # reference GPU codec keeps the query in fp32 (gpu/impl/GpuScalarQuantizer.cuh:503-570), like this backend
        d = 72
    xt, xb, xq = _data(qtype, d, 5000, nb, nq, seed=77 + qtype)
    idx = faiss_amd.GpuIndexIVFScalarQuantizer(res, d, nlist, qtype, metric, by_residual)
    idx.train(xt)
    idx.add(xb)
    idx.nprobe = nprobe
    D, I = idx.search(xq, k)
    ref = Ref.ivfsq(d, nlist, qtype, metric, by_residual)
    ref.set_sq_trained(idx.get_centroids(), idx.get_trained())
    ref.add(xb)
    ref.set_nprobe(nprobe)
    rs, rc, ri = ref.lists()
    gs, gc, gi = _gpu_lists(idx)
    # a vector at (near-)equal distance from two centroids may land in either list (the reference's coarse quantizer
    # sums in BLAS order); everywhere else the lists agree entry for entry
    same_lists = np.array_equal(rs, gs) and np.array_equal(ri, gi)
    gcode, rcode = np.empty_like(gc), np.empty_like(rc)
    gcode[gi], rcode[ri] = gc, rc
    glist, rlist = np.empty(nb, np.int64), np.empty(nb, np.int64)
    glist[gi], rlist[ri] = np.repeat(np.arange(nlist), gs), np.repeat(np.arange(nlist), rs)
    agree = glist == rlist
    assert agree.mean() > 0.99  # (byte-valued data under the inner product: many near-equal coarse scores)
    assert np.array_equal(gcode[agree], rcode[agree])
    if same_lists:
        assert np.array_equal(rc, gc)
    Dr, Ir = ref.search(xq, k)
    other = faiss_amd.GpuIndexIVFScalarQuantizer(res, d, nlist, qtype, metric, by_residual)
    other.copy_centroids(ref.centroids())
    if ref.sq_trained().size:
        other.copy_trained(ref.sq_trained())
    assert other.is_trained
    other.copy_lists(rs, rc, ri)
    other.nprobe = nprobe
    # the reference's own coarse assignment (an IndexFlat over the same centroids), so that both sides scan the same
    # lists even where two centroids are nearly equally close to a query (the inner product over byte-valued data
    # produces such queries: one in 200 had a different 8th probe)
    cq = Ref.index_factory(d, "Flat", metric)
    cq.add(ref.centroids())
    Dq, Iq = cq.search(xq, nprobe)
    D2, I2 = other.search_preassigned(xq, k, Iq, Dq)
    check_knn(D2, I2, Dr, Ir, rtol=1e-4, tie_rtol=1e-4, name="ivfsq vs reference")
    D3, I3 = other.search(xq, k)  # with its own coarse quantizer: the same wherever the probes agree
    _, Ig = other.quantizer_search(xq, nprobe)
    same_probes = (np.sort(Ig, axis=1) == np.sort(Iq, axis=1)).all(axis=1)
    assert same_probes.mean() > 0.5
    check_knn(D3[same_probes], I3[same_probes], Dr[same_probes], Ir[same_probes], rtol=1e-4, tie_rtol=1e-4,
              name="ivfsq vs reference, own probes")
    if same_lists:
        assert np.array_equal(D3, D) and np.array_equal(I3, I)


def test_ivfsq_incremental_adds_nan_rows_reset(res):
    d, nlist = 32, 16
    xt, xb, xq = synthetic_dataset(d, 4000, 9000, 64, seed=12)
    a = faiss_amd.GpuIndexIVFScalarQuantizer(res, d, nlist, SQ.QT_8bit, METRIC_L2, True)
    a.train(xt)
    b = faiss_amd.GpuIndexIVFScalarQuantizer(res, d, nlist, SQ.QT_8bit, METRIC_L2, True)
    b.copy_centroids(a.get_centroids())
    b.copy_trained(a.get_trained())
    a.add(xb)
    for i0 in range(0, 9000, 777):
        b.add(xb[i0:i0 + 777])
    a.nprobe = b.nprobe = 5
    Da, Ia = a.search(xq, 10)
    Db, Ib = b.search(xq, 10)
    assert np.array_equal(Da, Db) and np.array_equal(Ia, Ib)
    # NaN rows are counted by ntotal but not stored (faiss/gpu/GpuIndexIVF.cu:293-298); NaN queries find nothing
    bad = xb[:5].copy()
    bad[2, 3] = np.nan
    a.add(bad)
    assert a.ntotal == 9005 and a.stored_vectors == 9004
    q = xq[:3].copy()
    q[1, 0] = np.nan
    D, I = a.search(q, 4)
    assert (I[1] == -1).all() and (I[0] >= 0).all()
    a.reset()
    assert a.ntotal == 0 and a.is_trained
    D, I = a.search(xq[:2], 3)
    assert (I == -1).all()
    with pytest.raises(RuntimeError):
        faiss_amd.GpuIndexIVFScalarQuantizer(res, d, nlist, SQ.QT_8bit, METRIC_L2, True).add(xb[:10])  # untrained


# ---------------------------------------------------------------------- list-major scan (ivf_listmajor.hip, kind 2)
LM_QTYPES = QTYPES


def _lm_check(idx, qtype, metric, by_residual, xq, nprobe, k, nsel=48):
    """query-major vs the two list-major scans vs the oracle:
      * SCAN_LIST_MAJOR (round 5: behind the f16 filter, ivf_lm_filter.hip over an fp16 copy of the centred codes): the very
        bits of the query-major scan (arith 0), for every query;
      * SCAN_LIST_MAJOR_F32 (round 3's f32 MFMA scan, d <= 128): bit-exact against its own restatement (arith 1), within the
        north-star tolerance of the query-major scan."""
    d = xq.shape[1]
    idx.nprobe = nprobe
    idx.set_scan_mode(idx.SCAN_QUERY_MAJOR)
    D0, I0 = idx.search(xq, k)
    assert idx.scan_info()[1] == 1
    idx.set_scan_mode(idx.SCAN_LIST_MAJOR)
    D, I = idx.search(xq, k)
    assert idx.scan_info()[1] == 2 and idx.last_scan_arith() == 0
    assert np.array_equal(I, I0) and np.array_equal(D, D0), "filter path differs from the query-major scan"
    D2, I2 = idx.search(xq, k)
    assert np.array_equal(D, D2) and np.array_equal(I, I2)
    if d <= 128:
        idx.set_scan_mode(idx.SCAN_LIST_MAJOR_F32)
        D, I = idx.search(xq, k)
        assert idx.scan_info()[1] == 2 and idx.last_scan_arith() == 1
        # (two roundings of the same sums: where offset and code terms cancel -- inner products near zero at deep ranks --
        # neighbours a few 1e-6 of the row's scale apart may swap; each scan is bit-exact against its own restatement below)
        check_knn(D, I, D0, I0, rtol=1e-4, tie_rtol=2e-3, name="ivfsq list-major vs query-major")
        cent = idx.get_centroids()
        vmin, vdiff = Oracle.sq_unpack(qtype, d, idx.get_trained())
        sizes, codes, ids = _gpu_lists(idx)
        sel = np.r_[0:min(len(xq), nsel)]
        Do, Io = Oracle.ivfsq_search(qtype, by_residual, metric, cent, sizes, codes, ids, vmin, vdiff, xq[sel], nprobe, k, arith=1)
        check_knn(D[sel], I[sel], Do, Io, exact=True, name="ivfsq list-major vs oracle")
        D2, I2 = idx.search(xq, k)  # run to run: the order in which wavefronts append candidates never shows
        assert np.array_equal(D, D2) and np.array_equal(I, I2)
    idx.set_scan_mode(idx.SCAN_AUTO)
    return D, I


@pytest.mark.parametrize("by_residual", [True, False])
@pytest.mark.parametrize("metric", [METRIC_L2, METRIC_INNER_PRODUCT])
@pytest.mark.parametrize("qtype", LM_QTYPES, ids=[QNAMES[q] for q in LM_QTYPES])
def test_ivfsq_list_major_matches_oracle_and_query_major(res, qtype, metric, by_residual):
    """Every code type x metric x residual flag, d = 40 (dpad 40, rows of three 16-component
    chunks, the last one half filled; 32-row blocks in both halves of the 64-row code blocks, ragged list ends)."""
    d, nlist, nb, nq, nprobe, k = 40, 32, 20000, 700, 6, 50
    xt, xb, xq = _data(qtype, d, 6000, nb, nq, seed=31 + qtype)
    idx = faiss_amd.GpuIndexIVFScalarQuantizer(res, d, nlist, qtype, metric, by_residual)
    idx.train(xt)
    idx.add(xb)
    _lm_check(idx, qtype, metric, by_residual, xq, nprobe, k)


@pytest.mark.parametrize("qtype,metric,by_residual,d,nlist,nb,nq,nprobe,k", [
    (SQ.QT_8bit, METRIC_L2, True, 128, 64, 40000, 1500, 8, 100),            # bench shape: the fully unrolled kernels
    (SQ.QT_8bit, METRIC_INNER_PRODUCT, True, 128, 64, 40000, 1100, 8, 10),
    (SQ.QT_fp16, METRIC_L2, True, 128, 64, 30000, 900, 8, 100),
    (SQ.QT_4bit, METRIC_L2, False, 128, 64, 30000, 900, 8, 100),            # 16 levels: many equal distances (ties by position)
    (SQ.QT_8bit, METRIC_L2, True, 72, 8, 20000, 1100, 2, 1000),             # dpad 72 < dsq 80; lists of ~2500 rows: row chunks, big k
    (SQ.QT_8bit_uniform, METRIC_L2, True, 64, 8, 90000, 600, 3, 50),        # lists of ~11 000 rows: row chunks of 2816 rows
    (SQ.QT_fp16, METRIC_INNER_PRODUCT, False, 32, 128, 20000, 600, 100, 10),  # more than 64 probes
    (SQ.QT_4bit_uniform, METRIC_INNER_PRODUCT, True, 64, 32, 5000, 130, 5, 2048),  # k above the rows many queries see
    (SQ.QT_8bit_direct, METRIC_L2, False, 16, 8, 5000, 1200, 5, 2048),      # one chunk per row, k at the limit
    (SQ.QT_8bit, METRIC_L2, True, 8, 16, 9000, 300, 16, 7),                 # d = 8: half a chunk, every list probed
    (SQ.QT_6bit, METRIC_L2, True, 128, 64, 30000, 900, 8, 100),             # 6-bit fields cut out of dword pairs, unrolled kernels
    (SQ.QT_6bit, METRIC_INNER_PRODUCT, False, 72, 8, 20000, 700, 2, 300),   # ... dpad 72, row chunks
])
def test_ivfsq_list_major_shapes(res, qtype, metric, by_residual, d, nlist, nb, nq, nprobe, k):
    xt, xb, xq = _data(qtype, d, max(4000, 40 * nlist), nb, nq, seed=nb + k)
    idx = faiss_amd.GpuIndexIVFScalarQuantizer(res, d, nlist, qtype, metric, by_residual)
    idx.train(xt)
    idx.add(xb)
    _lm_check(idx, qtype, metric, by_residual, xq, nprobe, k, nsel=24)


def test_ivfsq_list_major_rule_and_refusals(res):
    d, nlist = 32, 16
    xt, xb, xq = synthetic_dataset(d, 4000, 200000, 2500, seed=3)
    idx = faiss_amd.GpuIndexIVFScalarQuantizer(res, d, nlist, SQ.QT_8bit, METRIC_L2, True)
    idx.train(xt)
    idx.add(xb)
    idx.nprobe = 4
    # large batches take the list-major scan on their own (the cost model of the filter path since round 5: the bytes a
    # query-major scan streams against the sweeps' fixed cost), small ones the query-major one
    assert idx.list_major_rule(2500) and not idx.list_major_rule(20)
    D, I = idx.search(xq, 10)
    assert idx.scan_info()[1] == 2
    D1, I1 = idx.search(xq[:100], 10)
    assert idx.scan_info()[1] == 1
    # (behind the f16 filter the two scans return the same bits)
    assert idx.last_scan_arith() == 0 and np.array_equal(D[:100], D1) and np.array_equal(I[:100], I1)
    # d > 128: the filter path serves up to d = 512 (same bits as the query-major scan), the f32 scan refuses
    xt2, xb2, xq2 = synthetic_dataset(136, 3000, 4000, 2100, seed=5)
    wide = faiss_amd.GpuIndexIVFScalarQuantizer(res, 136, nlist, SQ.QT_8bit, METRIC_L2, True)
    wide.train(xt2)
    wide.add(xb2)
    wide.nprobe = 4
    wide.set_scan_mode(wide.SCAN_QUERY_MAJOR)
    Dq, Iq = wide.search(xq2, 10)
    wide.set_scan_mode(wide.SCAN_LIST_MAJOR)
    Dl, Il = wide.search(xq2, 10)
    assert wide.scan_info()[1] == 2 and np.array_equal(Iq, Il) and np.array_equal(Dq, Dl)
    wide.set_scan_mode(wide.SCAN_LIST_MAJOR_F32)
    with pytest.raises(RuntimeError):
        wide.search(xq2, 10)
    huge = faiss_amd.GpuIndexIVFScalarQuantizer(res, 520, nlist, SQ.QT_8bit, METRIC_L2, True)
    xt3, xb3, xq3 = synthetic_dataset(520, 2000, 3000, 2100, seed=6)
    huge.train(xt3)
    huge.add(xb3)
    assert not huge.list_major_rule(2500)
    huge.set_scan_mode(huge.SCAN_LIST_MAJOR)
    with pytest.raises(RuntimeError):
        huge.search(xq3, 10)


def test_ivfsq_list_major_row_norms_follow_the_rows(res):
    """The per-row term |s o code|^2 (arena_rn) through list growth and relocation (many small adds), a bulk load
    (copy_lists) and a later copy_trained: the list-major results are those of one big add."""
    d, nlist, k = 32, 16, 10
    xt, xb, xq = synthetic_dataset(d, 4000, 9000, 400, seed=12)
    a = faiss_amd.GpuIndexIVFScalarQuantizer(res, d, nlist, SQ.QT_8bit, METRIC_L2, True)
    a.train(xt)
    b = faiss_amd.GpuIndexIVFScalarQuantizer(res, d, nlist, SQ.QT_8bit, METRIC_L2, True)
    b.copy_centroids(a.get_centroids())
    b.copy_trained(a.get_trained())
    a.add(xb)
    for i0 in range(0, 9000, 777):
        b.add(xb[i0:i0 + 777])
    for idx in (a, b):
        idx.nprobe = 5
        idx.set_scan_mode(idx.SCAN_LIST_MAJOR_F32)
    Da, Ia = a.search(xq, k)
    Db, Ib = b.search(xq, k)
    assert np.array_equal(Da, Db) and np.array_equal(Ia, Ib)
    c = faiss_amd.GpuIndexIVFScalarQuantizer(res, d, nlist, SQ.QT_8bit, METRIC_L2, True)
    c.copy_centroids(a.get_centroids())
    c.copy_trained(a.get_trained())
    c.copy_lists(*_gpu_lists(a))
    c.copy_trained(a.get_trained())  # the same ranges again, now under stored rows: the norms are recomputed
    c.nprobe = 5
    c.set_scan_mode(c.SCAN_LIST_MAJOR_F32)
    Dc, Ic = c.search(xq, k)
    assert np.array_equal(Da, Dc) and np.array_equal(Ia, Ic)
