"""GPU parity at the BASELINE.json shapes (pytest -m gpu): every search goes through the C ABI and is compared
  * BIT-EXACTLY with the CPU oracle restatement on a sample of the queries (the restatement takes seconds there), and
  * on ALL queries with the live reference (faiss v1.15.0 CPU indexes compiled into oracle/_ref) through
    check_knn(rtol = 1e-4): every label that differs must be a near-tie permutation or a tie at the k-th boundary
    (reference distances within 2e-5 relative of each other), otherwise the test fails; the counts are printed.
The larger fixtures of tests/golden (nb = 100k Flat, IVF4096 lists) are checked here against the HIP path as well."""
import time

import numpy as np
import pytest

import faiss_amd
from compare import check_knn
from faiss_amd.datasets import synthetic_more, synthetic_more_device
from oracle.pyoracle import METRIC_L2, Oracle, Ref, synthetic_dataset
from test_oracle_cpu import load_flat_case, load_ivf_case

pytestmark = pytest.mark.gpu
D_, NT, NB, NQ, K, NLIST, NPROBE = 128, 100000, 1000000, 10000, 100, 4096, 32


@pytest.fixture(scope="module")
def sift_shaped():
    return synthetic_dataset(D_, NT, NB, NQ, seed=1338)


def _report(name, st):
    print("%s: %d x %d labels, %d mismatches (%d near-tie swaps, %d boundary ties, 0 real), max rel dist err %.3g"
          % (name, st["n"], st["k"], st["label_mismatch"], st["tie_swaps"], st["boundary_ties"], st["max_rel_err"]))


# ------------------------------------------------------------------------------- fixtures of tests/golden at scale
def test_flat_100k_vs_reference_golden(res):
    z, xb, xq = load_flat_case("flat_l2_100k")
    idx = faiss_amd.GpuIndexFlatL2(res, xb.shape[1])
    idx.add(xb)
    for k in z["ks"]:
        D, I = idx.search(xq, int(k))
        assert idx.filter_stats()[0]
        st = check_knn(D, I, z["D_%d" % k], z["I_%d" % k], rtol=1e-4, name="flat 100k k=%d" % k)
        assert st["max_rel_err"] < 2e-5
        Do, Io = Oracle.flat_search(METRIC_L2, xb, xq[:32], int(k))
        check_knn(D[:32], I[:32], Do, Io, exact=True, name="flat 100k vs oracle")


@pytest.mark.parametrize("name", ["ivfflat_l2_4096", "ivfpq_l2_4096", "ivfflat_ip_4096", "ivfpq_ip_4096"])
def test_ivf4096_vs_reference_golden(res, name):
    """copy_lists of the reference's IVF4096 lists, search vs the reference's results; native add reproduces the
    reference's list sizes and ids exactly (codes up to argmin near-ties)."""
    c = load_ivf_case(name)
    z = c["z"]
    d, nlist = c["xb"].shape[1], z["centroids"].shape[0]

    def make():
        if c["kind"] == 0:
            i = faiss_amd.GpuIndexIVFFlat(res, d, nlist, c["metric"])
        else:
            i = faiss_amd.GpuIndexIVFPQ(res, d, nlist, c["M"], 8, c["metric"])
            i.copy_pq_centroids(c["pq"])
        i.copy_centroids(z["centroids"])
        i.nprobe = c["nprobe"]
        return i

    idx = make()
    idx.copy_lists(z["list_sizes"], c["codes"], z["list_ids"])
    D, I = idx.search(c["xq"], c["k"])
    st = check_knn(D, I, z["D"], z["I"], rtol=1e-4, name=name + " vs golden")
    assert st["max_rel_err"] < 2e-5
    _report(name, st)
    sel = np.r_[0:24]
    Do, Io, _, _ = Oracle.ivf_search(c["kind"], c["metric"], z["centroids"], z["list_sizes"], c["codes"], z["list_ids"],
                                     c["xq"][sel], c["nprobe"], c["k"], M=c["M"], pq=c["pq"])
    check_knn(D[sel], I[sel], Do, Io, exact=True, name=name + " vs oracle")
    # the list-major scans on the same lists (behind the f16 filter: the query-major bits; on the f32 matrix pipe: its
    # own restatement): vs the reference, and bit-exact vs the restatement that applies
    for mode in (idx.SCAN_LIST_MAJOR, idx.SCAN_LIST_MAJOR_F32):
        idx.set_scan_mode(mode)
        D2, I2 = idx.search(c["xq"], c["k"])
        arith = idx.last_scan_arith()
        assert arith == (1 if mode == idx.SCAN_LIST_MAJOR_F32 else 0)
        if arith == 0:
            assert np.array_equal(D2, D) and np.array_equal(I2, I)
        st = check_knn(D2, I2, z["D"], z["I"], rtol=1e-4, name=name + " list-major vs golden")
        assert st["max_rel_err"] < 2e-5
        Do, Io, _, _ = Oracle.ivf_search(c["kind"], c["metric"], z["centroids"], z["list_sizes"], c["codes"], z["list_ids"],
                                         c["xq"][sel], c["nprobe"], c["k"], M=c["M"], pq=c["pq"], arith=arith)
        check_knn(D2[sel], I2[sel], Do, Io, exact=True, name=name + " list-major (mode %d) vs oracle" % mode)
    nat = make()
    nat.add(c["xb"])
    sizes = np.array([nat.get_list_size(l) for l in range(nlist)], dtype=np.uint32)
    lids = np.concatenate([nat.get_list_ids(l) for l in range(nlist)])
    if np.array_equal(sizes, z["list_sizes"]):
        assert np.array_equal(lids, z["list_ids"])
    else:  # a coarse-assignment near-tie (MKL sgemm vs our fmaf chain) moved a vector to the neighbouring list
        assert (sizes != z["list_sizes"]).sum() <= 8
    if c["kind"] == 1 and np.array_equal(lids, z["list_ids"]):
        codes = np.concatenate([nat.get_list_codes(l) for l in range(nlist)])
        assert (codes == c["codes"]).mean() > 0.9995


# ------------------------------------------------------------------------------- BASELINE.json configs[1]
def test_flat_1m_x_10k_vs_oracle_and_live_reference(res, sift_shaped):
    _, xb, xq = sift_shaped
    idx = faiss_amd.GpuIndexFlatL2(res, D_)
    idx.add(xb)
    D, I = idx.search(xq, K)
    used, novf = idx.filter_stats()
    assert used and novf < 100
    assert (np.diff(D, axis=1) >= 0).all()
    # (1) bit-exact against the restatement on 64 sampled queries
    sel = np.random.RandomState(1).choice(NQ, 64, replace=False)
    Do, Io = Oracle.flat_search(METRIC_L2, xb, xq[sel], K)
    check_knn(D[sel], I[sel], Do, Io, exact=True, name="flat 1M vs oracle")
    # (2) the independent fp32 MFMA scan agrees bit for bit on every query
    idx.set_use_filter_kernel(False)
    D0, I0 = idx.search(xq, K)
    assert np.array_equal(I, I0) and np.array_equal(D, D0)
    # (3) every query against the live reference, every label mismatch classified
    if not Ref.available():
        pytest.skip("oracle/_ref not shipped: live-reference leg skipped (oracle leg passed)")
    ref = Ref.index_factory(D_, "Flat")
    ref.add(xb)
    t0 = time.time()
    Dr, Ir = ref.search(xq, K)
    print("reference IndexFlatL2: %.1f s" % (time.time() - t0))
    st = check_knn(D, I, Dr, Ir, rtol=1e-4, max_tie_frac=1e-3, name="flat 1M x 10k vs live reference")
    _report("flat 1M x 10k", st)
    assert (I[:, 0] == Ir[:, 0]).mean() > 0.9999


# ------------------------------------------------------------------------------- IVFFlat beyond d = 128 (round 4)
@pytest.mark.parametrize("d", [256, 384])
def test_ivfflat_wide_rows_vs_live_reference(res, d):
    """d > 128 takes the 16 / 24 k-step instantiations of the list-major filter sweeps (two / one query block per work item):
    IVF1024,Flat over 200 000 rows of d coordinates, 4000 queries, nprobe 16, k 100 -- ALL queries against the live
    reference index holding the same quantizer (north-star tolerance), list-major == query-major on all of them, a sample
    bit-exact against the oracle."""
    if not Ref.available():
        pytest.skip("oracle/_ref not shipped")
    nlist, nb, nq, nprobe, k = 1024, 200000, 4000, 16, 100
    xt, xb, xq = synthetic_dataset(d, 40000, nb, nq, seed=77 + d)
    g = faiss_amd.GpuIndexIVFFlat(res, d, nlist, METRIC_L2)
    g.train(xt)
    cent = g.get_centroids()
    ref = Ref.index_factory(d, "IVF%d,Flat" % nlist)
    ref.set_centroids(cent)
    ref.add(xb)
    ref.set_nprobe(nprobe)
    Dr, Ir = ref.search(xq, k)
    sizes, codes, lids = ref.lists()
    g.copy_lists(sizes, codes, lids)
    g.nprobe = nprobe
    assert g.list_major_rule(nq, nprobe, k)
    D, I = g.search(xq, k)
    assert g.scan_info()[1] == 2 and g.last_scan_arith() == 0
    st = check_knn(D, I, Dr, Ir, rtol=1e-4, max_tie_frac=2e-3, name="ivfflat d=%d vs live reference" % d)
    _report("ivfflat d=%d 200k x 4000 (list-major behind the f16 filter)" % d, st)
    g.set_scan_mode(g.SCAN_QUERY_MAJOR)
    D0, I0 = g.search(xq, k)
    assert g.scan_info()[1] == 1 and np.array_equal(D, D0) and np.array_equal(I, I0)
    sel = np.random.RandomState(3).choice(nq, 24, replace=False)
    Do, Io, _, _ = Oracle.ivf_search(0, METRIC_L2, cent, sizes, codes, lids, xq[sel], nprobe, k, arith=0)
    check_knn(D[sel], I[sel], Do, Io, exact=True, name="ivfflat d=%d sample vs oracle" % d)


# ------------------------------------------------------------------------------- IVF4096 at nb = 1M (configs[2]/[3] shape)
@pytest.mark.parametrize("kind", ["ivfflat", "ivfpq"])
def test_ivf4096_1m_vs_oracle_and_live_reference(res, sift_shaped, kind):
    """The reference index (index_factory IVF4096,Flat / IVF4096,PQ64) gets the quantizers the GPU index trained, is
    filled by its own add(); the GPU index is then loaded with the reference's lists (copy_lists = copyFrom), so both
    sides scan identical lists.  nprobe 32, k 100, all 10 000 queries vs the reference, 64 of them vs the oracle."""
    if not Ref.available():
        pytest.skip("oracle/_ref not shipped")
    xt, xb, xq = sift_shaped
    if kind == "ivfflat":
        g = faiss_amd.GpuIndexIVFFlat(res, D_, NLIST, METRIC_L2)
    else:
        g = faiss_amd.GpuIndexIVFPQ(res, D_, NLIST, 64, 8, METRIC_L2)
    g.train(xt)
    cent = g.get_centroids()
    pq = g.get_pq_centroids() if kind == "ivfpq" else None
    ref = Ref.index_factory(D_, "IVF4096,Flat" if kind == "ivfflat" else "IVF4096,PQ64")
    if kind == "ivfflat":
        ref.set_centroids(cent)
    else:
        ref.set_trained(cent, pq)
    t0 = time.time()
    ref.add(xb)
    ref.set_nprobe(NPROBE)
    Dr, Ir = ref.search(xq, K)
    print("reference %s add + search: %.1f s" % (kind, time.time() - t0))
    sizes, codes, lids = ref.lists()
    # native add builds the same lists (sizes/ids exactly unless a coarse-assignment near-tie moves a vector)
    g.add(xb)
    nat_sizes = np.array([g.get_list_size(l) for l in range(NLIST)], dtype=np.uint32)
    moved = int(np.abs(nat_sizes.astype(np.int64) - sizes.astype(np.int64)).sum())
    print("native add vs reference lists: %d list-size differences" % moved)
    assert moved <= 20
    g.nprobe = NPROBE
    Dn, In = g.search(xq[:512], K)
    # the reference's lists on the device: identical scan inputs on both sides
    g2 = (faiss_amd.GpuIndexIVFFlat(res, D_, NLIST, METRIC_L2) if kind == "ivfflat"
          else faiss_amd.GpuIndexIVFPQ(res, D_, NLIST, 64, 8, METRIC_L2))
    g2.copy_centroids(cent)
    if pq is not None:
        g2.copy_pq_centroids(pq)
    g2.copy_lists(sizes, codes, lids)
    g2.nprobe = NPROBE
    kd = 0 if kind == "ivfflat" else 1
    M = 64 if pq is not None else 0
    sel = np.random.RandomState(2).choice(NQ, 64, replace=False)
    # ---- the scan the library picks for this batch (10 000 queries x 32 probes over 4096 lists: list-major for IVFFlat)
    D, I = g2.search(xq, K)
    arith = g2.last_scan_arith()
    # (both index types take the list-major scan behind the f16 filter at this shape: 5 GB of code bytes / 40 GB of vectors)
    assert arith == 0 and g2.scan_info()[1] == 2
    st = check_knn(D, I, Dr, Ir, rtol=1e-4, max_tie_frac=2e-3, name="%s 1M vs live reference" % kind)
    _report("%s 1M x 10k (automatic scan, arith %d)" % (kind, arith), st)
    assert (np.diff(D, axis=1) >= 0).all()
    Do, Io, _, _ = Oracle.ivf_search(kd, METRIC_L2, cent, sizes, codes, lids, xq[sel], NPROBE, K, M=M, pq=pq, arith=arith)
    check_knn(D[sel], I[sel], Do, Io, exact=True, name="%s 1M vs oracle" % kind)
    # ---- both scans explicitly, each against the reference on all queries and against its own restatement on the sample
    res_by_mode = {}
    for mode in (g2.SCAN_QUERY_MAJOR, g2.SCAN_LIST_MAJOR, g2.SCAN_LIST_MAJOR_F32):
        g2.set_scan_mode(mode)
        Dm, Im = g2.search(xq, K)
        assert g2.scan_info()[1] == min(mode, 2) and g2.last_scan_arith() == (1 if mode == g2.SCAN_LIST_MAJOR_F32 else 0)
        st = check_knn(Dm, Im, Dr, Ir, rtol=1e-4, max_tie_frac=2e-3, name="%s 1M scan mode %d vs live reference" % (kind, mode))
        _report("%s 1M x 10k (scan mode %d)" % (kind, mode), st)
        Do, Io, _, _ = Oracle.ivf_search(kd, METRIC_L2, cent, sizes, codes, lids, xq[sel], NPROBE, K, M=M, pq=pq,
                                         arith=g2.last_scan_arith())
        check_knn(Dm[sel], Im[sel], Do, Io, exact=True, name="%s 1M scan mode %d vs oracle" % (kind, mode))
        res_by_mode[mode] = (Dm, Im)
    print("queries redone so far (segment overflow / fp16 range):", g2.scan_info()[2])
    # behind the f16 filter the list-major scan returns the query-major bits: ALL 10 000 x 100 results
    assert np.array_equal(res_by_mode[g2.SCAN_LIST_MAJOR][0], res_by_mode[g2.SCAN_QUERY_MAJOR][0])
    assert np.array_equal(res_by_mode[g2.SCAN_LIST_MAJOR][1], res_by_mode[g2.SCAN_QUERY_MAJOR][1])
    Dq_, Iq_ = res_by_mode[g2.SCAN_QUERY_MAJOR]
    # native lists: same results wherever the lists agree (PQ codes may differ in argmin near-ties)
    agree = (In[:, 0] == Iq_[:512, 0]).mean()
    assert agree > 0.99, agree
    # unfused cross-check path and search_preassigned reproduce the fused query-major scan bit for bit
    g2.set_scan_mode(g2.SCAN_QUERY_MAJOR)
    g2.set_use_fused_scan(False)
    D0, I0 = g2.search(xq[:400], K)
    g2.set_use_fused_scan(True)
    assert np.array_equal(I0, Iq_[:400]) and np.array_equal(D0, Dq_[:400])
    Dc, Ic = g2.quantizer_search(xq[:400], NPROBE)
    D1, I1 = g2.search_preassigned(xq[:400], K, Ic, Dc)
    assert np.array_equal(I1, Iq_[:400]) and np.array_equal(D1, Dq_[:400])
    # ... and the list-major ones
    Dc, Ic = g2.quantizer_search(xq, NPROBE)
    for mode in (g2.SCAN_LIST_MAJOR, g2.SCAN_LIST_MAJOR_F32):
        g2.set_scan_mode(mode)
        D2, I2 = g2.search_preassigned(xq, K, Ic, Dc)
        assert np.array_equal(I2, res_by_mode[mode][1]) and np.array_equal(D2, res_by_mode[mode][0])


def test_ivfsq8_4096_1m_vs_oracle_and_live_reference(res, sift_shaped):
    """IVF4096,SQ8 (QT_8bit, residual encoding) at the metric's database size: the reference index with the GPU-trained
    quantizers is filled by its own add(), its lists are loaded into a second GPU index (copyFrom); nprobe 32, k 100.
    The automatic choice for 10 000 queries is the list-major scan behind the f16 filter (round 5: the bits of the query-major
    scan); the query-major scan, the filter path and round 3's f32 list-major scan are compared with the reference on ALL
    queries and bit-exactly with their own restatement on 48 of them."""
    if not Ref.available():
        pytest.skip("oracle/_ref not shipped")
    from faiss_amd import ScalarQuantizer as SQ
    xt, xb, xq = sift_shaped
    g = faiss_amd.GpuIndexIVFScalarQuantizer(res, D_, NLIST, SQ.QT_8bit, METRIC_L2, True)
    g.train(xt)
    cent, trained = g.get_centroids(), g.get_trained()
    ref = Ref.index_factory(D_, "IVF4096,SQ8")
    ref.set_sq_trained(cent, trained)
    ref.add(xb)
    ref.set_nprobe(NPROBE)
    Dr, Ir = ref.search(xq, K)
    sizes, codes, lids = ref.lists()
    g2 = faiss_amd.GpuIndexIVFScalarQuantizer(res, D_, NLIST, SQ.QT_8bit, METRIC_L2, True)
    g2.copy_centroids(cent)
    g2.copy_trained(trained)
    g2.copy_lists(sizes, codes, lids)
    g2.nprobe = NPROBE
    vmin, vdiff = Oracle.sq_unpack(SQ.QT_8bit, D_, trained)
    sel = np.random.RandomState(4).choice(NQ, 48, replace=False)
    D, I = g2.search(xq, K)
    assert g2.scan_info()[1] == 2 and g2.last_scan_arith() == 0, "10 000 queries x 32 probes over 4096 lists: the list-major scan behind the filter"
    for mode in (g2.SCAN_QUERY_MAJOR, g2.SCAN_LIST_MAJOR, g2.SCAN_LIST_MAJOR_F32):
        g2.set_scan_mode(mode)
        Dm, Im = g2.search(xq, K)
        assert g2.scan_info()[1] == min(mode, 2)
        st = check_knn(Dm, Im, Dr, Ir, rtol=1e-4, max_tie_frac=2e-3, name="ivfsq8 1M scan mode %d vs live reference" % mode)
        _report("ivfsq8 1M x 10k (scan mode %d)" % mode, st)
        assert st["max_rel_err"] < 2e-5
        assert (np.diff(Dm, axis=1) >= 0).all()
        Do, Io = Oracle.ivfsq_search(SQ.QT_8bit, True, METRIC_L2, cent, sizes, codes, lids, vmin, vdiff, xq[sel], NPROBE, K,
                                     arith=g2.last_scan_arith())
        check_knn(Dm[sel], Im[sel], Do, Io, exact=True, name="ivfsq8 1M scan mode %d vs oracle" % mode)
        if mode == g2.SCAN_LIST_MAJOR:
            assert np.array_equal(Dm, D) and np.array_equal(Im, I)
    assert g2.scan_info()[2] == 0, "no query of this batch should overflow its candidate segment"
    # native add: the lists the reference builds (up to coarse near-ties), same nearest neighbour
    g.add(xb)
    g.nprobe = NPROBE
    Dn, In = g.search(xq[:2048], K)
    assert (In[:, 0] == Ir[:2048, 0]).mean() > 0.99


# ------------------------------------------------------------------------------- BASELINE.json configs[2]: nb = 10M
def test_ivfflat_10m_sample_vs_oracle(res):
    """GpuIndexIVFFlat nlist=4096 nprobe=32 at nb = 10M (added in 1M-row chunks, 20 calls of add), a 32-query sample:
    the lists those queries probe are read back from the device and the oracle restatement scans them -- distances and
    labels bit-exact; the returned rows' exact distances are re-derived in float64; a sample of the stored vectors sits
    in the list the restatement's coarse assignment gives."""
    xt, xb0, xq, dmap = synthetic_dataset(D_, NT, 500000, 32, seed=1338, return_map=True)
    idx = faiss_amd.GpuIndexIVFFlat(res, D_, NLIST, METRIC_L2)
    idx.train(xt)
    cent = idx.get_centroids()
    keep = {}  # a few chunks stay on the host for the float64 check
    idx.add(xb0)
    keep[0] = xb0
    n = len(xb0)
    for chunk in range(1, 20):
        xbc = synthetic_more(dmap, 500000, seed=1338 + chunk)
        idx.add(xbc)
        if chunk in (7, 19):
            keep[chunk] = xbc
        n += len(xbc)
    assert idx.ntotal == n == 10000000 and idx.stored_vectors == n
    used, holes, alloc = idx.arena_stats()
    print("arena rows: used %d, holes %d, allocated %d (%.2fx of the vectors)" % (used, holes, alloc, alloc / n))
    assert used - holes < 1.3 * n and alloc < 2.5 * n
    idx.nprobe = NPROBE
    D, I = idx.search(xq, K)
    assert (np.diff(D, axis=1) >= 0).all() and (I >= 0).all() and (I < n).all()
    # oracle on the probed lists only
    Dq, Iq = idx.quantizer_search(xq, NPROBE)
    probed = np.unique(Iq)
    sizes = np.zeros(NLIST, dtype=np.uint32)
    codes, ids = [], []
    for l in probed:
        sizes[l] = idx.get_list_size(int(l))
        codes.append(idx.get_list_codes(int(l)))
        ids.append(idx.get_list_ids(int(l)))
    codes, ids = np.concatenate(codes), np.concatenate(ids)
    Do, Io, cD, cI = Oracle.ivf_search(0, METRIC_L2, cent, sizes, codes, ids, xq, NPROBE, K)
    assert np.array_equal(cI, Iq) and np.array_equal(cD, Dq)
    check_knn(D, I, Do, Io, exact=True, name="ivfflat 10M vs oracle")
    # the list-major scan on the same queries: lists of ~2400 rows = three row chunks each (pass 1 sees the first chunk of
    # the leading lists, the rest goes through pass 2)
    for mode in (idx.SCAN_LIST_MAJOR, idx.SCAN_LIST_MAJOR_F32):
        idx.set_scan_mode(mode)
        D2, I2 = idx.search(xq, K)
        arith = idx.last_scan_arith()
        assert idx.scan_info()[1] == 2 and arith == (1 if mode == idx.SCAN_LIST_MAJOR_F32 else 0)
        Do2, Io2, _, _ = Oracle.ivf_search(0, METRIC_L2, cent, sizes, codes, ids, xq, NPROBE, K, arith=arith)
        check_knn(D2, I2, Do2, Io2, exact=True, name="ivfflat 10M list-major (mode %d) vs oracle" % mode)
        check_knn(D2, I2, D, I, rtol=1e-4, name="ivfflat 10M list-major vs query-major")
        if arith == 0:
            assert np.array_equal(D2, D) and np.array_equal(I2, I)
    idx.set_scan_mode(idx.SCAN_AUTO)
    # exact distances of returned rows that live in the kept chunks
    for chunk, xbc in keep.items():
        lo = chunk * 500000
        m = (I >= lo) & (I < lo + 500000)
        qi, ri = np.nonzero(m)
        ex = ((xq[qi].astype(np.float64) - xbc[I[qi, ri] - lo].astype(np.float64)) ** 2).sum(-1)
        assert np.allclose(D[qi, ri], ex, rtol=1e-5, atol=1e-4)
    # add path: stored ids sit in the list of their nearest centroid
    rows = np.arange(0, 500000, 997)
    lab = Oracle.ivf_assign(METRIC_L2, cent, keep[19][rows])
    for r, l in zip(rows[:40], lab[:40]):
        assert (19 * 500000 + r) in set(idx.get_list_ids(int(l)).tolist())


# ------------------------------------------------------------------------------- BASELINE.json configs[3]: nb = 100M
def test_ivfpq_100m_sample_vs_oracle(res):
    """GpuIndexIVFPQ nlist=4096 PQ64x8 nprobe=32 at nb = 100M on one MI355X (BASELINE.json configs[3]): the database is
    drawn chunk by chunk ON THE DEVICE (faiss_amd.datasets.synthetic_more_device, 1M rows per add call) -- the codes
    take 6.4 GB of HBM, the host never holds more than the first chunk.  All 10 000 queries are searched with the three
    scans (query-major, list-major behind the f16 filter, list-major on the f32 matrix pipe); a 32-query sample is
    compared BIT-EXACTLY with the oracle restatement that applies (arith 0 / 0 / 1) run on the ~900 probed lists read back
    from the device (their codes and ids); the coarse assignment of the sample is bit-exact too; all results ordered,
    labels valid and distinct; the filter scan equals the query-major scan on ALL queries."""
    import torch
    dev = torch.device("cuda", 0)
    nb, M, nsample = 100000000, 64, 32
    xt, xb0, xq, dmap = synthetic_dataset(D_, NT, 1000000, NQ, seed=1338, return_map=True)
    idx = faiss_amd.GpuIndexIVFPQ(res, D_, NLIST, M, 8, METRIC_L2)
    idx.train(xt)
    t0 = time.time()
    idx.add(xb0)
    for chunk in range(1, nb // 1000000):
        xbc = synthetic_more_device(dmap, 1000000, 1338 + chunk, dev)
        idx.add_ptr(1000000, xbc.data_ptr())
        del xbc
    print("100M rows added in %.1f s" % (time.time() - t0))
    assert idx.ntotal == nb and idx.stored_vectors == nb
    used, holes, alloc = idx.arena_stats()
    assert used - holes < 1.3 * nb and alloc < 2.0 * nb
    idx.nprobe = NPROBE
    cent, pq = idx.get_centroids(), idx.get_pq_centroids()
    sel = np.random.RandomState(11).choice(NQ, nsample, replace=False)
    Dq, Iq = idx.quantizer_search(xq[sel], NPROBE)
    sizes = np.zeros(NLIST, dtype=np.uint32)
    codes, ids = [], []
    for l in np.unique(Iq):
        sizes[l] = idx.get_list_size(int(l))
        codes.append(idx.get_list_codes(int(l)))
        ids.append(idx.get_list_ids(int(l)))
    codes, ids = np.concatenate(codes), np.concatenate(ids)
    print("%d probed lists, %d entries read back" % (len(np.unique(Iq)), len(ids)))
    results = {}
    for mode in (idx.SCAN_LIST_MAJOR, idx.SCAN_QUERY_MAJOR, idx.SCAN_LIST_MAJOR_F32):
        idx.set_scan_mode(mode)
        t0 = time.time()
        D, I = idx.search(xq, K)
        arith = idx.last_scan_arith()
        print("scan mode %d: %.3f s for 10 000 queries (host buffers), %d queries redone so far" % (mode, time.time() - t0, idx.scan_info()[2]))
        assert idx.scan_info()[1] == min(mode, 2) and arith == (1 if mode == idx.SCAN_LIST_MAJOR_F32 else 0)
        assert (np.diff(D, axis=1) >= 0).all() and (I >= 0).all() and (I < nb).all()
        srt = np.sort(I, axis=1)
        assert (srt[:, 1:] != srt[:, :-1]).all(), "a label was returned twice"
        Do, Io, cD, cI = Oracle.ivf_search(1, METRIC_L2, cent, sizes, codes, ids, xq[sel], NPROBE, K, M=M, pq=pq, arith=arith)
        assert np.array_equal(cI, Iq) and np.array_equal(cD, Dq)
        check_knn(D[sel], I[sel], Do, Io, exact=True, name="ivfpq 100M scan mode %d vs oracle" % mode)
        results[mode] = (D, I)
    # behind the f16 filter the list-major scan returns the query-major bits: ALL 10 000 x 100 results at nb = 100M
    assert np.array_equal(results[idx.SCAN_LIST_MAJOR][0], results[idx.SCAN_QUERY_MAJOR][0])
    assert np.array_equal(results[idx.SCAN_LIST_MAJOR][1], results[idx.SCAN_QUERY_MAJOR][1])
    idx.set_scan_mode(idx.SCAN_AUTO)
    D, I = idx.search(xq, K)
    assert idx.scan_info()[1] == 2 and idx.last_scan_arith() == 0, "configs[3] is a list-major workload"
    assert np.array_equal(D, results[idx.SCAN_LIST_MAJOR][0]) and np.array_equal(I, results[idx.SCAN_LIST_MAJOR][1])
    # the f32 list-major scan sums in another order (decoded residuals on the matrix pipe vs the ADC table on a
    # power-of-two grid): rank by rank the distances agree within the tolerance, the labels outside near-tie groups
    Dl, Il = results[idx.SCAN_LIST_MAJOR_F32]
    Dm, Im = results[idx.SCAN_QUERY_MAJOR]
    rel = np.abs(Dl - Dm) / np.maximum(np.abs(Dm), 1e-30)
    print("ivfpq 100M f32 list-major vs query-major: labels equal %.5f, max rel distance difference rank by rank %.3g"
          % ((Il == Im).mean(), rel.max()))
    assert rel.max() < 1e-4 and (Il == Im).mean() > 0.98 and (Il[:, 0] == Im[:, 0]).mean() > 0.995
