"""GPU parity tests (pytest -m gpu): every call goes through the C ABI (faiss_amd ctypes
mirror) and is compared with the CPU oracle restatement -- BIT-EXACT distances and labels, the
oracle using the same summation order as the kernels -- and with the golden outputs of the
real reference within the north-star tolerance (1e-4 relative, labels exact outside near-tie
groups).  Test matrix modelled on faiss/gpu/test/TestGpuIndexFlat.cpp:46-109 (L2/IP, k=1..2048,
odd dims, tiny/empty inputs) and TestGpuIndexIVF{Flat,PQ}.cpp (copyFrom, add vs CPU lists)."""
import numpy as np
import pytest

import faiss_amd
from compare import check_knn
from oracle.pyoracle import METRIC_INNER_PRODUCT, METRIC_L2, Oracle, Ref, integer_dataset, synthetic_dataset
from test_oracle_cpu import load_flat_case, load_ivf_case

pytestmark = pytest.mark.gpu
FMAX = np.finfo(np.float32).max


@pytest.mark.parametrize("metric", [METRIC_L2, METRIC_INNER_PRODUCT])
@pytest.mark.parametrize("d,nb,nq", [(128, 300, 70), (40, 1000, 33), (264, 200, 10), (256, 500, 40)])
def test_mfma_distance_matrix_bit_exact(res, metric, d, nb, nq):
    _, xb, xq = synthetic_dataset(d, 0, nb, nq, seed=d)
    idx = faiss_amd.GpuIndexFlat(res, d, metric)
    idx.add(xb)
    G = idx.pairwise_distances(xq)
    assert np.array_equal(G, Oracle.pairwise(metric, xb, xq))


FLAT_SHAPES = [(128, 5000, 300, 10), (128, 5000, 300, 1), (128, 20000, 64, 100), (40, 3000, 50, 7),
               (64, 50, 5, 8), (128, 3000, 1, 16), (128, 40000, 513, 128), (32, 9000, 100, 2048),
               (384, 4000, 30, 20), (100, 70000, 257, 50)]


@pytest.mark.parametrize("simple", [False, True])
@pytest.mark.parametrize("metric", [METRIC_L2, METRIC_INNER_PRODUCT])
@pytest.mark.parametrize("d,nb,nq,k", FLAT_SHAPES)
def test_flat_search_bit_exact(res, simple, metric, d, nb, nq, k):
    if simple and nb * nq > 5_000_000:
        pytest.skip("cross-check kernel only at small sizes")
    _, xb, xq = synthetic_dataset(d, 0, nb, nq, seed=nb + k)
    idx = faiss_amd.GpuIndexFlat(res, d, metric)
    idx.set_use_simple_kernel(simple)
    idx.add(xb)
    D, I = idx.search(xq, k)
    Do, Io = Oracle.flat_search(metric, xb, xq, k)
    check_knn(D, I, Do, Io, exact=True, name="flat")


def test_flat_integer_ties_exact_vs_reference_golden(res):
    """Tie rule: lowest ids win and come first -- identical to the reference CPU output."""
    z, xb, xq = load_flat_case("flat_l2_int_ties")
    idx = faiss_amd.GpuIndexFlatL2(res, xb.shape[1])
    idx.add(xb)
    for k in z["ks"]:
        D, I = idx.search(xq, int(k))
        check_knn(D, I, z["D_%d" % k], z["I_%d" % k], exact=True, name="ties k=%d" % k)


@pytest.mark.parametrize("name", ["flat_l2_small", "flat_ip_small", "flat_l2_blas"])
def test_flat_vs_reference_golden(res, name):
    z, xb, xq = load_flat_case(name)
    idx = faiss_amd.GpuIndexFlat(res, xb.shape[1], int(z["metric"]))
    idx.add(xb)
    for k in z["ks"]:
        D, I = idx.search(xq, int(k))
        st = check_knn(D, I, z["D_%d" % k], z["I_%d" % k], rtol=1e-4, name="%s k=%d" % (name, k))
        assert st["max_rel_err"] < 2e-5


def test_flat_incremental_add_reset_reconstruct(res):
    _, xb, xq = synthetic_dataset(72, 0, 5000, 20, seed=2)
    idx = faiss_amd.GpuIndexFlatL2(res, 72)
    idx.add(xb[:1234])
    idx.add(xb[1234:1235])
    idx.add(xb[1235:])
    assert idx.ntotal == 5000
    D, I = idx.search(xq, 9)
    check_knn(D, I, *Oracle.flat_search(METRIC_L2, xb, xq, 9), exact=True, name="incremental")
    assert np.array_equal(idx.reconstruct(777), xb[777])
    assert np.array_equal(idx.reconstruct_n(100, 50), xb[100:150])
    idx.reset()
    assert idx.ntotal == 0
    D, I = idx.search(xq, 3)
    assert (I == -1).all() and (D == FMAX).all()


def test_flat_k_larger_than_ntotal_pads(res):
    xb, xq = integer_dataset(16, 5, 3, seed=1)
    for metric, pad in ((METRIC_L2, FMAX), (METRIC_INNER_PRODUCT, -FMAX)):
        idx = faiss_amd.GpuIndexFlat(res, 16, metric)
        idx.add(xb)
        D, I = idx.search(xq, 8)
        Do, Io = Oracle.flat_search(metric, xb, xq, 8)
        check_knn(D, I, Do, Io, exact=True, name="pad")
        assert (I[:, 5:] == -1).all() and (D[:, 5:] == pad).all()


def test_flat_empty_query_batch_and_errors(res):
    idx = faiss_amd.GpuIndexFlatL2(res, 8)
    idx.add(np.zeros((4, 8), "float32"))
    D, I = idx.search(np.zeros((0, 8), "float32"), 3)
    assert D.shape == (0, 3)
    with pytest.raises(faiss_amd.FaissAmdError):
        idx.search(np.zeros((1, 8), "float32"), 2049)  # k limit, faiss/gpu/utils/DeviceDefs.cuh:39
    with pytest.raises(faiss_amd.FaissAmdError):
        idx.search(np.zeros((1, 8), "float32"), 0)
    with pytest.raises(ValueError):
        idx.add(np.zeros((1, 9), "float32"))
    with pytest.raises(faiss_amd.FaissAmdError):
        faiss_amd.GpuIndexFlat(res, 8, 99)  # not a value of faiss/MetricType.h:31-52


def test_flat_nan_query_returns_no_result(res):
    """A NaN query admits nothing on the CPU reference (strict compare), TestGpuIndexFlat QueryNaN."""
    _, xb, xq = synthetic_dataset(32, 0, 500, 4, seed=3)
    xq = xq.copy()
    xq[1, 5] = np.nan
    idx = faiss_amd.GpuIndexFlatL2(res, 32)
    idx.add(xb)
    D, I = idx.search(xq, 5)
    assert (I[1] == -1).all() and (I[0] >= 0).all()
    check_knn(D, I, *Oracle.flat_search(METRIC_L2, xb, xq, 5), exact=True, name="nan")


def test_flat_large_query_batch_tiles(res):
    """More queries than one scratch tile (reference test LargeBatch, >= 65536 queries)."""
    _, xb, xq = synthetic_dataset(16, 0, 300, 70000, seed=8)
    idx = faiss_amd.GpuIndexFlatL2(res, 16)
    idx.add(xb)
    res.setTempMemory(64 << 20)
    try:
        D, I = idx.search(xq, 4)
    finally:
        res.setTempMemory(4 << 30)
    sel = np.r_[0:50, 30000:30050, 69950:70000]
    Do, Io = Oracle.flat_search(METRIC_L2, xb, xq[sel], 4)
    check_knn(D[sel], I[sel], Do, Io, exact=True, name="large batch")


# ------------------------------------------------------------------------------- IVF
def _make_ivf(res, c):
    z = c["z"]
    d = c["xb"].shape[1]
    nlist = z["centroids"].shape[0]
    if c["kind"] == 0:
        idx = faiss_amd.GpuIndexIVFFlat(res, d, nlist, c["metric"])
    else:
        idx = faiss_amd.GpuIndexIVFPQ(res, d, nlist, c["M"], 8, c["metric"])
        idx.copy_pq_centroids(c["pq"])
    idx.copy_centroids(z["centroids"])
    return idx


@pytest.mark.parametrize("name", ["ivfflat_l2", "ivfflat_ip", "ivfpq_l2", "ivfpq_ip"])
def test_ivf_copy_from_reference_and_search(res, name):
    """copyFrom a CPU-trained reference index (TestGpuIndexIVFPQ.cpp:149-168 pattern)."""
    c = load_ivf_case(name)
    z = c["z"]
    idx = _make_ivf(res, c)
    idx.copy_lists(z["list_sizes"], c["codes"], z["list_ids"])
    assert idx.ntotal == len(z["list_ids"])
    for nprobe in (1, c["nprobe"], 64):  # 64 > nlist: clamps like the reference
        idx.nprobe = nprobe
        D, I = idx.search(c["xq"], c["k"])
        Do, Io, _, _ = Oracle.ivf_search(c["kind"], c["metric"], z["centroids"], z["list_sizes"], c["codes"],
                                         z["list_ids"], c["xq"], nprobe, c["k"], M=c["M"], pq=c["pq"])
        check_knn(D, I, Do, Io, exact=True, name=name + " vs oracle")
        if nprobe == c["nprobe"]:
            st = check_knn(D, I, z["D"], z["I"], rtol=1e-4, name=name + " vs golden")
            assert st["max_rel_err"] < 2e-5


@pytest.mark.parametrize("name", ["ivfflat_l2", "ivfpq_l2", "ivfpq_ip"])
def test_ivf_add_builds_reference_lists(res, name):
    """add_with_ids in several batches produces byte-identical lists (testIVFEquality,
    faiss/gpu/test/TestUtils.h:111-142): sizes, ids and codes, in insertion order."""
    c = load_ivf_case(name)
    z = c["z"]
    idx = _make_ivf(res, c)
    n = len(c["xb"])
    cuts = [0, n // 5, n // 5 + 1, n // 2, n]
    for a, b in zip(cuts[:-1], cuts[1:]):
        idx.add_with_ids(c["xb"][a:b], c["ids"][a:b])
    nlist = z["centroids"].shape[0]
    sizes, codes, lids, _ = Oracle.build_ivf_lists(c["kind"], c["metric"], z["centroids"], c["xb"], ids=c["ids"],
                                                   pq=c["pq"])
    assert np.array_equal(np.array([idx.get_list_size(l) for l in range(nlist)], dtype=np.uint32), sizes)
    assert np.array_equal(np.concatenate([idx.get_list_ids(l) for l in range(nlist)]), lids)
    got = np.concatenate([idx.get_list_codes(l) for l in range(nlist)])
    assert np.array_equal(got.reshape(-1), codes.reshape(-1))
    # and against the real reference's lists (golden): identical sizes / ids, codes up to argmin near-ties
    assert np.array_equal(sizes, z["list_sizes"]) and np.array_equal(lids, z["list_ids"])
    idx.nprobe = c["nprobe"]
    D, I = idx.search(c["xq"], c["k"])
    check_knn(D, I, z["D"], z["I"], rtol=1e-4, name=name + " add+search vs golden")


def test_ivf_untrained_and_errors(res):
    idx = faiss_amd.GpuIndexIVFFlat(res, 16, 8, METRIC_L2)
    assert not idx.is_trained
    with pytest.raises(faiss_amd.FaissAmdError):
        idx.add(np.zeros((3, 16), "float32"))
    with pytest.raises(faiss_amd.FaissAmdError):
        idx.search(np.zeros((3, 16), "float32"), 2)
    with pytest.raises(faiss_amd.FaissAmdError):
        faiss_amd.GpuIndexIVFPQ(res, 30, 8, 8, 8, METRIC_L2)  # d % M != 0
    with pytest.raises(faiss_amd.FaissAmdError):
        faiss_amd.GpuIndexIVFPQ(res, 32, 8, 8, 6, METRIC_L2)  # only 8-bit codes (GpuIndexIVFPQ.cu:574-592)
    with pytest.raises(faiss_amd.FaissAmdError):
        idx.nprobe = 5000


def test_ivf_native_train_recall(res):
    """Train / add / search entirely on the GPU; recall threshold in the style of
    tests/test_ivfpq_indexing.cpp and tests/test_index_accuracy.py."""
    xt, xb, xq = synthetic_dataset(64, 20000, 100000, 500, seed=4)
    flat = faiss_amd.GpuIndexFlatL2(res, 64)
    flat.add(xb)
    _, gt = flat.search(xq, 1)
    ivf = faiss_amd.GpuIndexIVFFlat(res, 64, 256, METRIC_L2)
    ivf.train(xt)
    ivf.add(xb)
    assert ivf.ntotal == 100000 and sum(ivf.get_list_size(l) for l in range(256)) == 100000
    ivf.nprobe = 256  # probing every list == exhaustive search: same label sets as Flat
    D, I = ivf.search(xq[:50], 10)
    Df, If = flat.search(xq[:50], 10)
    check_knn(D, I, Df, If, rtol=1e-4, tie_rtol=1e-4, name="ivfflat nprobe=nlist vs flat")
    ivf.nprobe = 16
    _, I = ivf.search(xq, 10)
    assert (I[:, :1] == gt).mean() > 0.9
    pq = faiss_amd.GpuIndexIVFPQ(res, 64, 256, 16, 8, METRIC_L2)
    pq.train(xt)
    pq.add(xb)
    pq.nprobe = 16
    _, I = pq.search(xq, 10)
    assert (I == gt).any(axis=1).mean() > 0.8


def test_kmeans_objective_matches_reference(res):
    """faiss/gpu/test/test_gpu_basics.py:117-133: GPU k-means objective close to the CPU one."""
    # 10 000 <= 40 * 256 points: no subsampling (faiss/Clustering.cpp subsample_training_set), so
    # the recorded objective and the oracle's are sums over the same point set
    xt, _, _ = synthetic_dataset(32, 10000, 0, 0, seed=3)
    cent, obj = faiss_amd.kmeans(res, xt, 40, niter=10, seed=1)
    assert np.all(np.diff(obj) <= obj[:-1] * 1e-3)  # Lloyd objective is (almost) monotone
    o = Oracle.kmeans_objective(xt, cent)
    assert o <= obj[-1] * 1.001
    if Ref.available():
        _, robj = Ref.kmeans(xt, 40, niter=10, seed=1)
        assert abs(obj[-1] / robj - 1) < 0.05


@pytest.mark.parametrize("d,n,k,niter", [(32, 10000, 40, 6), (2, 65536, 256, 5), (16, 30000, 100, 4), (128, 5000, 512, 3)])
def test_kmeans_device_matches_host_loop(res, d, n, k, niter):
    """The device-resident Lloyd loop (training set uploaded once; assignment, counting sort by cluster and centroid
    update as kernels) against the same loop driven through add()/search() with the update on the host, which is how
    the reference organises it (faiss/Clustering.cpp:255-357): identical centroids and objectives, bit for bit.
    (16, 30000, 100): 30000 > 100 * 256 exercises the subsample; (128, 5000, 512): < 39 points per centroid and
    clustered data leave empty clusters, which exercises the split."""
    if d == 128:
        rs = np.random.RandomState(5)
        xt = (rs.randint(0, 8, size=(n, 1)) * 10 + rs.rand(n, d) * 0.01).astype(np.float32)  # 8 tight blobs, 512 centroids
    else:
        xt, _, _ = synthetic_dataset(d, n, 0, 0, seed=d + k)
    dev_ix = faiss_amd.GpuIndexFlatL2(res, d)
    c_dev = faiss_amd.Clustering(d, k, niter=niter, seed=11)
    c_dev.train(xt, dev_ix)
    assert c_dev.on_device and dev_ix.ntotal == k
    rep = faiss_amd.IndexReplicas(d, threaded=False)  # not a GpuIndexFlat: the generic host loop
    rep.add_replica(faiss_amd.GpuIndexFlatL2(res, d))
    c_host = faiss_amd.Clustering(d, k, niter=niter, seed=11)
    c_host.train(xt, rep)
    assert not c_host.on_device and rep.ntotal == k
    assert np.array_equal(c_dev.obj, c_host.obj)
    assert np.array_equal(c_dev.centroids, c_host.centroids)
    assert np.array_equal(dev_ix.reconstruct_n(0, k), c_dev.centroids)
    cent, obj = faiss_amd.kmeans(res, xt, k, niter=niter, seed=11)
    assert np.array_equal(cent, c_dev.centroids) and np.array_equal(obj, c_dev.obj)


def test_kmeans_device_update_equals_numpy_sums(res):
    """One Lloyd iteration from known centroids: new centroid = float(double sum of the members in index order / count)
    (faiss/Clustering.cpp:307-324), checked against numpy on the assignment the index itself reports."""
    xt, _, _ = synthetic_dataset(24, 6000, 0, 0, seed=9)
    k = 50
    c1 = faiss_amd.Clustering(24, k, niter=1, seed=3)
    ix = faiss_amd.GpuIndexFlatL2(res, 24)
    c1.train(xt, ix)
    c0 = faiss_amd.Clustering(24, k, niter=0, seed=3)  # just the initial points
    ix0 = faiss_amd.GpuIndexFlatL2(res, 24)
    c0.train(xt, ix0)
    _, a = ix0.search(xt, 1)
    a = a.ravel()
    want = c0.centroids.copy()
    for c in range(k):
        m = xt[a == c].astype(np.float64)
        if len(m):
            acc = np.zeros(24, dtype=np.float64)
            for row in m:
                acc += row
            want[c] = (acc / len(m)).astype(np.float32)
    assert (np.bincount(a, minlength=k) > 0).all()  # no split in this configuration
    assert np.array_equal(c1.centroids, want)


def test_ivf_train_from_device_pointers(res):
    """GpuIndexIVF::train with the training set resident on the device (no host round trip) gives the quantizers the
    host-pointer call gives."""
    import torch
    xt, xb, xq = synthetic_dataset(32, 20000, 5000, 50, seed=21)
    a = faiss_amd.GpuIndexIVFPQ(res, 32, 64, 8, 8, METRIC_L2)
    a.train(xt)
    b = faiss_amd.GpuIndexIVFPQ(res, 32, 64, 8, 8, METRIC_L2)
    t = torch.from_numpy(xt).cuda()
    torch.cuda.synchronize()
    b.train_ptr(t.data_ptr(), t.shape[0])
    assert np.array_equal(a.get_centroids(), b.get_centroids())
    assert np.array_equal(a.get_pq_centroids(), b.get_pq_centroids())


# ------------------------------------------------------------------------------- shards / merge
def test_index_shards_equals_single_index(res):
    """faiss/gpu/test/test_multi_gpu.py:31-48: sharded flat must give I == I_ref exactly."""
    xb, xq = integer_dataset(24, 9000, 64, seed=11, hi=6)  # ties on purpose
    single = faiss_amd.GpuIndexFlatL2(res, 24)
    single.add(xb)
    Dr, Ir = single.search(xq, 40)
    sh = faiss_amd.IndexShards(24, threaded=True, successive_ids=True)
    for _ in range(3):
        sh.add_shard(faiss_amd.GpuIndexFlatL2(res, 24))
    sh.add(xb)
    assert sh.ntotal == 9000
    D, I = sh.search(xq, 40)
    check_knn(D, I, Dr, Ir, exact=True, name="shards")


def test_device_merge_matches_host_merge(res):
    import torch
    xb, xq = integer_dataset(16, 3000, 50, seed=2, hi=4)
    parts = [(0, 1000), (1000, 2100), (2100, 3000)]
    k = 33
    aD = np.stack([Oracle.flat_search(METRIC_L2, xb[a:b], xq, k)[0] for a, b in parts])
    aI = np.stack([Oracle.flat_search(METRIC_L2, xb[a:b], xq, k)[1] for a, b in parts])
    base = [a for a, _ in parts]
    Dh, Ih = faiss_amd.merge_knn_results(METRIC_L2, aD, aI, base)
    Do, Io = Oracle.merge_shards(METRIC_L2, aD, aI, base)
    check_knn(Dh, Ih, Do, Io, exact=True, name="host merge")
    check_knn(Dh, Ih, *Oracle.flat_search(METRIC_L2, xb, xq, k), exact=True, name="merge == unsharded")
    dD, dI = torch.from_numpy(aD).cuda(), torch.from_numpy(aI).cuda()
    oD = torch.empty((50, k), dtype=torch.float32, device="cuda")
    oI = torch.empty((50, k), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    faiss_amd.merge_knn_results_device(res, METRIC_L2, 50, k, 3, dD.data_ptr(), dI.data_ptr(), base,
                                       oD.data_ptr(), oI.data_ptr())
    check_knn(oD.cpu().numpy(), oI.cpu().numpy(), Do, Io, exact=True, name="device merge")


def test_device_pointers_in_and_out(res):
    import torch
    _, xb, xq = synthetic_dataset(64, 0, 4000, 100, seed=6)
    idx = faiss_amd.GpuIndexFlatL2(res, 64)
    idx.add_ptr(4000, torch.from_numpy(xb).cuda().data_ptr())
    q = torch.from_numpy(xq).cuda()
    D = torch.empty((100, 10), dtype=torch.float32, device="cuda")
    I = torch.empty((100, 10), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    idx.search_ptr(100, q.data_ptr(), 10, D.data_ptr(), I.data_ptr())
    check_knn(D.cpu().numpy(), I.cpu().numpy(), *Oracle.flat_search(METRIC_L2, xb, xq, 10), exact=True, name="devptr")


# ------------------------------------------------------------------------------- drop-in
@pytest.mark.skipif(not Ref.available(), reason="oracle/_ref not shipped")
def test_dropin_reference_clustering_and_shards_run_on_backend(res):
    """The reference's own callers of the hot path run unchanged on the backend through a
    faiss::Index subclass over the C ABI: faiss::Clustering::train (Clustering.cpp:255-357) and
    faiss::IndexShards (IndexShards.cpp:135-265)."""
    xt, xb, xq = synthetic_dataset(32, 6000, 5000, 40, seed=9)
    amd = faiss_amd.GpuIndexFlatL2(res, 32)
    ad = Ref.adapter(amd)
    cent_gpu, obj_gpu = Ref.kmeans_with_index(xt, 24, ad, niter=8, seed=7)
    cent_cpu, obj_cpu = Ref.kmeans(xt, 24, niter=8, seed=7)
    # same RNG + same assignment results => same trajectory (up to fp noise in near-tie assignments)
    assert abs(obj_gpu / obj_cpu - 1) < 1e-3
    # faiss::IndexShards over two backend indexes == faiss CPU flat
    subs = [Ref.adapter(faiss_amd.GpuIndexFlatL2(res, 32)) for _ in range(2)]
    sh = Ref.shards(32, subs, threaded=True, successive_ids=True)
    sh.add(xb)
    D, I = sh.search(xq, 10)
    ref = Ref.index_factory(32, "Flat")
    ref.add(xb)
    Dr, Ir = ref.search(xq, 10)
    check_knn(D, I, Dr, Ir, rtol=1e-4, name="faiss::IndexShards over backend")


# ------------------------------------------------------------------------------- fused IVF scan
@pytest.mark.parametrize("kind,metric,d,M,nlist,nb,nq,nprobe,k", [
    (1, METRIC_L2, 128, 64, 64, 40000, 1500, 8, 100),   # bench shape (dsub=2), one workgroup per query
    (1, METRIC_INNER_PRODUCT, 128, 64, 64, 40000, 1100, 8, 10),
    (1, METRIC_L2, 64, 16, 32, 30000, 40, 32, 600),      # dsub=4, probes split over workgroups, big k
    (1, METRIC_L2, 96, 12, 16, 5000, 1200, 5, 2048),     # dsub=8, k at the limit, k > candidates for some
    (1, METRIC_L2, 200, 20, 16, 6000, 50, 4, 20),        # d > 128: codebook re-read through L2
    (1, METRIC_L2, 32, 32, 8, 3000, 1030, 3, 5),         # dsub=1
    (0, METRIC_L2, 128, 0, 64, 40000, 1500, 8, 100),
    (0, METRIC_INNER_PRODUCT, 40, 0, 16, 9000, 30, 16, 7),
    (0, METRIC_L2, 72, 0, 8, 20000, 1100, 2, 1000),
])
def test_ivf_fused_scan_matches_unfused_and_oracle(res, kind, metric, d, M, nlist, nb, nq, nprobe, k):
    """The LDS-resident fused scan (table build + scan + reservoir top-k in one launch) returns
    bit-identical distances and labels to the unfused path (all keys in HBM + select kernel)
    and to the CPU oracle restatement."""
    xt, xb, xq = synthetic_dataset(d, 4000, nb, nq, seed=nb + k)
    cent, _ = faiss_amd.kmeans(res, xt, nlist, niter=4, seed=3)
    pq = None
    if kind == 0:
        idx = faiss_amd.GpuIndexIVFFlat(res, d, nlist, metric)
    else:
        idx = faiss_amd.GpuIndexIVFPQ(res, d, nlist, M, 8, metric)
        pq = (np.random.RandomState(7).rand(M, 256, d // M).astype("float32") - 0.5) * 0.4
        idx.copy_pq_centroids(pq)
    idx.copy_centroids(cent)
    idx.add(xb)
    idx.nprobe = nprobe
    D, I = idx.search(xq, k)
    idx.set_use_fused_scan(False)
    D0, I0 = idx.search(xq, k)
    assert np.array_equal(D, D0) and np.array_equal(I, I0)
    sel = np.r_[0:min(nq, 40)]
    sizes, codes, ids, _ = Oracle.build_ivf_lists(kind, metric, cent, xb, pq=pq)
    Do, Io, _, _ = Oracle.ivf_search(kind, metric, cent, sizes, codes, ids, xq[sel], nprobe, k, M=M, pq=pq)
    check_knn(D[sel], I[sel], Do, Io, exact=True, name="fused vs oracle")


# ------------------------------------------------------------------------------- fp16 filter + exact re-rank
@pytest.mark.parametrize("metric", [METRIC_L2, METRIC_INNER_PRODUCT])
@pytest.mark.parametrize("d,scale", [(128, 1.0), (96, 37.0), (200, 1e-3), (32, 300.0)])
def test_filter_error_bound_holds(res, metric, d, scale):
    """The per-query bound e_q used for the candidate band really bounds |approximate - exact| score
    (fp16 rounding of both operands + fp32 accumulation), with margin, on data of several scales."""
    _, xb, xq = synthetic_dataset(d, 0, 3000, 70, seed=d)
    rs = np.random.RandomState(d)
    xb = (xb * scale * (0.2 + rs.rand(len(xb), 1))).astype("float32")
    xq = (xq * scale).astype("float32")
    idx = faiss_amd.GpuIndexFlat(res, d, metric)
    idx.add(xb)
    sc, eb = idx.filter_scores(xq)
    ip = xq.astype("float64") @ xb.astype("float64").T
    exact = ip - (0.5 * (xb.astype("float64") ** 2).sum(1))[None, :] if metric == METRIC_L2 else ip
    err = np.abs(sc - exact).max(axis=1)
    assert (err <= eb).all(), (err / eb).max()
    assert (err > 0).any() and (err / eb).max() < 0.8  # fp16 rounding is visible, the bound has slack


FILTER_SHAPES = [(128, 30000, 300, 10), (128, 30000, 300, 100), (64, 20000, 513, 1), (100, 25000, 257, 50),
                 (32, 18000, 100, 1000), (384, 17000, 30, 20), (200, 40000, 1100, 128),
                 (128, 50000, 2500, 100), (120, 33333, 3000, 7)]  # >= 2048 queries, d <= 128: 8-wave geometry


@pytest.mark.parametrize("metric", [METRIC_L2, METRIC_INNER_PRODUCT])
@pytest.mark.parametrize("d,nb,nq,k", FILTER_SHAPES)
def test_flat_filter_path_bit_exact(res, metric, d, nb, nq, k):
    """fp16 MFMA filter + exact fp32 re-rank returns the very same bits as the fp32 MFMA scan and
    the CPU oracle restatement."""
    _, xb, xq = synthetic_dataset(d, 0, nb, nq, seed=nb + k)
    idx = faiss_amd.GpuIndexFlat(res, d, metric)
    idx.add(xb)
    D, I = idx.search(xq, k)
    used, novf = idx.filter_stats()
    assert used and novf == 0
    idx.set_use_filter_kernel(False)
    D0, I0 = idx.search(xq, k)
    assert not idx.filter_stats()[0]
    assert np.array_equal(D, D0) and np.array_equal(I, I0)
    sel = np.r_[0:min(nq, 60)]
    Do, Io = Oracle.flat_search(metric, xb, xq[sel], k)
    check_knn(D[sel], I[sel], Do, Io, exact=True, name="filter vs oracle")


def test_flat_filter_overflow_falls_back_to_exact(res):
    """Adversarial data: 20 distinct vectors repeated 5000 times (every neighbour tied 5000-fold, so the
    error band of a query holds thousands of rows) and integer data full of ties.  Queries whose band
    does not fit are flagged and re-run through the exact scan; the (distance, id) tie order must survive."""
    _, base, xq = synthetic_dataset(64, 0, 20, 40, seed=9)
    xb = np.tile(base, (5000, 1))
    idx = faiss_amd.GpuIndexFlatL2(res, 64)
    idx.add(xb)
    for k in (10, 500):
        D, I = idx.search(xq, k)
        used, novf = idx.filter_stats()
        assert used and novf > 0
        check_knn(D, I, *Oracle.flat_search(METRIC_L2, xb, xq, k), exact=True, name="dup rows k=%d" % k)
    # moderately tied data stays inside the band machinery (no fallback needed)
    xb = np.tile(synthetic_dataset(64, 0, 2000, 0, seed=5)[1], (10, 1))
    idx = faiss_amd.GpuIndexFlatL2(res, 64)
    idx.add(xb)
    D, I = idx.search(xq, 25)
    assert idx.filter_stats()[0]
    check_knn(D, I, *Oracle.flat_search(METRIC_L2, xb, xq, 25), exact=True, name="10-fold ties")
    z, xb, xq = load_flat_case("flat_l2_int_ties")
    idx = faiss_amd.GpuIndexFlatL2(res, xb.shape[1])
    idx.set_use_filter_kernel(True, 0)
    idx.add(xb)
    for k in z["ks"]:
        D, I = idx.search(xq, int(k))
        assert idx.filter_stats()[0]
        check_knn(D, I, z["D_%d" % k], z["I_%d" % k], exact=True, name="ties k=%d (filter)" % k)


def test_flat_filter_range_and_nan_handling(res):
    """Values outside the fp16 range: a query is flagged and served by the exact scan; a database that
    leaves the range (or holds a NaN) disables the filter for the index.  NaN queries behave as on the fp32 path."""
    _, xb, xq = synthetic_dataset(48, 0, 20000, 12, seed=4)
    xq = xq.copy()
    xq[3] *= 1e6      # beyond 65504 after conversion
    xq[5, 7] = np.nan  # admits nothing
    idx = faiss_amd.GpuIndexFlatL2(res, 48)
    idx.add(xb)
    D, I = idx.search(xq, 8)
    used, novf = idx.filter_stats()
    assert used and novf >= 1
    assert (I[5] == -1).all() and (I[3] >= 0).all()
    idx.set_use_filter_kernel(False)
    D0, I0 = idx.search(xq, 8)
    assert np.array_equal(D, D0) and np.array_equal(I, I0)
    big = faiss_amd.GpuIndexFlatL2(res, 48)
    xb2 = xb.copy()
    xb2[77] *= 1e5
    big.add(xb2)
    D, I = big.search(xq[:3], 8)
    assert not big.filter_stats()[0]
    check_knn(D, I, *Oracle.flat_search(METRIC_L2, xb2, xq[:3], 8), exact=True, name="out of range db")


# ------------------------------------------------------------------------------- rest of the Index surface
@pytest.mark.parametrize("metric", [METRIC_L2, METRIC_INNER_PRODUCT])
@pytest.mark.parametrize("kind", ["ivfflat", "ivfpq"])
def test_search_preassigned_equals_search(res, kind, metric):
    """faiss/gpu/test/test_gpu_index.py:124-194: search(x) and search_preassigned(x, quantizer.search(x)) return the
    same bits; probes knocked out with -1 (or out-of-range ids) behave like a shorter probe list."""
    d, nlist, nb, nq, k, nprobe = 64, 64, 20000, 300, 10, 8
    xt, xb, xq = synthetic_dataset(d, 6000, nb, nq, seed=77)
    if kind == "ivfflat":
        idx = faiss_amd.GpuIndexIVFFlat(res, d, nlist, metric)
    else:
        idx = faiss_amd.GpuIndexIVFPQ(res, d, nlist, 8, 8, metric)
    idx.train(xt)
    idx.add(xb)
    idx.nprobe = nprobe
    D, I = idx.search(xq, k)
    Dq, Iq = idx.quantizer_search(xq, nprobe)
    D2, I2 = idx.search_preassigned(xq, k, Iq, Dq)
    assert np.array_equal(I, I2) and np.array_equal(D, D2)
    for fused in (True, False):
        idx.set_use_fused_scan(fused)
        Iq4 = Iq.copy()
        Iq4[:, 4:] = -1
        Iq4[::3, 5] = nlist + 7  # not a list: ignored
        D3, I3 = idx.search_preassigned(xq, k, Iq4, Dq)
        idx.nprobe = 4
        D4, I4 = idx.search(xq, k)
        idx.nprobe = nprobe
        assert np.array_equal(I3, I4) and np.array_equal(D3, D4)
    # all probes invalid: no result
    Dn, In = idx.search_preassigned(xq[:5], k, np.full((5, nprobe), -1, dtype=np.int64), Dq[:5])
    assert (In == -1).all()
    with pytest.raises(ValueError):
        idx.search_preassigned(xq, k, Iq[:, :4], Dq[:, :4])


def test_flat_compute_residual_and_reconstruct_batch(res):
    """GpuIndexFlat::compute_residual_n / reconstruct_batch (faiss/gpu/GpuIndexFlat.cu:294-361): one fp32
    subtraction per element (bit-exact), key -1 -> NaN row (impl/VectorResidual.cu:35-47)."""
    d, nb, n = 40, 3000, 257
    _, xb, xq = synthetic_dataset(d, 0, nb, n, seed=9)
    idx = faiss_amd.GpuIndexFlatL2(res, d)
    idx.add(xb)
    rs = np.random.RandomState(3)
    keys = rs.randint(0, nb, size=n).astype(np.int64)
    keys[5] = -1
    keys[100] = -1
    good = keys >= 0
    R = idx.compute_residual_n(xq, keys)
    assert np.array_equal(R[good], xq[good] - xb[keys[good]])
    assert np.isnan(R[~good]).all()
    B = idx.reconstruct_batch(keys)
    assert np.array_equal(B[good], xb[keys[good]]) and np.isnan(B[~good]).all()
    r1 = idx.compute_residual(xq[7], int(keys[7]))
    assert np.array_equal(r1, xq[7] - xb[keys[7]])
    assert idx.compute_residual_n(xq[:0], keys[:0]).shape == (0, d)


@pytest.mark.parametrize("metric", [METRIC_L2, METRIC_INNER_PRODUCT])
@pytest.mark.parametrize("d,nb,nq,k", [(128, 30000, 300, 10), (33, 2000, 17, 5), (64, 100, 4, 200)])
def test_bfknn_equals_flat_index_and_oracle(res, metric, d, nb, nq, k):
    """faiss::gpu::bfKnn on raw arrays (faiss/gpu/GpuDistance.h:32-152; test: faiss/gpu/test/test_gpu_basics.py
    bfKnn cases): same answer as a flat index holding the vectors, and as the oracle."""
    _, xb, xq = synthetic_dataset(d, 0, nb, nq, seed=d + k)
    D, I = faiss_amd.knn_gpu(res, xq, xb, k, metric=metric)
    idx = faiss_amd.GpuIndexFlat(res, d, metric)
    idx.add(xb)
    D2, I2 = idx.search(xq, k)
    assert np.array_equal(I, I2) and np.array_equal(D, D2)
    Do, Io = Oracle.flat_search(metric, xb, xq, k)
    check_knn(D, I, Do, Io, exact=True, name="bfKnn")
    assert xb.flags.c_contiguous  # inputs untouched


def test_index_replicas_equals_single_index(res):
    """faiss/tests/test_threaded_index.cpp:165-214 (replicas): add goes to every replica, queries are dealt out in
    ceil(n / count) blocks, results equal the single index bit for bit."""
    xb, xq = integer_dataset(24, 9000, 67, seed=12, hi=6)
    single = faiss_amd.GpuIndexFlatL2(res, 24)
    single.add(xb)
    Dr, Ir = single.search(xq, 40)
    rep = faiss_amd.IndexReplicas(24, threaded=True)
    for _ in range(3):
        rep.add_replica(faiss_amd.GpuIndexFlatL2(faiss_amd.StandardGpuResources(0), 24))
    rep.add(xb)
    assert rep.ntotal == 9000
    D, I = rep.search(xq, 40)
    assert np.array_equal(I, Ir) and np.array_equal(D, Dr)
    D1, I1 = rep.search(xq[:2], 40)  # fewer queries than replicas
    assert np.array_equal(I1, Ir[:2])
    assert np.array_equal(rep.reconstruct(17), xb[17])
    rep.reset()
    assert rep.ntotal == 0


def test_partial_query_groups_in_the_8_wave_geometry(res):
    """query counts that leave whole wavefronts of the last workgroup without a query (they skip the MFMAs)"""
    _, xb, xq = synthetic_dataset(128, 0, 40000, 2600, seed=21)
    idx = faiss_amd.GpuIndexFlatL2(res, 128)
    idx.add(xb)
    idx.set_use_filter_kernel(False)
    Dr, Ir = idx.search(xq, 20)
    idx.set_use_filter_kernel(True)
    for n in (2049, 2176, 2600):
        D, I = idx.search(xq[:n], 20)
        used, novf = idx.filter_stats()
        assert used and novf == 0
        assert np.array_equal(I, Ir[:n]) and np.array_equal(D, Dr[:n])


@pytest.mark.parametrize("kind", ["ivfflat", "ivfpq"])
def test_ivf_nan_queries_and_nan_adds(res, kind):
    """faiss/gpu/test/TestGpuIndexIVFFlat.cpp:566-605 (QueryNaN: every result is -1 / FLT_MAX) and :633-675 (AddNaN:
    NaN vectors are not stored, the valid one among them is, nothing crashes); same pair in TestGpuIndexIVFPQ.cpp."""
    d, nlist, k = 32, 16, 5
    xt, xb, xq = synthetic_dataset(d, 3000, 4000, 10, seed=31)
    if kind == "ivfflat":
        idx = faiss_amd.GpuIndexIVFFlat(res, d, nlist, METRIC_L2)
    else:
        idx = faiss_amd.GpuIndexIVFPQ(res, d, nlist, 4, 8, METRIC_L2)
    idx.train(xt)
    idx.nprobe = 4
    # AddNaN on the empty index: one valid vector (not the first) among NaN rows
    nans = np.full((10, d), np.nan, dtype=np.float32)
    nans[1] = xb[123]
    idx.add(nans)
    assert sum(idx.get_list_size(l) for l in range(nlist)) == 1
    # ntotal counts the vectors add() was given, stored or not (faiss/gpu/GpuIndexIVF.cu:293-298) ...
    assert idx.ntotal == 10 and idx.stored_vectors == 1
    idx.nprobe = nlist
    D, I = idx.search(xq, k)
    assert (I[:, 0] == 1).all() and (I[:, 1:] == -1).all()
    # more data, then QueryNaN
    idx.add(xb)
    # ... so the ids generated for the next add() start behind them and never collide (GpuIndex.cu:137-144)
    assert idx.ntotal == 10 + len(xb) and idx.stored_vectors == 1 + len(xb)
    all_ids = np.concatenate([idx.get_list_ids(l) for l in range(nlist)])
    assert len(np.unique(all_ids)) == len(all_ids) == 1 + len(xb)
    assert set(all_ids.tolist()) == {1} | set(range(10, 10 + len(xb)))
    idx.nprobe = 4
    qn = np.full((10, d), np.nan, dtype=np.float32)
    D, I = idx.search(qn, k)
    assert (I == -1).all() and (D == FMAX).all()
    # a NaN query between valid ones leaves its neighbours' results untouched
    mix = xq.copy()
    mix[3] = np.nan
    Dm, Im = idx.search(mix, k)
    Dv, Iv = idx.search(xq, k)
    keep = np.arange(10) != 3
    assert np.array_equal(Im[keep], Iv[keep]) and np.array_equal(Dm[keep], Dv[keep]) and (Im[3] == -1).all()


# ------------------------------------------------------------------------------- round 2: storage, add path, parameters
@pytest.mark.parametrize("metric", [METRIC_L2, METRIC_INNER_PRODUCT])
@pytest.mark.parametrize("kind", ["ivfflat", "ivfpq"])
def test_ivf_trained_but_empty_returns_padding(res, kind, metric):
    """A trained index without vectors (a fresh index, an IndexShards shard that received no rows) answers with
    -1 / +-FLT_MAX like the reference (faiss/utils/Heap.h:427-457), on every path."""
    d, nlist = 32, 8
    xt, _, xq = synthetic_dataset(d, 3000, 0, 9, seed=5)
    idx = (faiss_amd.GpuIndexIVFFlat(res, d, nlist, metric) if kind == "ivfflat"
           else faiss_amd.GpuIndexIVFPQ(res, d, nlist, 4, 8, metric))
    idx.train(xt)
    assert idx.is_trained and idx.ntotal == 0
    idx.nprobe = 3
    pad = FMAX if metric == METRIC_L2 else -FMAX
    for fused in (True, False):
        idx.set_use_fused_scan(fused)
        D, I = idx.search(xq, 4)
        assert (I == -1).all() and (D == pad).all()
    # one shard of three stays empty: the sharded search still answers
    if kind == "ivfpq" and metric == METRIC_L2:
        sh = faiss_amd.IndexShards(d, threaded=True, successive_ids=False)
        subs = []
        for _ in range(3):
            s = faiss_amd.GpuIndexIVFPQ(res, d, nlist, 4, 8, metric)
            s.copy_centroids(idx.get_centroids())
            s.copy_pq_centroids(idx.get_pq_centroids())
            s.nprobe = nlist
            subs.append(s)
            sh.add_shard(s)
        sh.add_with_ids(synthetic_dataset(d, 0, 2, 0, seed=6)[1], np.array([5, 9]))  # rows go to shards 0 and 1... or 1 and 2
        D, I = sh.search(xq, 3)
        assert (np.sort(I[:, :2], axis=1) == np.array([5, 9])).all() and (I[:, 2] == -1).all()


def test_ivfpq_is_trained_needs_both_quantizers(res):
    d, nlist, M = 32, 8, 4
    xt, xb, _ = synthetic_dataset(d, 3000, 100, 0, seed=5)
    cent, _ = faiss_amd.kmeans(res, xt, nlist, niter=3, seed=1)
    idx = faiss_amd.GpuIndexIVFPQ(res, d, nlist, M, 8, METRIC_L2)
    idx.copy_centroids(cent)
    assert not idx.is_trained  # no codebook yet
    with pytest.raises(faiss_amd.FaissAmdError):
        idx.add(xb)
    with pytest.raises(faiss_amd.FaissAmdError):
        idx.copy_lists(np.zeros(nlist, np.uint32), np.zeros((0, M), np.uint8), np.zeros(0, np.int64))
    assert idx.ntotal == 0
    idx.train(xt)  # trains only what is missing: the centroids stay
    assert idx.is_trained and np.array_equal(idx.get_centroids(), cent)
    idx.add(xb)
    assert idx.ntotal == 100


@pytest.mark.parametrize("name", ["ivfflat_l2", "ivfpq_l2"])
def test_ivf_many_incremental_adds_equal_one_add(res, name):
    """20 add_with_ids calls of uneven size leave byte-identical lists (sizes, ids, codes, insertion order) to one big
    add and to the reference's own lists (testIVFEquality, faiss/gpu/test/TestUtils.h:111-142;
    reference add path faiss/gpu/impl/IVFBase.cu:595-905), and identical search results."""
    c = load_ivf_case(name)
    z = c["z"]
    n = len(c["xb"])
    nlist = z["centroids"].shape[0]
    one = _make_ivf(res, c)
    one.add_with_ids(c["xb"], c["ids"])
    many = _make_ivf(res, c)
    cuts = np.unique(np.r_[0, np.sort(np.random.RandomState(4).randint(1, n, size=19)), n])
    for a, b in zip(cuts[:-1], cuts[1:]):
        many.add_with_ids(c["xb"][a:b], c["ids"][a:b])
    assert many.ntotal == one.ntotal == n
    for l in range(nlist):
        assert np.array_equal(many.get_list_ids(l), one.get_list_ids(l))
        assert np.array_equal(many.get_list_codes(l), one.get_list_codes(l))
    assert np.array_equal(np.array([many.get_list_size(l) for l in range(nlist)], dtype=np.uint32), z["list_sizes"])
    assert np.array_equal(np.concatenate([many.get_list_ids(l) for l in range(nlist)]), z["list_ids"])
    used, holes, alloc = many.arena_stats()
    assert holes < used <= alloc and used < 8 * n + 64 * nlist  # geometric growth, bounded waste
    for idx in (one, many):
        idx.nprobe = c["nprobe"]
    D1, I1 = one.search(c["xq"], c["k"])
    D2, I2 = many.search(c["xq"], c["k"])
    assert np.array_equal(I1, I2) and np.array_equal(D1, D2)  # distances do not depend on where a vector is stored
    check_knn(D2, I2, z["D"], z["I"], rtol=1e-4, name=name + " incremental vs golden")


def test_ivf_search_parameters_nprobe_override(res):
    """SearchParametersIVF.nprobe (faiss/IndexIVF.h:70-80, GpuIndexIVF.cu:358-381) overrides index.nprobe for one call"""
    c = load_ivf_case("ivfpq_l2")
    idx = _make_ivf(res, c)
    idx.copy_lists(c["z"]["list_sizes"], c["codes"], c["z"]["list_ids"])
    idx.nprobe = 1
    D1, I1 = idx.search(c["xq"], c["k"])
    Dp, Ip = idx.search(c["xq"], c["k"], params=faiss_amd.SearchParametersIVF(nprobe=c["nprobe"]))
    assert idx.nprobe == 1
    idx.nprobe = c["nprobe"]
    D2, I2 = idx.search(c["xq"], c["k"])
    assert np.array_equal(Ip, I2) and np.array_equal(Dp, D2) and not np.array_equal(I1, I2)
    with pytest.raises(faiss_amd.FaissAmdError):
        idx.search(c["xq"], c["k"], params=faiss_amd.SearchParametersIVF(nprobe=5000))


def test_ivfflat_reconstruct_n(res):
    """GpuIndexIVFFlat::reconstruct_n (faiss/gpu/GpuIndexIVFFlat.cu:370-390): rows of a contiguous id range"""
    c = load_ivf_case("ivfflat_l2")
    idx = _make_ivf(res, c)
    idx.add(c["xb"])  # ids 0..n-1
    assert np.array_equal(idx.reconstruct_n(100, 300), c["xb"][100:400])
    assert np.array_equal(idx.reconstruct(7), c["xb"][7])


@pytest.mark.parametrize("M,d", [(4, 32), (8, 64), (12, 96), (20, 200), (32, 32), (48, 96), (64, 128), (96, 192)])
def test_ivfpq_code_layout_round_trip(res, M, d):
    """copy_lists -> rotated 64-row block layout -> get_list_codes is the identity for every chunk width (M % 16 == 0:
    16-byte chunks, else 4), on lists that are empty, shorter than a block, and several blocks long; the scan over that
    layout matches the oracle bit for bit."""
    nlist, nq, k = 5, 30, 10
    rs = np.random.RandomState(M)
    xt, _, xq = synthetic_dataset(d, 2000, 0, nq, seed=M)
    cent, _ = faiss_amd.kmeans(res, xt, nlist, niter=3, seed=2)
    pq = (rs.rand(M, 256, d // M).astype("float32") - 0.5) * 0.5
    sizes = np.array([0, 1, 63, 64, 200], dtype=np.uint32)
    n = int(sizes.sum())
    codes = rs.randint(0, 256, size=(n, M)).astype(np.uint8)
    ids = rs.permutation(10 * n)[:n].astype(np.int64)
    for metric in (METRIC_L2, METRIC_INNER_PRODUCT):
        idx = faiss_amd.GpuIndexIVFPQ(res, d, nlist, M, 8, metric)
        idx.copy_centroids(cent)
        idx.copy_pq_centroids(pq)
        idx.copy_lists(sizes, codes, ids)
        off = 0
        for l in range(nlist):
            assert np.array_equal(idx.get_list_codes(l), codes[off:off + sizes[l]])
            assert np.array_equal(idx.get_list_ids(l), ids[off:off + sizes[l]])
            off += int(sizes[l])
        idx.nprobe = nlist
        Do, Io, _, _ = Oracle.ivf_search(1, metric, cent, sizes, codes, ids, xq, nlist, k, M=M, pq=pq)
        for fused in (True, False):
            idx.set_use_fused_scan(fused)
            D, I = idx.search(xq, k)
            check_knn(D, I, Do, Io, exact=True, name="layout M=%d fused=%s" % (M, fused))


# ------------------------------------------------------------------------------- selection primitives in isolation
@pytest.mark.parametrize("which", [0, 1, 2])
@pytest.mark.parametrize("k", [1, 64, 100, 1024, 2048])
def test_select_primitives_standalone(res, which, k):
    """faiss/gpu/test/TestGpuSelect.cu:23-198 (testForSize): for every row the k selected VALUES are exactly the k best of
    a CPU sort, every returned index gathers to the returned value, indices are unique; here additionally ties go to the
    lower column (the backend's total order), so indices are compared exactly too.  which = 0: select_k_kernel (the
    BlockSelect role), 1: workgroup LDS reservoir (wg_select.h), 2: wavefront select (wave_select.h, unordered winners).
    Inputs: random, heavily tied (16 distinct values), all-equal, and fewer columns than k."""
    rs = np.random.RandomState(k + which)
    cases = [rs.rand(7, 5000).astype(np.float32), rs.randint(0, 16, size=(5, 4097)).astype(np.float32),
             np.full((3, 3000), 2.5, dtype=np.float32), rs.rand(4, max(1, k // 2)).astype(np.float32),
             rs.rand(2, k).astype(np.float32)]
    for metric in (METRIC_L2, METRIC_INNER_PRODUCT):
        for vals in cases:
            rows, cols = vals.shape
            D, I = faiss_amd.test_select(res, which, vals, k, metric)
            kk = min(k, cols)
            order = np.lexsort((np.broadcast_to(np.arange(cols), vals.shape), vals if metric == METRIC_L2 else -vals), axis=1)
            Iref = order[:, :kk]
            Dref = np.take_along_axis(vals, Iref, axis=1)
            if which == 2:  # unordered winners: bring them into (value, index) order first
                o = np.lexsort((I, D if metric == METRIC_L2 else -D), axis=1)
                pad = I == -1
                o = np.lexsort((I, np.where(pad, np.inf, D if metric == METRIC_L2 else -D)), axis=1)
                D, I = np.take_along_axis(D, o, axis=1), np.take_along_axis(I, o, axis=1)
            assert np.array_equal(D[:, :kk], Dref), (which, k, metric, vals.shape)
            assert np.array_equal(I[:, :kk], Iref), (which, k, metric, vals.shape)
            assert np.array_equal(np.take_along_axis(vals, I[:, :kk], axis=1), D[:, :kk])  # indices gather to the values
            assert all(len(set(r)) == kk for r in I[:, :kk])
            pad = FMAX if metric == METRIC_L2 else -FMAX
            assert (I[:, kk:] == -1).all() and (D[:, kk:] == pad).all()


def test_flat_l2_clamp_of_negative_distances(res):
    """runSumAlongRows(zeroClamp = true), faiss/gpu/impl/BroadcastSum.cu:201-354 / faiss/utils/distances.cpp:480-495: a
    query equal to a database row can come out of |x|^2 + |y|^2 - 2<x,y> slightly negative; the result is clamped to 0.
    Large-norm rows make the cancellation error visible; every path (fp32 scan, fp16 filter + re-rank, scalar
    cross-check) returns exactly what the oracle returns, never a negative distance."""
    rs = np.random.RandomState(3)
    xb = (rs.rand(20000, 96).astype(np.float32) + 0.5) * 37.0
    xq = xb[rs.choice(len(xb), 200, replace=False)].copy()
    Do, Io = Oracle.flat_search(METRIC_L2, xb, xq, 5)
    assert (Do >= 0).all()
    # the clamp really fires on this data: the unclamped expression fmaf(-2, <x,x>, |x|^2 + |x|^2) of the restatement is
    # negative for some (query, own row) pairs -- they must come out as exactly 0
    import ctypes
    lib = Oracle.lib()
    lib.orc_norm_l2sqr.restype = ctypes.c_float
    neg = 0
    for i in range(len(xq)):
        x = np.ascontiguousarray(xq[i])
        xp = ctypes.c_void_p(x.ctypes.data)
        ip = lib.orc_ip_chain(xp, xp, ctypes.c_int(96))
        xn = lib.orc_norm_l2sqr(xp, ctypes.c_int(96))
        raw = np.float32(np.float64(np.float32(xn + xn)) - 2.0 * np.float64(ip))
        if raw < 0:
            neg += 1
            assert Do[i, 0] == 0.0 and Io[i, 0] >= 0
    assert neg >= 10, neg
    for simple, filt in ((False, True), (False, False), (True, False)):
        idx = faiss_amd.GpuIndexFlatL2(res, 96)
        idx.set_use_simple_kernel(simple)
        idx.set_use_filter_kernel(filt)
        idx.add(xb)
        D, I = idx.search(xq, 5)
        check_knn(D, I, Do, Io, exact=True, name="clamp simple=%s filter=%s" % (simple, filt))
        assert (D >= 0).all()


# ------------------------------------------------------------------------------- config structs, fp16 storage
@pytest.mark.parametrize("metric", [METRIC_L2, METRIC_INNER_PRODUCT])
@pytest.mark.parametrize("d,nb,nq,k", [(128, 30000, 300, 10), (96, 20000, 257, 100), (40, 3000, 50, 7), (64, 50, 5, 8),
                                        (128, 40000, 64, 2048)])
def test_flat_use_float16_storage(res, metric, d, nb, nq, k):
    """GpuIndexFlatConfig::useFloat16 (faiss/gpu/GpuIndexFlat.h:24-40, impl/FlatIndex.cu:39-135): vectors stored as fp16
    only, queries converted to fp16, distances = fp32 arithmetic on those fp16 values.  Parity contract: bit-identical to
    the fp32 path (and to the oracle) run on the fp16-rounded inputs -- on the filter path, the exact scan (small
    database, k > 1024) and the scalar cross-check; half the resident bytes; reconstruct returns the stored values."""
    _, xb, xq = synthetic_dataset(d, 0, nb, nq, seed=nb + k)
    xb16, xq16 = xb.astype(np.float16).astype(np.float32), xq.astype(np.float16).astype(np.float32)
    idx = faiss_amd.GpuIndexFlat(res, d, metric, faiss_amd.GpuIndexFlatConfig(useFloat16=True))
    idx.add(xb[: nb // 3])
    idx.add(xb[nb // 3:])
    assert idx.ntotal == nb
    D, I = idx.search(xq, k)
    Do, Io = Oracle.flat_search(metric, xb16, xq16[:64], k)
    check_knn(D[:64], I[:64], Do, Io, exact=True, name="fp16 storage vs oracle on rounded inputs")
    ref = faiss_amd.GpuIndexFlat(res, d, metric)
    ref.add(xb16)
    D32, I32 = ref.search(xq16, k)
    assert np.array_equal(I, I32) and np.array_equal(D, D32)
    if nb >= 16384 and k <= 1024:
        assert idx.filter_stats() == (True, 0)
    idx.set_use_filter_kernel(False)
    D2, I2 = idx.search(xq, k)
    assert np.array_equal(I, I2) and np.array_equal(D, D2)
    # the fp32 rows are gone (the fp16 rows are what the filter kernel reads anyway): 264 instead of 776 bytes per
    # vector at d = 128
    assert idx.resident_bytes == ref.resident_bytes - nb * ((d + 7) // 8 * 8) * 4
    if d >= 96:
        assert idx.resident_bytes < 0.45 * ref.resident_bytes
    assert np.array_equal(idx.reconstruct_n(5, 40), xb16[5:45])
    keys = np.array([0, nb - 1, 7, -1], dtype=np.int64)
    rb = idx.reconstruct_batch(keys)
    assert np.array_equal(rb[:3], xb16[keys[:3]]) and np.isnan(rb[3]).all()
    assert np.array_equal(idx.compute_residual_n(xq[:3], keys[:3]), xq[:3] - xb16[keys[:3]])
    # and against the original fp32 data: the fp16 rounding of the inputs is all that separates the two
    Df, If = Oracle.flat_search(metric, xb, xq[:64], min(k, 10))
    assert (I[:64, 0] == If[:, 0]).mean() > 0.9
    assert np.allclose(D[:64, :1], Df[:, :1], rtol=5e-3, atol=5e-3 * float(np.abs(Df).max()))


def test_config_structs_on_the_constructors(res):
    """GpuIndexConfig / GpuIndexIVFConfig / GpuIndexIVFPQConfig fields (faiss/gpu/GpuIndex.h:30-47, GpuIndexIVF.h:24-38,
    GpuIndexIVFPQ.h:25-49): accepted where they are meaningful or harmless here, refused loudly where not."""
    d = 32
    faiss_amd.GpuIndexFlat(res, d, METRIC_L2, faiss_amd.GpuIndexFlatConfig(device=0, storeTransposed=True))
    with pytest.raises(faiss_amd.FaissAmdError):
        faiss_amd.GpuIndexFlat(res, d, METRIC_L2, faiss_amd.GpuIndexFlatConfig(device=3))
    with pytest.raises(faiss_amd.FaissAmdError):
        faiss_amd.GpuIndexFlat(res, d, METRIC_L2, faiss_amd.GpuIndexFlatConfig(memorySpace=1))
    faiss_amd.GpuIndexIVFFlat(res, d, 8, METRIC_L2, faiss_amd.GpuIndexIVFConfig(indicesOptions=2))  # INDICES_32_BIT
    # round 6: INDICES_CPU / INDICES_IVF and an fp16 coarse quantizer are served (tests/test_gpu_round6.py); what is left to refuse
    # is a value outside the enum and Unified memory
    for ok in (dict(indicesOptions=0), dict(indicesOptions=1), dict(flat_useFloat16=True)):
        faiss_amd.GpuIndexIVFFlat(res, d, 8, METRIC_L2, faiss_amd.GpuIndexIVFConfig(**ok))
    for bad in (dict(indicesOptions=4), dict(indicesOptions=-1), dict(memorySpace=1)):
        with pytest.raises(faiss_amd.FaissAmdError):
            faiss_amd.GpuIndexIVFFlat(res, d, 8, METRIC_L2, faiss_amd.GpuIndexIVFConfig(**bad))
    xt, xb, xq = synthetic_dataset(d, 2000, 3000, 20, seed=3)
    a = faiss_amd.GpuIndexIVFPQ(res, d, 8, 4, 8, METRIC_L2,
                                faiss_amd.GpuIndexIVFPQConfig(useFloat16LookupTables=True, usePrecomputedTables=True))
    b = faiss_amd.GpuIndexIVFPQ(res, d, 8, 4, 8, METRIC_L2)
    for i in (a, b):
        i.train(xt)
        i.add(xb)
        i.nprobe = 4
    Da, Ia = a.search(xq, 5)
    Db, Ib = b.search(xq, 5)
    assert np.array_equal(Ia, Ib) and np.array_equal(Da, Db)  # the options do not change the arithmetic here


# ------------------------------------------------------------------------------- host-query pipeline, caller streams
@pytest.mark.parametrize("kind", ["flat", "ivfflat", "ivfpq"])
def test_paged_host_search_equals_direct(kind):
    """GpuIndex::searchFromCpuPaged_ (faiss/gpu/GpuIndex.cu:554-774): host-resident batches above the paging threshold go
    page by page through pinned double buffers with the copies on a second stream -- same bits as the direct path, for
    uneven last pages, one page, and with preassigned coarse quantization."""
    res = faiss_amd.StandardGpuResources(0)
    d, k = 48, 12
    xt, xb, xq = synthetic_dataset(d, 3000, 30000, 5003, seed=13)
    if kind == "flat":
        idx = faiss_amd.GpuIndexFlatL2(res, d)
    elif kind == "ivfflat":
        idx = faiss_amd.GpuIndexIVFFlat(res, d, 32, METRIC_L2)
    else:
        idx = faiss_amd.GpuIndexIVFPQ(res, d, 32, 8, 8, METRIC_L2)
    idx.train(xt)
    idx.add(xb)
    if kind != "flat":
        idx.nprobe = 6
    D0, I0 = idx.search(xq, k)
    assert res.paged_search_count == 0
    for page in (1024, 5003, 7000):
        res.setPagedSearch(min_bytes=1, page_queries=page)
        D, I = idx.search(xq, k)
        assert np.array_equal(I, I0) and np.array_equal(D, D0), page
    assert res.paged_search_count == 3
    if kind != "flat":
        res.setPagedSearch(min_bytes=1 << 40)
        Dq, Iq = idx.quantizer_search(xq, 6)
        res.setPagedSearch(min_bytes=1, page_queries=999)
        D, I = idx.search_preassigned(xq, k, Iq, Dq)
        assert np.array_equal(I, I0) and np.array_equal(D, D0)
    res.setPagedSearch()  # defaults


def test_set_default_stream_orders_work_on_the_callers_stream():
    """StandardGpuResources::setDefaultStream: inputs produced on a torch stream and results consumed on it need no
    synchronisation once the index works on that stream."""
    import torch
    res = faiss_amd.StandardGpuResources(0)
    _, xb, xq = synthetic_dataset(64, 0, 20000, 500, seed=3)
    idx = faiss_amd.GpuIndexFlatL2(res, 64)
    idx.add(xb)
    D0, I0 = idx.search(xq, 10)
    st = torch.cuda.Stream()
    res.setDefaultStream(st.cuda_stream)
    with torch.cuda.stream(st):
        q = torch.from_numpy(xq).cuda(non_blocking=True) * 1.0  # produced on st
        D = torch.empty((500, 10), dtype=torch.float32, device="cuda")
        I = torch.empty((500, 10), dtype=torch.int64, device="cuda")
        idx.search_ptr(500, q.data_ptr(), 10, D.data_ptr(), I.data_ptr())
        Dh, Ih = D.cpu(), I.cpu()  # consumed on st
    assert np.array_equal(Ih.numpy(), I0) and np.array_equal(Dh.numpy(), D0)
    res.setDefaultStream(None)
    D1, I1 = idx.search(xq, 10)
    assert np.array_equal(I1, I0)


# ------------------------------------------------------------------------------- bfKnn: the full GpuDistanceParams surface
def test_bfknn_dtypes_layouts_indices_and_pairwise(res):
    """faiss::gpu::bfKnn with GpuDistanceParams (faiss/gpu/GpuDistance.h:32-152; tests: faiss/gpu/test/test_gpu_basics.py
    TestKnn / bfKnn cases): float16 inputs, column-major inputs, int32 labels, k = -1 (all pairwise distances)."""
    d, nb, nq, k = 64, 20000, 200, 10
    _, xb, xq = synthetic_dataset(d, 0, nb, nq, seed=17)
    D0, I0 = faiss_amd.knn_gpu(res, xq, xb, k)
    Do, Io = Oracle.flat_search(METRIC_L2, xb, xq, k)
    check_knn(D0, I0, Do, Io, exact=True, name="bfKnn f32 row major")
    # column-major vectors and queries: the same values, the same answer
    D1, I1 = faiss_amd.knn_gpu(res, np.asfortranarray(xq), np.asfortranarray(xb), k)
    assert np.array_equal(I1, I0) and np.array_equal(D1, D0)
    # int32 labels
    I32 = np.empty((nq, k), dtype=np.int32)
    D2, _ = faiss_amd.knn_gpu(res, xq, xb, k, I=I32)
    assert np.array_equal(I32, I0.astype(np.int32)) and np.array_equal(D2, D0)
    # float16 vectors and queries: fp32 arithmetic on those fp16 values (the useFloat16 contract)
    xb16, xq16 = xb.astype(np.float16), xq.astype(np.float16)
    D3, I3 = faiss_amd.knn_gpu(res, xq16, xb16, k)
    check_knn(D3, I3, *Oracle.flat_search(METRIC_L2, xb16.astype(np.float32), xq16.astype(np.float32), k), exact=True,
              name="bfKnn f16")
    # mixed: fp16 vectors, fp32 queries
    D4, I4 = faiss_amd.knn_gpu(res, xq, xb16, k)
    check_knn(D4, I4, *Oracle.flat_search(METRIC_L2, xb16.astype(np.float32), xq, k), exact=True, name="bfKnn mixed")
    # inner product, k = -1: the whole matrix
    G, none = faiss_amd.knn_gpu(res, xq[:40], xb[:3000], -1, metric=METRIC_INNER_PRODUCT)
    assert none is None and np.array_equal(G, Oracle.pairwise(METRIC_INNER_PRODUCT, xb[:3000], xq[:40]))


@pytest.mark.parametrize("vlim,qlim", [(300000, 0), (0, 40000), (700001, 123457)])
def test_bfknn_tiling_equals_untiled(res, vlim, qlim):
    """faiss::gpu::bfKnn_tiling (GpuDistance.cu:430-570): vectors / queries beyond the given device-memory limits are
    processed tile by tile; the merged result equals the untiled search (ids exactly, ties to the lower id)."""
    xb, xq = integer_dataset(32, 9001, 301, seed=21, hi=6)  # many exact ties across the chunk borders
    k = 25
    D0, I0 = faiss_amd.knn_gpu(res, xq, xb, k)
    D, I = faiss_amd.knn_gpu(res, xq, xb, k, vectorsMemoryLimit=vlim, queriesMemoryLimit=qlim)
    assert np.array_equal(I, I0) and np.array_equal(D, D0)
    with pytest.raises(faiss_amd.FaissAmdError):
        faiss_amd.knn_gpu(res, xq, xb, k, vectorsMemoryLimit=16)  # below one vector


# ------------------------------------------------------------------------------- ParameterSpace, InterruptCallback
def test_gpu_parameter_space_and_interrupt(res):
    """GpuParameterSpace::set_index_parameter (faiss/gpu/GpuAutoTune.cpp:81-114): nprobe on IVF indexes, through
    IndexReplicas / IndexShards; InterruptCallback polled between tiles (faiss/gpu/impl/Distance.cu:245,266)."""
    d = 32
    xt, xb, xq = synthetic_dataset(d, 2000, 6000, 30, seed=8)
    ivf = faiss_amd.GpuIndexIVFFlat(res, d, 16, METRIC_L2)
    ivf.train(xt)
    ivf.add(xb)
    ps = faiss_amd.GpuParameterSpace()
    assert ps.initialize(ivf)["nprobe"] == [1, 2, 4, 8]  # (below nlist = 16, like the reference: GpuAutoTune.cpp:57-63)
    ps.set_index_parameters(ivf, "nprobe=8")
    assert ivf.nprobe == 8
    sh = faiss_amd.IndexShards(d, threaded=False, successive_ids=False)
    subs = [faiss_amd.GpuIndexIVFFlat(res, d, 16, METRIC_L2) for _ in range(2)]
    for s in subs:
        s.copy_centroids(ivf.get_centroids())
        sh.add_shard(s)
    ps.set_index_parameter(sh, "nprobe", 4)
    assert [s.nprobe for s in subs] == [4, 4]
    with pytest.raises(faiss_amd.FaissAmdError):
        ps.set_index_parameter(faiss_amd.GpuIndexFlatL2(res, d), "nprobe", 4)
    calls = []
    faiss_amd.set_interrupt_callback(lambda: calls.append(1) or len(calls) > 1)
    try:
        ivf.search(xq, 3)  # first poll passes
        with pytest.raises(faiss_amd.FaissAmdError, match="interrupted"):
            ivf.search(xq, 3)
    finally:
        faiss_amd.set_interrupt_callback(None)
    D, I = ivf.search(xq, 3)
    assert (I >= 0).all()


@pytest.mark.parametrize("metric", [METRIC_L2, METRIC_INNER_PRODUCT])
@pytest.mark.parametrize("d,nb,nq", [(2, 256, 70000), (8, 256, 5000), (30, 300, 1000), (4, 1, 10), (16, 700, 333)])
def test_flat_k1_small_database_assign_kernel(res, metric, d, nb, nq):
    """k = 1 on a database that fits LDS (the k-means assignment of a PQ sub-space, ProductQuantizer::train ->
    Clustering::train -> index.search(k = 1), faiss/Clustering.cpp:351-356): the one-launch assign kernel returns the
    bits of the scan + select pair and of the oracle; ties go to the lower id; NaN queries get -1."""
    _, xb, xq = synthetic_dataset(d, 0, nb, nq, seed=d + nb)
    xq = xq.copy()
    xq[min(7, nq - 1)] = np.nan
    idx = faiss_amd.GpuIndexFlat(res, d, metric)
    idx.add(xb)
    D, I = idx.search(xq, 1)
    Do, Io = Oracle.flat_search(metric, xb, xq[:2000], 1)
    check_knn(D[:2000], I[:2000], Do, Io, exact=True, name="k=1 small db")
    idx.set_use_simple_kernel(True)  # the generic path: every distance as a key + select
    D2, I2 = idx.search(xq, 1)
    assert np.array_equal(I, I2) and np.array_equal(D, D2)
    assert I[min(7, nq - 1), 0] == -1
    xbi, xqi = integer_dataset(4, 200, 500, seed=1, hi=3)  # exact ties everywhere
    idx = faiss_amd.GpuIndexFlat(res, 4, metric)
    idx.add(xbi)
    D, I = idx.search(xqi, 1)
    check_knn(D, I, *Oracle.flat_search(metric, xbi, xqi, 1), exact=True, name="k=1 ties")


# ------------------------------------------------------------------------------- storage corner cases (round 2 arena)
def test_ivf_arena_corner_cases(res):
    """The list arena through its less travelled paths: add after copy_lists (no slack: every touched list relocates),
    compaction once holes pile up, reset + reuse, many lists (nlist > 16384: the rank kernel's global-memory variant),
    an add that touches a single list only.  Lists stay byte-identical to the restatement's, searches to the oracle's."""
    d, nlist, M = 32, 24, 8
    xt, xb, xq = synthetic_dataset(d, 2000, 12000, 25, seed=71)
    cent, _ = faiss_amd.kmeans(res, xt, nlist, niter=3, seed=5)
    pq = (np.random.RandomState(2).rand(M, 256, d // M).astype("float32") - 0.5) * 0.5

    def fresh():
        i = faiss_amd.GpuIndexIVFPQ(res, d, nlist, M, 8, METRIC_L2)
        i.copy_centroids(cent)
        i.copy_pq_centroids(pq)
        i.nprobe = 6
        return i

    def check(idx, rows, ids):
        sizes, codes, lids, _ = Oracle.build_ivf_lists(1, METRIC_L2, cent, rows, ids=ids, pq=pq)
        assert np.array_equal(np.array([idx.get_list_size(l) for l in range(nlist)], dtype=np.uint32), sizes)
        assert np.array_equal(np.concatenate([idx.get_list_ids(l) for l in range(nlist)]), lids)
        assert np.array_equal(np.concatenate([idx.get_list_codes(l) for l in range(nlist)]), codes)
        D, I = idx.search(xq, 10)
        Do, Io, _, _ = Oracle.ivf_search(1, METRIC_L2, cent, sizes, codes, lids, xq, 6, 10, M=M, pq=pq)
        check_knn(D, I, Do, Io, exact=True, name="arena")

    ids = np.arange(len(xb), dtype=np.int64) * 3 + 1
    # copy_lists (capacity = length rounded to the granule) followed by adds
    s0, c0, l0, _ = Oracle.build_ivf_lists(1, METRIC_L2, cent, xb[:5000], ids=ids[:5000], pq=pq)
    idx = fresh()
    idx.copy_lists(s0, c0, l0)
    idx.add_with_ids(xb[5000:5001], ids[5000:5001])  # one vector: one list grows
    idx.add_with_ids(xb[5001:9000], ids[5001:9000])
    check(idx, xb[:9000], ids[:9000])
    # many small adds: relocations leave holes, compaction keeps the arena bounded
    for a in range(9000, 12000, 250):
        idx.add_with_ids(xb[a:a + 250], ids[a:a + 250])
    check(idx, xb, ids)
    used, holes, alloc = idx.arena_stats()
    assert used <= 4 * (12000 + 64 * nlist) and holes <= used and used <= alloc  # (small arenas are not compacted)
    # reset and reuse
    idx.reset()
    assert idx.ntotal == 0 and idx.stored_vectors == 0 and (idx.search(xq, 3)[1] == -1).all()
    idx.add_with_ids(xb[:3000], ids[:3000])
    check(idx, xb[:3000], ids[:3000])
    # nlist beyond the LDS variant of the rank kernel
    big = 20000
    rs = np.random.RandomState(9)
    cb = rs.rand(big, 8).astype(np.float32)
    iv = faiss_amd.GpuIndexIVFFlat(res, 8, big, METRIC_L2)
    iv.copy_centroids(cb)
    x8 = rs.rand(30000, 8).astype(np.float32)
    iv.add(x8[:17000])
    iv.add(x8[17000:])
    lab = Oracle.ivf_assign(METRIC_L2, cb, x8)
    sizes = np.bincount(lab, minlength=big)
    probe = np.nonzero(sizes > 1)[0][:50]
    for l in probe:
        assert np.array_equal(iv.get_list_ids(int(l)), np.nonzero(lab == l)[0])  # insertion order = id order here
    assert iv.stored_vectors == 30000
