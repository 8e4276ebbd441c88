"""GPU tests of the interface corners added at the end of round 2 and first run on hardware in round 3 (all green on
the first run, gpurun_out/r3a_validate.log):
  * the "extra" metrics of GpuIndexFlat / bfKnn (L1, Linf, Lp, Canberra, BrayCurtis, JensenShannon, Jaccard:
    faiss/gpu/impl/GeneralDistance.cuh over the functors of faiss/gpu/impl/DistanceUtils.cuh:47-281; reference tests
    faiss/gpu/test/TestGpuIndexFlat.cpp L1_Float32 / Lp_Float32, TestGpuDistance.cu L1 .. Jaccard); the oracle they are
    compared with is pinned on the real reference by tests/test_golden_extra_cpu.py;
  * reserveMemory / reclaimMemory / updateQuantizer and the IVFPQ getters of the reference's GPU index classes;
  * Index::search_and_reconstruct."""
import os

import numpy as np
import pytest

import faiss_amd
from compare import check_knn
from oracle.pyoracle import Oracle, Ref
from test_golden_extra_cpu import EXTRA_METRICS, GOLD, positive_dataset

pytestmark = pytest.mark.gpu
EXACT = {2, 3, 20, 21, 23}  # no transcendental function: bit-exact against the oracle


@pytest.mark.parametrize("tag,metric,arg", EXTRA_METRICS, ids=[m[0] for m in EXTRA_METRICS])
@pytest.mark.parametrize("d,nb,nq,k", [(40, 3000, 64, 20), (128, 20000, 33, 100), (7, 500, 5, 600)])
def test_flat_extra_metric_matches_oracle(res, tag, metric, arg, d, nb, nq, k):
    xb, xq = positive_dataset(d, nb, nq, 5)
    idx = faiss_amd.GpuIndexFlat(res, d, metric)
    idx.metric_arg = arg
    idx.add(xb)
    D, I = idx.search(xq, k)
    Do, Io = Oracle.flat_search_general(metric, xb, xq, k, metric_arg=arg)
    if metric in EXACT:
        check_knn(D, I, Do, Io, exact=True, name="extra metric " + tag)
    else:
        check_knn(D, I, Do, Io, rtol=1e-5, name="extra metric " + tag)
    if (d, nb) == (40, 3000):
        z = np.load(os.path.join(GOLD, "flat_extra_metrics.npz"))
        check_knn(D, I, z["D_" + tag], z["I_" + tag], rtol=1e-4, name="extra metric vs reference golden " + tag)
    # bfKnn on raw arrays: the same results
    D2, I2 = faiss_amd.knn_gpu(res, xq, xb, k, metric=metric, metric_arg=arg)
    assert np.array_equal(I2, I) and np.array_equal(D2, D)


def test_flat_extra_metric_fp16_storage_and_incremental_add(res):
    d, nb, nq, k = 32, 4000, 20, 10
    xb, xq = positive_dataset(d, nb, nq, 9)
    idx = faiss_amd.GpuIndexFlat(res, d, faiss_amd.METRIC_L1, config=faiss_amd.GpuIndexFlatConfig(useFloat16=True))
    idx.add(xb[:1500])
    idx.add(xb[1500:])
    D, I = idx.search(xq, k)
    xbh, xqh = xb.astype(np.float16).astype(np.float32), xq.astype(np.float16).astype(np.float32)
    Do, Io = Oracle.flat_search_general(2, xbh, xqh, k)
    check_knn(D, I, Do, Io, exact=True, name="L1 on fp16 storage")


def test_extra_metrics_are_flat_only(res):
    with pytest.raises(faiss_amd.FaissAmdError, match="unsupported metric"):
        faiss_amd.GpuIndexIVFFlat(res, 32, 16, faiss_amd.METRIC_L1)
    idx = faiss_amd.GpuIndexFlat(res, 16, faiss_amd.METRIC_Canberra)
    xb, xq = positive_dataset(16, 100, 4, 2)
    idx.add(xb)
    with pytest.raises(faiss_amd.FaissAmdError, match="IDSelector"):
        idx.search(xq, 5, params=faiss_amd.SearchParameters(sel=faiss_amd.IDSelectorRange(0, 50)))


@pytest.mark.skipif(not Ref.available(), reason="oracle/_ref not shipped")
@pytest.mark.parametrize("tag,metric,arg", EXTRA_METRICS, ids=[m[0] for m in EXTRA_METRICS])
def test_extra_metric_through_the_bridge(tag, metric, arg):
    """IndexFlat(d, metric) of the reference cloned to the backend (metric_arg travels), searched through faiss::Index"""
    d, nb, nq, k = 40, 3000, 64, 20
    xb, xq = positive_dataset(d, nb, nq, 5)
    cpu = Ref.index_factory(d, "Flat", metric)
    if metric == 4:
        cpu.set_metric_arg(arg)
    cpu.add(xb)
    Dr, Ir = cpu.search(xq, k)
    bres = Ref.amd_resources(0)
    try:
        gpu = Ref.index_cpu_to_gpu(bres, cpu)
        D, I = gpu.search(xq, k)
        check_knn(D, I, Dr, Ir, rtol=1e-4, name="bridge extra metric " + tag)
        del gpu
    finally:
        Ref.amd_resources_free(bres)


def test_ivf_reserve_and_reclaim_memory(res):
    """GpuIndexIVFFlat::reserveMemory / reclaimMemory (faiss/gpu/GpuIndexIVFFlat.h:64-76), GpuIndexIVFPQ getters: the arena
    does not grow during the add a reservation covers, reclaiming shrinks it and changes no result."""
    from oracle.pyoracle import synthetic_dataset
    d, nlist, nb, nq, k = 32, 64, 30000, 200, 10
    xt, xb, xq = synthetic_dataset(d, 4000, nb, nq, seed=6)
    cent, _ = faiss_amd.kmeans(res, xt, nlist, niter=4, seed=3)
    for kind in (0, 1):
        if kind == 0:
            idx = faiss_amd.GpuIndexIVFFlat(res, d, nlist, faiss_amd.METRIC_L2)
        else:
            idx = faiss_amd.GpuIndexIVFPQ(res, d, nlist, 8, 8, faiss_amd.METRIC_L2)
            idx.copy_pq_centroids((np.random.RandomState(7).rand(8, 256, 4).astype("float32") - 0.5) * 0.4)
            assert idx.getNumSubQuantizers() == 8 and idx.getBitsPerCode() == 8 and idx.getCentroidsPerSubQuantizer() == 256
            # the getters report what is in force: L2 always uses the per-vector term, tables are always fp32
            assert idx.getPrecomputedCodes() and idx.getTableInfo() == (True, False, False)
            idx.setPrecomputedCodes(True)
            assert idx.getTableInfo() == (True, True, False)
            ip = faiss_amd.GpuIndexIVFPQ(res, d, nlist, 8, 8, faiss_amd.METRIC_INNER_PRODUCT)
            ip.setPrecomputedCodes(True)
            assert not ip.getPrecomputedCodes() and ip.getTableInfo() == (False, True, False)
        idx.copy_centroids(cent)
        idx.nprobe = 8
        idx.reserveMemory(nb)
        alloc0 = idx.arena_stats()[2]
        assert alloc0 >= nb
        idx.add(xb)  # one call: every list gets its final capacity at once, inside the reservation
        assert idx.arena_stats()[2] == alloc0, "the reservation did not cover the add"
        D0, I0 = idx.search(xq, k)
        lists0 = [(idx.get_list_ids(l).copy(), idx.get_list_codes(l).copy()) for l in range(nlist)]
        freed = idx.reclaimMemory()
        used, holes, alloc1 = idx.arena_stats()
        assert freed > 0 and alloc1 < alloc0 and holes == 0 and alloc1 <= nb + nlist * 64 + 64
        D1, I1 = idx.search(xq, k)
        assert np.array_equal(D0, D1) and np.array_equal(I0, I1)
        for l in range(nlist):
            assert np.array_equal(idx.get_list_ids(l), lists0[l][0]) and np.array_equal(idx.get_list_codes(l), lists0[l][1])
        idx.add(xb[:3000])  # the index keeps working after a reclaim
        assert idx.ntotal == nb + 3000
        idx.updateQuantizer()
        D2, I2 = idx.search(xq, k)
        assert (I2[:, 0] >= 0).all()


def test_flat_search_and_reconstruct(res):
    """faiss::Index::search_and_reconstruct on the flat index (TestGpuIndexFlat.cpp SearchAndReconstruct): composed of
    search + reconstruct_batch."""
    from oracle.pyoracle import synthetic_dataset
    d, nb, nq, k = 32, 50, 6, 60  # k > nb: the tail of every row is missing
    _, xb, xq = synthetic_dataset(d, 0, nb, nq, seed=2)
    idx = faiss_amd.GpuIndexFlatL2(res, d)
    idx.add(xb)
    D, I, R = idx.search_and_reconstruct(xq, k)
    assert R.shape == (nq, k, d) and (I[:, nb:] == -1).all()
    assert np.array_equal(R[:, :nb], xb[I[:, :nb]]) and np.isnan(R[:, nb:]).all()


# ------------------------------------------------------------------------------- ClusteringParameters (faiss/Clustering.h:27-60)
@pytest.mark.parametrize("flags", [dict(spherical=1), dict(int_centroids=1), dict(nredo=3), dict(frozen_centroids=1),
                                   dict(spherical=1, nredo=2, frozen_centroids=1)])
def test_clustering_parameters_device_loop_equals_host_loop(res, flags):
    """nredo / spherical / int_centroids / frozen_centroids: the device-resident loop and the loop driven through
    add() / search() (the reference's organisation, faiss/Clustering.cpp:255-420) give the same centroids bit for bit,
    and each flag does what faiss::ClusteringParameters says."""
    from oracle.pyoracle import synthetic_dataset
    d, n, k, niter = 24, 12000, 30, 5
    xt, _, _ = synthetic_dataset(d, n, 0, 0, seed=77)
    if flags.get("int_centroids"):
        xt = (xt * 20).astype(np.float32)
    metric = faiss_amd.METRIC_INNER_PRODUCT if flags.get("spherical") else faiss_amd.METRIC_L2
    init = None
    if flags.get("frozen_centroids"):
        init = xt[:7].copy() * (1.0 if not flags.get("spherical") else 1.0 / np.linalg.norm(xt[:7], axis=1, keepdims=True))
        init = init.astype(np.float32)
    runs = []
    for engine in ("device", "host"):
        if engine == "device":
            ix = faiss_amd.GpuIndexFlat(res, d, metric)
        else:
            ix = faiss_amd.IndexReplicas(d, threaded=False)
            ix.add_replica(faiss_amd.GpuIndexFlat(res, d, metric))
        c = faiss_amd.Clustering(d, k, niter=niter, seed=5, **flags)
        c.centroids = init
        c.train(xt, ix)
        assert c.on_device == (engine == "device") and ix.ntotal == k
        runs.append(c)
    a, b = runs
    assert np.array_equal(a.centroids, b.centroids) and np.array_equal(a.obj, b.obj)
    if flags.get("spherical"):
        free = a.centroids[7:] if init is not None else a.centroids
        assert np.allclose(np.linalg.norm(free, axis=1), 1.0, atol=1e-5)
    if flags.get("int_centroids"):
        assert np.array_equal(a.centroids, np.round(a.centroids))
    if flags.get("frozen_centroids"):
        assert np.array_equal(a.centroids[:7], init)
    if flags.get("nredo", 1) > 1 and not flags.get("frozen_centroids"):
        # the winner is at least as good as every single run with the seeds the redo loop uses (seed + 1 + redo)
        finals = []
        for redo in range(flags["nredo"]):
            one = faiss_amd.Clustering(d, k, niter=niter, seed=5 + 1 + redo, **{f: v for f, v in flags.items() if f != "nredo"})
            one.train(xt, faiss_amd.GpuIndexFlat(res, d, metric))
            finals.append(float(one.obj[-1]))
        best = max(finals) if metric == faiss_amd.METRIC_INNER_PRODUCT else min(finals)
        assert float(a.obj[-1]) == best


def test_ivf_training_honours_clustering_parameters(res):
    """GpuIndexIVF::cp: spherical k-means for an inner-product IVF index (unit-norm coarse centroids), through the ABI"""
    from oracle.pyoracle import synthetic_dataset
    d, nlist = 32, 24
    xt, xb, xq = synthetic_dataset(d, 6000, 4000, 20, seed=3)
    idx = faiss_amd.GpuIndexIVFFlat(res, d, nlist, faiss_amd.METRIC_INNER_PRODUCT)
    idx.set_clustering_params(niter=6, spherical=1, nredo=2, seed=9)
    idx.train(xt)
    cent = idx.get_centroids()
    assert np.allclose(np.linalg.norm(cent, axis=1), 1.0, atol=1e-5)
    idx.add(xb)
    idx.nprobe = nlist
    D, I = idx.search(xq, 5)
    Do, Io = Oracle.flat_search(faiss_amd.METRIC_INNER_PRODUCT, xb, xq, 5)
    check_knn(D, I, Do, Io, rtol=1e-4, name="IP IVF with spherical coarse quantizer, all lists probed")
    with pytest.raises(faiss_amd.FaissAmdError):
        idx.set_clustering_params(nredo=0)


@pytest.mark.parametrize("kind", ["flat", "ivfpq", "ivfflat_listmajor"])
def test_concurrent_searches_on_one_index_equal_serial(res, kind):
    """Four host threads searching ONE index at the same time (ctypes drops the GIL): the entry points serialise on the
    index's lock, every thread gets exactly the serial answer (faiss/impl/ThreadedIndex-inl.h:80-133 is what drives
    replicas from threads; an index object of the reference's GPU classes is not re-entrant at all)."""
    import threading
    from oracle.pyoracle import synthetic_dataset
    d, nb = 64, 30000
    xt, xb, xq = synthetic_dataset(d, 4000, nb, 2400, seed=21)
    if kind == "flat":
        idx = faiss_amd.GpuIndexFlatL2(res, d)
    elif kind == "ivfpq":
        idx = faiss_amd.GpuIndexIVFPQ(res, d, 32, 16, 8, faiss_amd.METRIC_L2)
        idx.train(xt)
    else:
        idx = faiss_amd.GpuIndexIVFFlat(res, d, 32, faiss_amd.METRIC_L2)
        idx.train(xt)
    idx.add(xb)
    if kind != "flat":
        idx.nprobe = 8
    blocks = [xq[i * 600:(i + 1) * 600] for i in range(4)]
    if kind == "ivfflat_listmajor":
        idx.set_scan_mode(idx.SCAN_LIST_MAJOR)
    want = [idx.search(b, 10) for b in blocks]
    got = [None] * 4

    def run(i):
        for _ in range(3):
            got[i] = idx.search(blocks[i], 10)

    th = [threading.Thread(target=run, args=(i,)) for i in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for i in range(4):
        assert np.array_equal(got[i][0], want[i][0]) and np.array_equal(got[i][1], want[i][1])


@pytest.mark.gpu
@pytest.mark.parametrize("d,M,nt,tight", [(128, 64, 70000, False), (32, 8, 3000, False), (64, 4, 2500, True), (24, 24, 1200, False)])
def test_pq_training_as_one_kmeans_equals_the_per_subspace_loop(res, d, M, nt, tight, monkeypatch):
    """The M sub-quantizers trained as ONE k-means over M * nt points (labels m * 256 + c, one counting sort and one
    update launch per iteration) against one Clustering per sub-space (faiss/impl/ProductQuantizer.cpp:140-190, the
    round-2 loop): the same codebook bit for bit.  (128, 64, 70000): the bench shape, training set sub-sampled to 65536;
    (64, 4, 2500, tight): 2500 points in a few tight blobs for 256 centroids leave empty clusters -- the refill of the
    sub-space's own generator; (24, 24): dsub = 1."""
    from faiss_amd.datasets import synthetic_dataset
    if tight:
        rs = np.random.RandomState(3)
        xt = (rs.randint(0, 6, size=(nt, 1)) * 5 + rs.rand(nt, d) * 0.01).astype(np.float32)
    else:
        xt, _, _ = synthetic_dataset(d, nt, 0, 0, seed=d + M)
    nlist = 16
    cent = xt[:: nt // nlist][:nlist].copy()
    books = []
    monkeypatch.setenv("FAISS_AMD_EXPERIMENTS", "1")  # the library reads its knobs only behind this gate
    for loop in ("1", "0"):
        monkeypatch.setenv("FAISS_AMD_PQ_TRAIN_LOOP", loop)
        idx = faiss_amd.GpuIndexIVFPQ(res, d, nlist, M, 8, faiss_amd.METRIC_L2)
        idx.train(xt)
        books.append((idx.get_centroids(), idx.get_pq_centroids()))
    assert np.array_equal(books[0][0], books[1][0])
    assert np.array_equal(books[0][1], books[1][1]), "batched product-quantizer training differs from the per-sub-space loop"
    assert np.isfinite(books[1][1]).all()
