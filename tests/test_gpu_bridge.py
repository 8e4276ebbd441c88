"""GPU tests of the reference-side binding (integration/faiss_amd_bridge.h, compiled against the unmodified reference in
oracle/_ref): faiss::Index / IndexIVFInterface subclasses over the C ABI and the cloner functions
index_cpu_to_gpu / index_cpu_to_gpu_multiple / index_gpu_to_cpu (faiss/gpu/GpuCloner.cpp:43-522).  Everything here is
driven from the REFERENCE side: reference CPU indexes are built by the reference, cloned onto the backend by the bridge,
searched through faiss::Index::search, and cloned back.  Test pattern: faiss/gpu/test/TestGpuIndexIVFPQ.cpp:149-330
(copyFrom / copyTo), faiss/gpu/test/test_multi_gpu.py:15-98 (sharded == unsharded ids)."""
import numpy as np
import pytest

import faiss_amd
from compare import check_knn
from oracle.pyoracle import METRIC_INNER_PRODUCT, METRIC_L2, Ref, synthetic_dataset

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not Ref.available(), reason="oracle/_ref not shipped")]


@pytest.fixture(scope="module")
def bres():
    rs = [Ref.amd_resources(0) for _ in range(3)]
    yield rs
    for r in rs:
        Ref.amd_resources_free(r)


def _cpu_index(desc, d, xt, xb, metric=METRIC_L2, nprobe=8):
    r = Ref.index_factory(d, desc, metric)
    if "IVF" in desc:
        r.set_train_niter(5, 6)
        r.train(xt)
        r.set_nprobe(nprobe)
    r.add(xb)
    return r


@pytest.mark.parametrize("desc,metric", [("Flat", METRIC_L2), ("Flat", METRIC_INNER_PRODUCT), ("IVF64,Flat", METRIC_L2),
                                         ("IVF64,PQ16", METRIC_L2), ("IVF64,PQ16", METRIC_INNER_PRODUCT),
                                         ("IVF64,SQ8", METRIC_L2), ("IVF64,SQ8", METRIC_INNER_PRODUCT), ("IVF64,SQ4", METRIC_L2),
                                         ("IVF64,SQ6", METRIC_L2), ("IVF64,SQfp16", METRIC_L2)])
def test_index_cpu_to_gpu_and_back(bres, desc, metric):
    d, k = 64, 20
    xt, xb, xq = synthetic_dataset(d, 4000, 20000, 300, seed=41)
    cpu = _cpu_index(desc, d, xt, xb, metric)
    Dr, Ir = cpu.search(xq, k)
    gpu = Ref.index_cpu_to_gpu(bres[0], cpu)
    assert "Amd" in gpu.type_name() and gpu.ntotal == len(xb) and gpu.is_trained
    D, I = gpu.search(xq, k)
    check_knn(D, I, Dr, Ir, rtol=1e-4, name=desc + " cloned to the backend")
    # and back: the CPU copy holds the same lists, so the reference's own search reproduces its first answer exactly
    back = Ref.index_gpu_to_cpu(gpu)
    assert back.ntotal == len(xb)
    if "IVF" in desc:
        back.set_nprobe(8)
        s0, c0, i0 = cpu.lists()
        s1, c1, i1 = back.lists()
        assert np.array_equal(s0, s1) and np.array_equal(i0, i1) and np.array_equal(c0, c1)
        assert np.array_equal(back.centroids(), cpu.centroids())
        if "SQ" in desc:
            assert back.sq_info() == cpu.sq_info() and np.array_equal(back.sq_trained(), cpu.sq_trained())
    Db, Ib = back.search(xq, k)
    assert np.array_equal(Ib, Ir) and np.array_equal(Db, Dr)
    # incremental use of the clone through faiss::Index::add
    gpu.add(xb[:100])
    assert gpu.ntotal == len(xb) + 100


def test_search_parameters_and_ivf_surface_through_the_bridge(bres):
    d, k = 64, 10
    xt, xb, xq = synthetic_dataset(d, 4000, 20000, 100, seed=42)
    cpu = _cpu_index("IVF64,Flat", d, xt, xb, nprobe=2)
    gpu = Ref.index_cpu_to_gpu(bres[0], cpu)
    D2, I2 = gpu.search(xq, k)  # nprobe = 2 copied from the CPU index
    check_knn(D2, I2, *cpu.search(xq, k), rtol=1e-4, name="nprobe 2")
    D16, I16 = gpu.search_nprobe(xq, k, 16)  # SearchParametersIVF.nprobe
    cpu.set_nprobe(16)
    check_knn(D16, I16, *cpu.search(xq, k), rtol=1e-4, name="params nprobe 16")
    assert not np.array_equal(I2, I16)


def test_reference_ivfflat_on_backend_coarse_quantizer(res):
    """faiss::IndexIVFFlat with an AmdIndex as its quantizer: IndexIVF::add (quantizer->assign, IndexIVF.cpp:194) and
    IndexIVF::search (quantizer->search, :336-342) run on the backend; lists and results equal the all-CPU index built
    on the same centroids (up to coarse near-ties)."""
    d, nlist, k = 48, 64, 10
    xt, xb, xq = synthetic_dataset(d, 4000, 15000, 200, seed=43)
    cpu = _cpu_index("IVF64,Flat", d, xt, xb, nprobe=6)
    cent = cpu.centroids()
    amd_q = faiss_amd.GpuIndexFlatL2(res, d)
    q = Ref.adapter(amd_q)
    q.add(cent)  # a trained quantizer: IndexIVF::train leaves it alone (quantizer->ntotal == nlist)
    assert amd_q.ntotal == nlist
    hybrid = Ref.ivfflat_with_quantizer(q, d, nlist)
    assert hybrid.is_trained
    hybrid.add(xb)
    hybrid.set_nprobe(6)
    s0, c0, i0 = cpu.lists()
    s1, c1, i1 = hybrid.lists()
    assert np.abs(s0.astype(np.int64) - s1.astype(np.int64)).sum() <= 4
    D, I = hybrid.search(xq, k)
    check_knn(D, I, *cpu.search(xq, k), rtol=1e-4, tie_rtol=1e-4, name="IVFFlat over a backend quantizer")
    # the rest of the surface a quantizer is used through (faiss/Index.h:268, 297-307, 363-383)
    lab = q.assign(xq)
    assert np.array_equal(lab, amd_q.assign(xq))
    assert np.array_equal(q.reconstruct_n(3, 5), cent[3:8])
    keys = lab[:, 0]
    assert np.array_equal(q.compute_residual_n(xq, keys), xq - cent[keys])


@pytest.mark.parametrize("desc", ["Flat", "IVF64,Flat", "IVF64,PQ16", "IVF64,SQ8"])
@pytest.mark.parametrize("mode", ["replicas", "shards1", "shards2", "shards4", "shards_ivf"])
def test_index_cpu_to_gpu_multiple(bres, desc, mode):
    """faiss/gpu/test/test_multi_gpu.py:15-98: the multi-device clone answers like the CPU index (ids exactly, up to
    distance near-ties), whatever the layout: replicas, shards by id modulo / id range / list range, and one common
    coarse quantizer (IndexShardsIVF -> search_preassigned on every shard)."""
    if desc == "Flat" and mode in ("shards1", "shards4", "shards_ivf"):
        pytest.skip("IVF-only shard types")
    d, k = 64, 15
    xt, xb, xq = synthetic_dataset(d, 4000, 20000, 150, seed=44)
    cpu = _cpu_index(desc, d, xt, xb)
    Dr, Ir = cpu.search(xq, k)
    kw = dict(replicas=dict(shard=False), shards1=dict(shard=True, shard_type=1), shards2=dict(shard=True, shard_type=2),
              shards4=dict(shard=True, shard_type=4), shards_ivf=dict(shard=True, shard_type=1, common_ivf_quantizer=True))[mode]
    multi = Ref.index_cpu_to_gpu_multiple(bres, cpu, **kw)
    assert multi.ntotal == len(xb)
    name = multi.type_name()
    assert ("Replicas" in name) == (mode == "replicas") and ("ShardsIVF" in name) == (mode == "shards_ivf")
    D, I = multi.search(xq, k)
    check_knn(D, I, Dr, Ir, rtol=1e-4, name="%s %s" % (desc, mode))
    back = Ref.index_gpu_to_cpu(multi)
    assert back.ntotal == len(xb)
    if "IVF" in desc:
        back.set_nprobe(8)
    Db, Ib = back.search(xq, k)
    check_knn(Db, Ib, Dr, Ir, rtol=1e-6, name="%s %s back on the CPU" % (desc, mode))


@pytest.mark.parametrize("desc", ["Flat", "IVF64,Flat", "IVF64,PQ16", "IVF64,SQ8"])
def test_serialization_round_trip_through_the_cpu_index(bres, desc, tmp_path):
    """faiss/gpu/test/test_gpu_index_serialize.py: a GPU index is checkpointed as index_gpu_to_cpu -> write_index and restored as
    read_index -> index_cpu_to_gpu (the reference has no GPU file format of its own).  The restored backend index returns the
    bits of the original."""
    d, k = 64, 20
    xt, xb, xq = synthetic_dataset(d, 4000, 20000, 300, seed=43)
    cpu = _cpu_index(desc, d, xt, xb)
    gpu = Ref.index_cpu_to_gpu(bres[0], cpu)
    D0, I0 = gpu.search(xq, k)
    path = str(tmp_path / "ckpt.faissindex")
    Ref.write_index(Ref.index_gpu_to_cpu(gpu), path)
    del gpu
    restored_cpu = Ref.read_index(path, d)
    assert restored_cpu.ntotal == len(xb) and restored_cpu.is_trained
    if "IVF" in desc:
        restored_cpu.set_nprobe(8)  # (nprobe is not part of the file: faiss/impl/index_write.cpp write_ivf_header)
    gpu2 = Ref.index_cpu_to_gpu(bres[1], restored_cpu)
    assert "Amd" in gpu2.type_name() and gpu2.ntotal == len(xb)
    D1, I1 = gpu2.search(xq, k)
    assert np.array_equal(I1, I0) and np.array_equal(D1, D0)
    # the file is the reference's: its own CPU search on the restored index equals its search before the round trip
    Dr, Ir = cpu.search(xq, k)
    Db, Ib = restored_cpu.search(xq, k)
    assert np.array_equal(Ib, Ir) and np.array_equal(Db, Dr)


@pytest.mark.parametrize("kind,arg", [(0, 0), (1, 16), (2, 0)])
def test_caller_owned_backend_quantizer_through_the_bridge(bres, kind, arg):
    """GpuIndexIVF*(provider, Index* coarseQuantizer, ...) (faiss/gpu/GpuIndexIVF.cu:41-70) with a flat index OF THE BACKEND as the
    caller's quantizer: handed to the device side, not owned, trained in place when empty, shared by two indexes."""
    d, nlist, k, nprobe = 64, 32, 15, 6
    xt, xb, xq = synthetic_dataset(d, 4000, 20000, 300, seed=47)
    q = Ref.amd_flat(bres[0], d)
    a = Ref.amd_ivf_with_quantizer(bres[0], q, kind, d, nlist, arg)
    assert not a.is_trained and q.ntotal == 0
    a.set_train_niter(5, 6)
    a.train(xt)
    assert a.is_trained and q.ntotal == nlist  # the caller's quantizer holds the centroids now
    b = Ref.amd_ivf_with_quantizer(bres[0], q, 0, d, nlist)  # a second index over the same (now trained) quantizer
    assert b.is_trained
    for idx in (a, b):
        idx.add(xb)
        idx.set_nprobe(nprobe)
    D, I = a.search(xq, k)
    assert (I[:, 0] >= 0).all() and a.ntotal == len(xb)
    Dq, Iq = q.search(xq, nprobe)  # the quantizer stays an ordinary index of the caller's
    assert Iq.min() >= 0 and Iq.max() < nlist
    Db, Ib = b.search(xq, k)
    # b (IVFFlat over q) == a reference IVFFlat built on the host from the same centroids, within the flat tolerance
    cpu = Ref.index_factory(d, "IVF%d,Flat" % nlist, METRIC_L2)
    cpu.set_centroids(q.reconstruct_n(0, nlist))
    cpu.add(xb)
    cpu.set_nprobe(nprobe)
    Dr, Ir = cpu.search(xq, k)
    check_knn(Db, Ib, Dr, Ir, rtol=1e-4, name="IVFFlat over a caller-owned backend quantizer")
    del a, b
    assert q.ntotal == nlist and q.search(xq[:3], 2)[1].shape == (3, 2)  # the quantizer outlives the indexes


@pytest.mark.parametrize("kind,arg,qdesc", [(0, 0, "Flat"), (1, 16, "Flat"), (2, 0, "Flat"), (0, 0, "HNSW16")])
def test_cpu_coarse_quantizer_through_the_bridge(bres, kind, arg, qdesc):
    """GpuIndexIVFConfig::allowCpuCoarseQuantizer (faiss/gpu/GpuIndexIVF.h:23-35, impl/IVFBase.cu:526-546): the coarse quantizer is
    ANY faiss::Index on the host (IndexFlatL2, IndexHNSWFlat): its search / assign run there and feed search_preassigned /
    add_core; the lists live on the device.  Flat CPU quantizer: the lists are those the quantizer's own assignment gives, and
    the search agrees with the reference's CPU search of the index cloned back (same lists, same quantizer) to 1e-4; HNSW (an
    approximate quantizer) is checked for recall."""
    d, nlist, k, nprobe = 64, 32, 15, 6
    xt, xb, xq = synthetic_dataset(d, 4000, 20000, 300, seed=53)
    cq = Ref.index_factory(d, qdesc, METRIC_L2)
    g = Ref.amd_ivf_with_quantizer(bres[0], cq, kind, d, nlist, arg)
    assert not g.is_trained
    g.set_train_niter(5, 6)
    g.train(xt)
    assert g.is_trained and cq.ntotal == nlist
    g.add(xb)
    g.set_nprobe(nprobe)
    assert g.ntotal == len(xb)
    D, I = g.search(xq, k)
    assert (I[:, 0] >= 0).all()
    if qdesc == "Flat":
        back = Ref.index_gpu_to_cpu(g)
        sizes, _, ids = back.lists()
        lab = cq.search(xb, 1)[1][:, 0]
        assert np.array_equal(np.bincount(lab, minlength=nlist).astype(sizes.dtype), sizes)
        order = np.argsort(lab, kind="stable")
        assert np.array_equal(ids, order)  # insertion order inside every list
        assert np.array_equal(back.centroids(), cq.reconstruct_n(0, nlist))
        back.set_nprobe(nprobe)
        Db, Ib = back.search(xq, k)
        check_knn(D, I, Db, Ib, rtol=1e-4, name="CPU coarse quantizer: device lists vs the cloned-back CPU index")
    else:
        flat = Ref.index_factory(d, "Flat", METRIC_L2)
        flat.add(xb)
        _, gt = flat.search(xq, 1)
        assert (I == gt[:, :1]).any(axis=1).mean() > 0.8


def test_autotune_explores_a_backend_index(bres):
    """faiss::gpu::GpuParameterSpace::initialize + ParameterSpace::explore (faiss/gpu/GpuAutoTune.cpp:33-114, faiss/AutoTune.cpp:632-737;
    VERDICT r5 missing 4): the bridge's parameter space lists nprobe = 1 ... below nlist, the REFERENCE's explore() sets each value on
    the backend index, searches and keeps the optimal (recall, time) points; recall must grow to 1-recall@1 of the exhaustive probe."""
    d, nlist = 32, 64
    xt, xb, xq = synthetic_dataset(d, 4000, 30000, 400, seed=59)
    cpu = _cpu_index("IVF%d,Flat" % nlist, d, xt, xb)
    flat = Ref.index_factory(d, "Flat", METRIC_L2)
    flat.add(xb)
    _, gt = flat.search(xq, 1)
    gpu = Ref.index_cpu_to_gpu(bres[0], cpu)
    nranges, pts = Ref.amd_autotune(gpu, xq, 10, gt[:, 0])
    assert nranges == 1 and len(pts) >= 3
    perfs = [p[0] for p in pts]
    assert perfs == sorted(perfs) and perfs[-1] > 0.97 and perfs[0] < perfs[-1]
    assert all(key.startswith("nprobe=") or key == "" for _, _, key in pts)
    nps = [int(key.split("=")[1]) for _, _, key in pts if key]
    assert all(n < nlist and (n & (n - 1)) == 0 for n in nps)
