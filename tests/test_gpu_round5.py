"""GPU tests of round 5's host-side changes to the IVF filter path: memory policy of the sweeps' copies (opt-out, report,
fallback), the redo set in scratch-bounded sub-tiles, incremental maintenance of the copies on add(), the sampled first
sweep + tightening, the scalar quantizer behind the filter.  Every comparison is bit-exact (same arithmetic on both sides).
Reference behaviour: faiss/gpu/GpuIndexIVF.cu:321-406 (add / search), impl/IVFBase.cu:595-905 (append in place)."""
import numpy as np
import pytest

import faiss_amd
from oracle.pyoracle import METRIC_INNER_PRODUCT, METRIC_L2, synthetic_dataset

pytestmark = pytest.mark.gpu


def _make(res, kind, d, nlist, M=0, metric=METRIC_L2):
    if kind == 0:
        return faiss_amd.GpuIndexIVFFlat(res, d, nlist, metric)
    if kind == 1:
        return faiss_amd.GpuIndexIVFPQ(res, d, nlist, M, 8, metric)
    return faiss_amd.GpuIndexIVFScalarQuantizer(res, d, nlist, faiss_amd.ScalarQuantizer.QT_8bit, metric, True)


@pytest.mark.parametrize("kind", [0, 1])
def test_filter_shadow_opt_out_and_memory_report(res, kind):
    """set_use_filter_shadow(False): the automatic mode serves a large batch without the sweeps' copy of the lists (no
    device bytes held for it) and returns the same bits; with it the copy shows up in resident_bytes at the documented
    size (IVFFlat + 2 d bytes per row, IVFPQ + M bytes per row, in whole 32-row blocks)."""
    d, nlist, M, nb, nq, k = 64, 32, 16, 60000, 4096, 20
    xt, xb, xq = synthetic_dataset(d, 4000, nb, nq, seed=11)
    idx = _make(res, kind, d, nlist, M)
    idx.train(xt)
    idx.add(xb)
    idx.nprobe = 8
    idx.set_use_filter_shadow(False)
    D0, I0 = idx.search(xq, k)
    lists, shadow = idx.resident_bytes()
    assert idx.scan_info()[1] == 1 and shadow <= (64 << 10) and lists >= nb * (4 * d if kind == 0 else M)
    idx.set_use_filter_shadow(True)
    idx.set_scan_mode(2)  # (the automatic rule may prefer the query-major scan at this size: ask for the sweeps)
    D1, I1 = idx.search(xq, k)
    lists1, shadow1 = idx.resident_bytes()
    assert idx.scan_info()[1] == 2 and idx.last_scan_arith() == 0
    per_row = 2 * d if kind == 0 else M
    assert lists1 == lists and nb * per_row <= shadow1 <= 2.2 * nb * per_row + (1 << 20)
    assert np.array_equal(I0, I1) and np.array_equal(D0, D1)
    freed = idx.reclaimMemory()
    assert freed >= shadow1 and idx.resident_bytes()[1] <= (64 << 10)
    D2, I2 = idx.search(xq, k)  # rebuilt on demand
    assert np.array_equal(I0, I2) and np.array_equal(D0, D2)


def test_redo_set_goes_through_the_key_scan_in_sub_tiles(res):
    """with the fused scan switched off the redo of a list-major search (here: EVERY query -- fewer granules than
    k under a forced scan_mode 2) runs the key-segment scan, whose scratch is nprobe x longest list x 8 bytes per query; the
    redo set is cut into sub-tiles of the scratch budget instead of asking for all of it at once."""
    d, nlist, nb, nq, k = 32, 4, 40000, 600, 50
    xt, xb, xq = synthetic_dataset(d, 2000, nb, nq, seed=5)
    ref = faiss_amd.GpuIndexIVFFlat(res, d, nlist, METRIC_L2)
    ref.train(xt)
    ref.add(xb)
    ref.nprobe = 4
    ref.set_scan_mode(1)
    Dr, Ir = ref.search(xq, k)
    res2 = faiss_amd.StandardGpuResources(0)
    res2.setTempMemory(64 << 20)  # the minimum, 64 MiB: ~200 queries of 4 x ~10 000 rows x 8 bytes per sub-tile (600 are redone)
    idx = faiss_amd.GpuIndexIVFFlat(res2, d, nlist, METRIC_L2)
    idx.copy_centroids(ref.get_centroids())
    idx.add(xb)
    idx.nprobe = 4
    idx.set_use_fused_scan(False)
    idx.set_scan_mode(2)
    # far outside the fp16 range: every query is sent to the redo path
    big = xq.copy()
    big[:, 0] = 1.0e6
    ref.set_scan_mode(1)
    Db, Ib = ref.search(big, k)
    before = idx.scan_info()[2]
    D, I = idx.search(big, k)
    assert idx.scan_info()[2] - before == nq, "the redo path was not exercised"
    assert np.array_equal(I, Ib) and np.array_equal(D, Db)
    D, I = idx.search(xq, k)
    assert np.array_equal(I, Ir) and np.array_equal(D, Dr)


def test_index_shards_over_the_devices_of_this_box():
    """faiss/IndexShards.cpp:196-265 + utils/Heap.cpp:166-240 across DEVICES in one process: one IVFPQ shard per GPU (up to
    two), shared quantizers, global ids, threaded search, host merge -- and the device-to-device variant: the second shard's
    top-k copied to device 0 (hipMemcpyPeer through torch) and merged by the device kernel.  Both equal the one-index search
    bit for bit.  On a 1-GPU box only the multi-device half is skipped; the first 2-GPU box exercises it without code changes."""
    import torch
    ngpu = min(2, faiss_amd.get_num_gpus())
    d, nlist, M, nb, nq, k = 64, 32, 16, 40000, 500, 30
    xt, xb, xq = synthetic_dataset(d, 4000, nb, nq, seed=21)
    ress = [faiss_amd.StandardGpuResources(g) for g in range(ngpu)]
    single = faiss_amd.GpuIndexIVFPQ(ress[0], d, nlist, M, 8, METRIC_L2)
    single.train(xt)
    single.add(xb)
    single.nprobe = 8
    Dr, Ir = single.search(xq, k)
    cent, pqc = single.get_centroids(), single.get_pq_centroids()
    nshard = 2
    bounds = [(s * nb // nshard, (s + 1) * nb // nshard) for s in range(nshard)]
    shards = []
    for s, (lo, hi) in enumerate(bounds):
        ix = faiss_amd.GpuIndexIVFPQ(ress[s % ngpu], d, nlist, M, 8, METRIC_L2)
        ix.copy_centroids(cent)
        ix.copy_pq_centroids(pqc)
        ix.add_with_ids(xb[lo:hi], np.arange(lo, hi, dtype=np.int64))
        ix.nprobe = 8
        shards.append(ix)
    sh = faiss_amd.IndexShards(d, threaded=True, successive_ids=False)
    for ix in shards:
        sh.add_shard(ix)
    D, I = sh.search(xq, k)
    assert np.array_equal(I, Ir) and np.array_equal(D, Dr), "host merge over %d device(s)" % ngpu
    if ngpu < 2:
        pytest.skip("one GPU on this box: the device-to-device merge needs two (host merge over one device checked)")
    # device-to-device: per-shard results stay on their GPUs, shard 1's travel to device 0, device merge there
    outs = []
    for s, ix in enumerate(shards):
        dev = torch.device("cuda", s % ngpu)
        with torch.cuda.device(dev):
            xq_d = torch.from_numpy(xq).to(dev)
            Dd = torch.empty((nq, k), dtype=torch.float32, device=dev)
            Id = torch.empty((nq, k), dtype=torch.int64, device=dev)
            ix.search_ptr(nq, xq_d.data_ptr(), k, Dd.data_ptr(), Id.data_ptr())
            torch.cuda.synchronize(dev)
        outs.append((Dd, Id))
    dev0 = torch.device("cuda", 0)
    allD = torch.stack([o[0].to(dev0) for o in outs]).contiguous()
    allI = torch.stack([o[1].to(dev0) for o in outs]).contiguous()
    Dm = torch.empty((nq, k), dtype=torch.float32, device=dev0)
    Im = torch.empty((nq, k), dtype=torch.int64, device=dev0)
    torch.cuda.synchronize(dev0)
    faiss_amd.merge_knn_results_device(ress[0], METRIC_L2, nq, k, nshard, allD.data_ptr(), allI.data_ptr(), None,
                                       Dm.data_ptr(), Im.data_ptr())
    assert np.array_equal(Im.cpu().numpy(), Ir) and np.array_equal(Dm.cpu().numpy(), Dr), "device merge across two GPUs"


@pytest.mark.parametrize("kind,metric,d,M,nlist,nb,nprobe,k", [
    (0, METRIC_L2, 64, 0, 8, 90000, 3, 50),                # lists of ~11 000 rows: row chunks of 2816 rows, prefix 768
    (0, METRIC_INNER_PRODUCT, 128, 0, 16, 60000, 6, 100),
    (1, METRIC_L2, 64, 32, 8, 90000, 3, 50),
    (1, METRIC_L2, 128, 64, 32, 120000, 8, 100),            # bench shape of the sweeps (PQ64, dsub 2)
    (1, METRIC_INNER_PRODUCT, 32, 16, 8, 50000, 4, 10),
    (0, METRIC_L2, 256, 0, 8, 40000, 4, 300),               # d > 128: two query blocks per item; k above the fused selection
])
def test_sampled_first_sweep_and_tightening_keep_the_bits(res, kind, metric, d, M, nlist, nb, nprobe, k):
    """Sweep 1 of the filter path on a prefix of every work item (set_lmf_sampling 1 ... 4, and the rule) bounds the k-th best
    estimate from a SAMPLE; sweep 2 then collects more rows and the tightening launch cuts them back.  Whatever the sample,
    the results are the query-major scan's bit for bit, nobody is redone, and the rerank sees no more candidates than
    without sampling (the tightened set does not depend on the sample at all)."""
    xt, xb, xq = synthetic_dataset(d, 4000, nb, 700, seed=d + nlist)
    idx = _make(res, kind, d, nlist, M, metric)
    idx.train(xt)
    idx.add(xb)
    idx.nprobe = nprobe
    idx.set_scan_mode(1)
    Dr, Ir = idx.search(xq, k)
    idx.set_scan_mode(2)
    before = idx.scan_info()[2]
    for shift in (-1, 0, 1, 2, 3, 4):
        idx.set_lmf_sampling(shift)
        D, I = idx.search(xq, k)
        assert idx.scan_info()[1] == 2 and idx.last_scan_arith() == 0
        assert np.array_equal(I, Ir) and np.array_equal(D, Dr), "sample shift %d" % shift
    assert idx.scan_info()[2] == before, "queries were redone"
    idx.set_lmf_sampling(0)


@pytest.mark.parametrize("kind,metric", [(0, METRIC_L2), (1, METRIC_L2), (0, METRIC_INNER_PRODUCT)])
def test_add_keeps_the_sweeps_copy_of_the_lists_up_to_date(res, kind, metric):
    """add() after a list-major search patches the sweeps' copy of the lists in place (the 32-row blocks that received rows,
    every block of a relocated list) instead of invalidating it: many small and a few large adds interleaved with searches
    -- lists outgrow their slack and move, the arena is reallocated, reclaimMemory / a forced compaction drop the copy in
    between -- and every search returns the bits of the query-major scan on the same index, and of an index built in one
    add.  resident_bytes shows the copy alive (never rebuilt from nothing) across the adds."""
    d, nlist, M, nq, k = 64, 16, 16, 300, 40
    rs = np.random.RandomState(9)
    xt, xb, xq = synthetic_dataset(d, 3000, 90000, nq, seed=31)
    idx = _make(res, kind, d, nlist, M, metric)
    idx.train(xt)
    idx.nprobe = 5
    sizes = [3000, 17, 1, 2000, 64, 5, 33, 20000, 100, 1, 31, 32, 40000, 7, 999]
    done = 0
    for step, n in enumerate(sizes):
        idx.add(xb[done:done + n])
        done += n
        if step == 0:
            idx.set_scan_mode(2)
            idx.search(xq, k)  # builds the copy
            assert idx.resident_bytes()[1] > 0
            continue
        assert idx.resident_bytes()[1] > 0, "the copy was dropped by add()"
        if step == 9:
            idx.reclaimMemory()  # drops the copy and compacts the lists: the next search rebuilds it
        idx.set_scan_mode(2)
        D, I = idx.search(xq, k)
        assert idx.scan_info()[1] == 2 and idx.last_scan_arith() == 0
        idx.set_scan_mode(1)
        Dr, Ir = idx.search(xq, k)
        assert np.array_equal(I, Ir) and np.array_equal(D, Dr), "after add #%d (%d rows)" % (step, n)
    fresh = _make(res, kind, d, nlist, M, metric)
    fresh.copy_centroids(idx.get_centroids())
    if kind == 1:
        fresh.copy_pq_centroids(idx.get_pq_centroids())
    fresh.add(xb[:done])
    fresh.nprobe = 5
    fresh.set_scan_mode(2)
    Df, If = fresh.search(xq, k)
    assert np.array_equal(I, If) and np.array_equal(D, Df)


def test_add_then_search_loop_at_10m_stays_list_major_and_fast(res):
    """alternate add(1000) / search(4096) 50 times on an IVFFlat index of 10M rows.  Before round 5 every add
    invalidated the fp16 shadow and the next list-major search rebuilt ALL of it (an O(nb) pass, ~10 ms at this size); now the
    add patches the blocks it touched.  Asserted: list-major every time, the last results equal those of the query-major
    scan, and the searches behind an add cost what a search without an add costs (within 25 %, medians)."""
    import time
    import torch
    from faiss_amd.datasets import synthetic_dataset as sd, synthetic_more_device
    d, nlist, nq, k = 128, 4096, 4096, 100
    xt, xb, xq, dmap = sd(d, 100000, 1000000, nq, seed=1338, return_map=True)
    dev = torch.device("cuda", 0)
    idx = faiss_amd.GpuIndexIVFFlat(res, d, nlist, METRIC_L2)
    idx.train(xt)
    idx.add(xb)
    for c in range(1, 10):
        x = synthetic_more_device(dmap, 1000000, 1338 + c, dev)
        idx.add_ptr(1000000, x.data_ptr())
        del x
    idx.nprobe = 32
    extra = synthetic_more_device(dmap, 50000, 4242, dev)
    xq_dev = torch.from_numpy(xq).to(dev)
    Dd = torch.empty((nq, k), dtype=torch.float32, device=dev)
    Id = torch.empty((nq, k), dtype=torch.int64, device=dev)

    def search():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        idx.search_ptr(nq, xq_dev.data_ptr(), k, Dd.data_ptr(), Id.data_ptr())
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    search()
    steady = float(np.median([search() for _ in range(7)]))
    assert idx.scan_info()[1] == 2
    after_add = []
    for i in range(50):
        idx.add_ptr(1000, extra.data_ptr() + i * 1000 * d * 4)
        after_add.append(search())
        assert idx.scan_info()[1] == 2 and idx.last_scan_arith() == 0, "add #%d sent the search to another scan" % i
    med = float(np.median(after_add))
    print("search of 4096 queries at nb = 10M: %.3f ms steady, %.3f ms behind an add of 1000 rows" % (steady * 1e3, med * 1e3))
    assert med <= 1.25 * steady, (med, steady)
    D, I = Dd.cpu().numpy(), Id.cpu().numpy()
    idx.set_scan_mode(1)
    idx.search_ptr(nq, xq_dev.data_ptr(), k, Dd.data_ptr(), Id.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(I, Id.cpu().numpy()) and np.array_equal(D, Dd.cpu().numpy())


@pytest.mark.parametrize("metric", [METRIC_L2, METRIC_INNER_PRODUCT])
def test_two_copy_codebook_of_the_pq_sweeps_keeps_the_bits(res, metric):
    """The A/B knob of round 5's LDS experiment (PQ64 over d = 128: the sweeps' codebook twice in LDS with different code ->
    bank maps, the copy of every (row, sub-quantizer) chosen when the operand-major copy of the codes is written): measured
    slower and off by default, but a supported path -- same bits as the one-copy sweeps and the query-major scan, also
    through add() (the blocks add() touches are re-dealt) and with a selector."""
    d, nlist, M, k = 128, 16, 64, 60
    xt, xb, xq = synthetic_dataset(d, 4000, 50000, 500, seed=77)
    idx = _make(res, 1, d, nlist, M, metric)
    idx.train(xt)
    idx.add(xb[:30000])
    idx.nprobe = 6
    idx.set_scan_mode(1)
    Dr, Ir = idx.search(xq, k)
    idx.set_scan_mode(2)
    for two in (True, False, True):
        idx.set_lmf_two_copies(two)
        D, I = idx.search(xq, k)
        assert idx.scan_info()[1] == 2 and np.array_equal(I, Ir) and np.array_equal(D, Dr), "two copies %s" % two
    idx.add(xb[30000:30007])
    idx.add(xb[30007:])
    D, I = idx.search(xq, k)
    idx.set_scan_mode(1)
    Dr, Ir = idx.search(xq, k)
    assert np.array_equal(I, Ir) and np.array_equal(D, Dr)
    idx.set_lmf_two_copies(False)


# ------------------------------------------------------------------ scalar quantizer behind the f16 filter
from faiss_amd import ScalarQuantizer as SQ  # noqa: E402


@pytest.mark.parametrize("qtype,metric,by_residual,d", [
    (SQ.QT_8bit, METRIC_L2, True, 128), (SQ.QT_8bit, METRIC_INNER_PRODUCT, True, 128), (SQ.QT_8bit, METRIC_L2, False, 96),
    (SQ.QT_4bit, METRIC_L2, True, 64), (SQ.QT_6bit, METRIC_INNER_PRODUCT, False, 72), (SQ.QT_fp16, METRIC_L2, True, 128),
    (SQ.QT_fp16, METRIC_INNER_PRODUCT, True, 40), (SQ.QT_8bit_uniform, METRIC_L2, True, 256), (SQ.QT_8bit, METRIC_L2, True, 500),
    (SQ.QT_8bit_direct, METRIC_L2, False, 32),
])
def test_scalar_quantizer_results_do_not_depend_on_the_batch_size(res, qtype, metric, by_residual, d):
    """the scalar quantizer's list-major scan summed in its own order (f32 MFMA chains), so a query's bits
    depended on the batch it arrived in.  Behind the f16 filter (the IVFFlat sweeps over an fp16 copy of the centred codes +
    the exact rerank with ivfsq_fused_kernel's arithmetic) a large batch returns what the same queries return one batch at a
    time, for every code type, both metrics, with and without residual encoding, d up to 512 -- also with an IDSelector and
    after incremental adds."""
    nlist, k = 32, 40
    xt, xb, xq = synthetic_dataset(d, 4000, 50000, 1500, seed=d + qtype)
    if qtype == SQ.QT_8bit_direct:
        sc = 255.0 / max(xt.max(), xb.max(), xq.max())
        xt, xb, xq = (np.floor(np.abs(v) * sc).astype(np.float32) for v in (xt, xb, xq))
    idx = faiss_amd.GpuIndexIVFScalarQuantizer(res, d, nlist, qtype, metric, by_residual)
    idx.train(xt)
    idx.add(xb[:40000])
    idx.nprobe = 8
    idx.set_scan_mode(idx.SCAN_LIST_MAJOR)
    D, I = idx.search(xq, k)
    assert idx.scan_info()[1] == 2 and idx.last_scan_arith() == 0
    idx.set_scan_mode(idx.SCAN_QUERY_MAJOR)
    for lo, hi in ((0, 30), (30, 31), (700, 1500)):
        D1, I1 = idx.search(xq[lo:hi], k)
        assert np.array_equal(D1, D[lo:hi]) and np.array_equal(I1, I[lo:hi]), (lo, hi)
    # incremental adds keep the sweeps' copy alive; a selector search through the sweeps == through the query-major scan
    idx.add(xb[40000:40003])
    idx.add(xb[40003:])
    sel = faiss_amd.IDSelectorRange(1000, 30000)
    idx.set_scan_mode(idx.SCAN_LIST_MAJOR)
    Ds, Is = idx.search(xq, k, params=faiss_amd.SearchParametersIVF(sel=sel))
    assert idx.scan_info()[1] == 2
    idx.set_scan_mode(idx.SCAN_QUERY_MAJOR)
    Dq, Iq = idx.search(xq, k, params=faiss_amd.SearchParametersIVF(sel=sel))
    assert np.array_equal(Ds, Dq) and np.array_equal(Is, Iq)


@pytest.mark.parametrize("qtype,metric,by_residual,d,scale", [
    (SQ.QT_8bit, METRIC_L2, True, 128, 1.0), (SQ.QT_8bit, METRIC_INNER_PRODUCT, True, 128, 1.0),
    (SQ.QT_8bit, METRIC_L2, False, 64, 1.0), (SQ.QT_8bit, METRIC_L2, False, 64, -1.0),   # scale < 0: every coordinate + 20
    (SQ.QT_4bit, METRIC_L2, True, 64, 1.0), (SQ.QT_6bit, METRIC_INNER_PRODUCT, False, 72, 1.0),
    (SQ.QT_fp16, METRIC_L2, True, 128, 1.0), (SQ.QT_fp16, METRIC_INNER_PRODUCT, False, 128, 1.0),
    (SQ.QT_8bit, METRIC_L2, True, 128, 300.0), (SQ.QT_8bit, METRIC_L2, True, 256, 1.0),
])
def test_scalar_quantizer_filter_error_bound_holds(res, qtype, metric, by_residual, d, scale):
    """|estimate - exact| <= E_q for every probed row (the superset argument): the estimates of the sweeps (test hook) against
    the distances the query-major scan returns for the same rows (k = all rows a query probes), with the band the bound
    kernel grants -- also for data far from the origin (the offsets b' are then large against the distances) and large
    values."""
    nlist, nb, nq, nprobe = 16, 4000, 64, 3
    xt, xb, xq = synthetic_dataset(d, 3000, nb, nq, seed=d + qtype)
    if scale < 0:
        xt, xb, xq = xt + np.float32(20.0), xb + np.float32(20.0), xq + np.float32(20.0)
        scale = 1.0
    xt, xb, xq = xt * np.float32(scale), xb * np.float32(scale), xq * np.float32(scale)
    idx = faiss_amd.GpuIndexIVFScalarQuantizer(res, d, nlist, qtype, metric, by_residual)
    idx.train(xt)
    idx.add(xb)
    idx.nprobe = nprobe
    Dq, Iq = idx.quantizer_search(xq, nprobe)
    sizes = np.array([idx.get_list_size(l) for l in range(nlist)])
    lids = [idx.get_list_ids(l) for l in range(nlist)]
    rows = np.array([int(sum(sizes[l] for l in Iq[q] if l >= 0)) for q in range(nq)])
    assert rows.max() <= 2048
    stride = int(nprobe * sizes.max())
    est, band = idx.filter_dump(xq, 10, stride)
    assert np.isfinite(band).all() and (band > 0).all()
    idx.set_scan_mode(idx.SCAN_QUERY_MAJOR)
    De, Ie = idx.search(xq, int(rows.max()))
    worst = 0.0
    for q in range(nq):
        pos_ids = np.concatenate([lids[l] for l in Iq[q] if l >= 0])
        assert len(pos_ids) == rows[q]
        order = {int(i): r for r, i in enumerate(Ie[q, :rows[q]])}
        exact = np.array([De[q, order[int(i)]] for i in pos_ids], dtype=np.float64)
        err = np.abs(est[q, :rows[q]].astype(np.float64) - exact)
        assert (err <= band[q]).all(), (q, float(err.max()), float(band[q]))
        worst = max(worst, float(err.max() / band[q]))
    print("qtype %d metric %d d %d scale %g: worst |estimate - exact| / band = %.5f (band %.3g, distances ~ %.3g)" % (
        qtype, metric, d, scale, worst, float(band.mean()), float(np.abs(De[:, 0]).mean())))
    assert worst > 1e-6


@pytest.mark.parametrize("metric,d,M", [
    (METRIC_L2, 256, 128),            # PQ128 over d = 256 (dsub 2)
    (METRIC_L2, 192, 48),             # dsub 4, d = 192: rows padded to 256 halfs
    (METRIC_INNER_PRODUCT, 256, 32),  # dsub 8
    (METRIC_L2, 96, 32),              # dsub 3: a shape the codebook sweeps refuse although d <= 128
    (METRIC_L2, 120, 24),             # dsub 5, d not a multiple of 16
    (METRIC_INNER_PRODUCT, 384, 64),  # dsub 6, three groups of eight k-steps
])
def test_ivfpq_beyond_the_codebook_kernel_through_decoded_residuals(res, metric, d, M):
    """IVFPQ shapes the LDS-codebook sweeps do not serve (d > 128, d not a multiple of 16, dsub 3 / 5 / 6) take the filter path
    over an fp16 copy of the DECODED residuals (the pair-operand IVFFlat sweeps): same bits as the query-major scan, whatever
    the batch, through incremental adds and with a selector; the error band holds row by row."""
    nlist, k = 16, 40
    xt, xb, xq = synthetic_dataset(d, 6000, 40000, 900, seed=d + M)
    idx = faiss_amd.GpuIndexIVFPQ(res, d, nlist, M, 8, metric)
    idx.train(xt)
    idx.add(xb[:30000])
    idx.nprobe = 5
    idx.set_scan_mode(idx.SCAN_QUERY_MAJOR)
    Dr, Ir = idx.search(xq, k)
    idx.set_scan_mode(idx.SCAN_LIST_MAJOR)
    D, I = idx.search(xq, k)
    assert idx.scan_info()[1] == 2 and idx.last_scan_arith() == 0
    assert np.array_equal(I, Ir) and np.array_equal(D, Dr)
    D1, I1 = idx.search(xq[40:77], k)
    assert np.array_equal(D1, D[40:77]) and np.array_equal(I1, I[40:77])
    lists, shadow = idx.resident_bytes()
    assert shadow >= 30000 * 2 * d  # 2 d bytes per row (rows padded to whole groups of k-steps beyond d = 128)
    idx.add(xb[30000:30011])
    idx.add(xb[30011:])
    sel = faiss_amd.IDSelectorRange(500, 25000)
    Ds, Is = idx.search(xq, k, params=faiss_amd.SearchParametersIVF(sel=sel))
    idx.set_scan_mode(idx.SCAN_QUERY_MAJOR)
    Dq, Iq = idx.search(xq, k, params=faiss_amd.SearchParametersIVF(sel=sel))
    assert np.array_equal(Ds, Dq) and np.array_equal(Is, Iq)
    # the estimates of the sweeps against the exact distances, row by row (a small index: k = every probed row)
    small = faiss_amd.GpuIndexIVFPQ(res, d, nlist, M, 8, metric)
    small.copy_centroids(idx.get_centroids())
    small.copy_pq_centroids(idx.get_pq_centroids())
    small.add(xb[:4000])
    small.nprobe = 3
    nq = 48
    Dc, Ic = small.quantizer_search(xq[:nq], 3)
    sizes = np.array([small.get_list_size(l) for l in range(nlist)])
    lids = [small.get_list_ids(l) for l in range(nlist)]
    rows = np.array([int(sum(sizes[l] for l in Ic[q] if l >= 0)) for q in range(nq)])
    assert rows.max() <= 2048
    est, band = small.filter_dump(xq[:nq], 10, int(3 * sizes.max()))
    small.set_scan_mode(small.SCAN_QUERY_MAJOR)
    De, Ie = small.search(xq[:nq], int(rows.max()))
    worst = 0.0
    for q in range(nq):
        pos_ids = np.concatenate([lids[l] for l in Ic[q] if l >= 0])
        order = {int(i): r for r, i in enumerate(Ie[q, :rows[q]])}
        exact = np.array([De[q, order[int(i)]] for i in pos_ids], dtype=np.float64)
        err = np.abs(est[q, :rows[q]].astype(np.float64) - exact)
        assert (err <= band[q]).all(), (q, float(err.max()), float(band[q]))
        worst = max(worst, float(err.max() / band[q]))
    print("IVFPQ decoded d %d M %d metric %d: worst |estimate - exact| / band = %.4f" % (d, M, metric, worst))
