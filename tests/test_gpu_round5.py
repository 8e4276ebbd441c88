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
