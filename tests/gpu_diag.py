"""tests/gpu_diag.py -- one-shot GPU diagnostic: runs every kernel family against the oracle and
keeps going after failures, so one gpurun call reports as much as possible.
Usage (GPU box): python tests/gpu_diag.py [--big]
"""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import faiss_amd  # noqa: E402
from compare import check_knn  # noqa: E402
from oracle.pyoracle import (METRIC_INNER_PRODUCT, METRIC_L2, Oracle, Ref, integer_dataset,  # noqa: E402
                             synthetic_dataset)

RESULTS = []


def step(name):
    def deco(fn):
        t = time.time()
        try:
            msg = fn()
            RESULTS.append((name, "OK", msg))
            print("[OK  ] %-40s %6.2fs %s" % (name, time.time() - t, msg or ""), flush=True)
        except Exception as e:  # noqa: BLE001
            RESULTS.append((name, "FAIL", repr(e)))
            print("[FAIL] %-40s %6.2fs %s" % (name, time.time() - t, repr(e)[:600]), flush=True)
            traceback.print_exc(limit=3)
        return fn
    return deco


def describe_mismatch(A, B):
    bad = np.argwhere(A != B)
    s = "mismatch %d/%d maxabs %g" % (len(bad), A.size, float(np.nanmax(np.abs(A - B))))
    for (i, j) in bad[:6]:
        s += " [%d,%d]: %r vs %r;" % (i, j, float(A[i, j]), float(B[i, j]))
    return s


def main():
    big = "--big" in sys.argv
    print("gpus:", faiss_amd.get_num_gpus(), flush=True)
    res = faiss_amd.StandardGpuResources(0)

    for metric, mname in ((METRIC_L2, "L2"), (METRIC_INNER_PRODUCT, "IP")):
        for (d, nb, nq) in ((128, 300, 70), (40, 1000, 33), (264, 200, 10)):
            @step("pairwise MFMA %s d=%d nb=%d nq=%d" % (mname, d, nb, nq))
            def _():
                _, xb, xq = synthetic_dataset(d, 0, nb, nq, seed=d)
                idx = faiss_amd.GpuIndexFlat(res, d, metric)
                idx.add(xb)
                G = idx.pairwise_distances(xq)
                Oracle.set_pair_order(0)
                O = Oracle.pairwise(metric, xb, xq)
                if np.array_equal(G, O):
                    return "bit-exact"
                Oracle.set_pair_order(1)
                O1 = Oracle.pairwise(metric, xb, xq)
                Oracle.set_pair_order(0)
                if np.array_equal(G, O1):
                    raise AssertionError("bit-exact only with SWAPPED pair order")
                raise AssertionError(describe_mismatch(G, O))

    for simple in (True, False):
        for metric, mname in ((METRIC_L2, "L2"), (METRIC_INNER_PRODUCT, "IP")):
            for (d, nb, nq, k) in ((128, 5000, 300, 10), (128, 5000, 300, 1), (128, 20000, 64, 100),
                                   (40, 3000, 50, 7), (64, 50, 5, 8), (128, 3000, 1, 16),
                                   (128, 40000, 513, 128), (32, 9000, 100, 2048)):
                @step("flat %s %s d=%d nb=%d nq=%d k=%d" % ("simple" if simple else "MFMA", mname, d, nb, nq, k))
                def _():
                    _, xb, xq = synthetic_dataset(d, 0, nb, nq, seed=nb + k)
                    idx = faiss_amd.GpuIndexFlat(res, d, metric)
                    idx.set_use_simple_kernel(simple)
                    idx.add(xb)
                    D, I = idx.search(xq, k)
                    Do, Io = Oracle.flat_search(metric, xb, xq, k)
                    check_knn(D, I, Do, Io, exact=True, name="flat")
                    return "bit-exact"

    @step("flat MFMA integer ties")
    def _():
        xb, xq = integer_dataset(32, 20000, 300, seed=5, hi=8)
        idx = faiss_amd.GpuIndexFlatL2(res, 32)
        idx.add(xb)
        for k in (1, 10, 100, 300):
            D, I = idx.search(xq, k)
            Do, Io = Oracle.flat_search(METRIC_L2, xb, xq, k)
            check_knn(D, I, Do, Io, exact=True, name="ties k=%d" % k)
        return "bit-exact incl. ties"

    import test_oracle_cpu as T
    for name in ("ivfflat_l2", "ivfflat_ip", "ivfpq_l2", "ivfpq_ip"):
        @step("ivf copyFrom + search " + name)
        def _():
            c = T.load_ivf_case(name)
            z = c["z"]
            d = c["xb"].shape[1]
            nlist = z["centroids"].shape[0]
            if c["kind"] == 0:
                idx = faiss_amd.GpuIndexIVFFlat(res, d, nlist, c["metric"])
            else:
                idx = faiss_amd.GpuIndexIVFPQ(res, d, nlist, c["M"], 8, c["metric"])
                idx.copy_pq_centroids(c["pq"])
            idx.copy_centroids(z["centroids"])
            idx.copy_lists(z["list_sizes"], c["codes"], z["list_ids"])
            idx.nprobe = c["nprobe"]
            D, I = idx.search(c["xq"], c["k"])
            Do, Io, _, _ = Oracle.ivf_search(c["kind"], c["metric"], z["centroids"], z["list_sizes"], c["codes"],
                                             z["list_ids"], c["xq"], c["nprobe"], c["k"], M=c["M"], pq=c["pq"])
            check_knn(D, I, Do, Io, exact=True, name=name + " vs oracle")
            st = check_knn(D, I, z["D"], z["I"], rtol=1e-4, name=name + " vs golden")
            return "oracle bit-exact; golden max_rel %.2e ties %d" % (st["max_rel_err"], st["label_mismatch"])

        @step("ivf add path " + name)
        def _():
            c = T.load_ivf_case(name)
            z = c["z"]
            d = c["xb"].shape[1]
            nlist = z["centroids"].shape[0]
            if c["kind"] == 0:
                idx = faiss_amd.GpuIndexIVFFlat(res, d, nlist, c["metric"])
            else:
                idx = faiss_amd.GpuIndexIVFPQ(res, d, nlist, c["M"], 8, c["metric"])
                idx.copy_pq_centroids(c["pq"])
            idx.copy_centroids(z["centroids"])
            h = len(c["xb"]) // 3
            idx.add_with_ids(c["xb"][:h], c["ids"][:h])   # two adds: exercises the arena rebuild
            idx.add_with_ids(c["xb"][h:], c["ids"][h:])
            sizes, codes, lids, _ = Oracle.build_ivf_lists(c["kind"], c["metric"], z["centroids"], c["xb"],
                                                           ids=c["ids"], pq=c["pq"])
            gs = np.array([idx.get_list_size(l) for l in range(nlist)], dtype=np.uint32)
            assert np.array_equal(gs, sizes), "list sizes differ"
            gi = np.concatenate([idx.get_list_ids(l) for l in range(nlist)])
            assert np.array_equal(gi, lids), "list ids differ"
            gc = np.concatenate([idx.get_list_codes(l) for l in range(nlist)])
            assert np.array_equal(gc.reshape(-1), codes.reshape(-1)), "codes differ: %.5f equal" % (
                (gc.reshape(-1) == codes.reshape(-1)).mean())
            return "lists/ids/codes identical to oracle"

    @step("k-means objective vs reference")
    def _():
        xt, _, _ = synthetic_dataset(32, 20000, 0, 0, seed=3)
        cent, obj = faiss_amd.kmeans(res, xt, 64, niter=10, seed=1)
        o = Oracle.kmeans_objective(xt, cent)
        msg = "obj %g (oracle recompute of final centroids %g)" % (obj[-1], o)
        if Ref.available():
            _, robj = Ref.kmeans(xt, 64, niter=10, seed=1)
            msg += " ref %g ratio %.4f" % (robj, obj[-1] / robj)
            assert abs(obj[-1] / robj - 1) < 0.05
        return msg

    @step("native train+add+search IVFPQ recall")
    def _():
        xt, xb, xq = synthetic_dataset(64, 20000, 100000, 500, seed=4)
        idx = faiss_amd.GpuIndexIVFPQ(res, 64, 256, 16, 8, METRIC_L2)
        idx.train(xt)
        idx.add(xb)
        idx.nprobe = 16
        D, I = idx.search(xq, 10)
        flat = faiss_amd.GpuIndexFlatL2(res, 64)
        flat.add(xb)
        _, gt = flat.search(xq, 1)
        r1 = float((I[:, :1] == gt).mean())
        r10 = float((I == gt).any(axis=1).mean())
        assert r10 > 0.5, (r1, r10)
        return "R@1 %.3f R@10 %.3f" % (r1, r10)

    if big:
        @step("C2 timing: flat nb=1M nq=10k k=100 d=128")
        def _():
            rs = np.random.RandomState(0)
            _, xb, xq = synthetic_dataset(128, 0, 1000000, 10000, seed=1338)
            idx = faiss_amd.GpuIndexFlatL2(res, 128)
            t = time.time(); idx.add(xb); ta = time.time() - t
            D, I = idx.search(xq, 100)  # warm-up
            res.profile_enable(True)
            res.profile_reset()
            t = time.time(); D, I = idx.search(xq, 100); ts = time.time() - t
            scan = res.profile_get("flat_scan_kernel")
            sel = res.profile_get("select_k_kernel")
            res.profile_enable(False)
            np.save(os.path.join(ROOT, "gpurun_out", "c2_I_head.npy"), I[:64])
            flops = 2.0 * 10000 * 1e6 * 128
            return "add %.2fs search %.4fs (%.0f QPS host-to-host) scan %.3f ms select %.3f ms => %.1f TF" % (
                ta, ts, 10000 / ts, scan[0], sel[0], flops / (scan[0] * 1e-3) / 1e12)

    nfail = sum(1 for r in RESULTS if r[1] != "OK")
    print("SUMMARY: %d steps, %d failed" % (len(RESULTS), nfail), flush=True)
    return 1 if nfail else 0


if __name__ == "__main__":
    sys.exit(main())
