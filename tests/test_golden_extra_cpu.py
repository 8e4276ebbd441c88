"""CPU tests against golden outputs of the REAL reference (tests/golden/make_golden.py extra): the "extra" metrics of
IndexFlat (faiss/utils/extra_distances.cpp -- what GpuIndexFlat / bfKnn run through faiss/gpu/impl/GeneralDistance.cuh)
and searches restricted by IDSelector objects.  They pin the oracle restatements (orc_flat_search_general; "selector
search == search of the selected subset") on machines without oracle/_ref."""
import os

import numpy as np
import pytest

from compare import check_knn
from oracle.pyoracle import METRIC_L2, Oracle, synthetic_dataset
from selector_cases import filter_lists, selector_cases
from test_selector_cpu import subset_flat

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
EXTRA_METRICS = [("L1", 2, 0.0), ("Linf", 3, 0.0), ("Lp3", 4, 3.0), ("Lp1_5", 4, 1.5), ("Canberra", 20, 0.0),
                 ("BrayCurtis", 21, 0.0), ("JensenShannon", 22, 0.0), ("Jaccard", 23, 0.0)]


def positive_dataset(d, nb, nq, seed):
    _, xb, xq = synthetic_dataset(d, 0, nb, nq, seed=seed)
    return (np.abs(xb) + 0.01).astype(np.float32), (np.abs(xq) + 0.01).astype(np.float32)


@pytest.mark.parametrize("tag,metric,arg", EXTRA_METRICS, ids=[m[0] for m in EXTRA_METRICS])
def test_extra_metric_oracle_vs_reference_golden(tag, metric, arg):
    z = np.load(os.path.join(GOLD, "flat_extra_metrics.npz"))
    d, nb, nq, seed = (int(v) for v in z["gen"])
    xb, xq = positive_dataset(d, nb, nq, seed)
    D, I = Oracle.flat_search_general(metric, xb, xq, int(z["k"]), metric_arg=arg)
    st = check_knn(D, I, z["D_" + tag], z["I_" + tag], rtol=1e-4, name="extra metric " + tag)
    assert st["max_rel_err"] < 2e-6  # (the reference sums L1 in a vectorised order, pow / log come from libm)
    if metric == 23:  # a similarity: best = largest first
        assert (np.diff(D, axis=1) <= 0).all()
    else:
        assert (np.diff(D, axis=1) >= 0).all()


def test_extra_metric_oracle_padding_and_empty():
    xb, xq = positive_dataset(8, 5, 3, 1)
    D, I = Oracle.flat_search_general(2, xb, xq, 8)
    assert (I[:, 5:] == -1).all() and (D[:, 5:] == np.finfo(np.float32).max).all() and (I[:, :5] >= 0).all()
    D, I = Oracle.flat_search_general(23, xb, xq, 8)  # Jaccard: padded like the inner product
    assert (I[:, 5:] == -1).all() and (D[:, 5:] == -np.finfo(np.float32).max).all()


def test_selector_subset_rule_vs_reference_golden():
    z = np.load(os.path.join(GOLD, "selectors.npz"))
    k = int(z["k"])
    d, nb, nq, seed = (int(v) for v in z["gen_flat"])
    _, xb, xq = synthetic_dataset(d, 0, nb, nq, seed=seed)
    labels = np.arange(nb, dtype=np.int64)
    n = 0
    for case in selector_cases(0, nb, seed=4):
        if "flatD_" + case["name"] in z:
            Do, Io = subset_flat(METRIC_L2, xb, xq, k, case["member"](labels))
            check_knn(Do, Io, z["flatD_" + case["name"]], z["flatI_" + case["name"]], rtol=1e-4,
                      name="flat selector " + case["name"])
            n += 1
    assert n >= 7
    d, nt, nbi, nq, seed = (int(v) for v in z["gen_ivf"])
    _, xbi, xqi = synthetic_dataset(d, nt, nbi, nq, seed=seed)
    ids = np.random.RandomState(1).permutation(nbi).astype(np.int64) * 5 + 11
    sizes, lids = z["list_sizes"], z["list_ids"]
    order = np.argsort(ids)
    rows = order[np.searchsorted(ids[order], lids)]
    assert np.array_equal(ids[rows], lids)
    codes = np.ascontiguousarray(xbi[rows]).view(np.uint8).reshape(len(rows), -1)
    n = 0
    for case in selector_cases(11, 11 + 5 * nbi, seed=5):
        if "ivfD_" + case["name"] in z:
            s2, c2, i2 = filter_lists(sizes, codes, lids, case["member"](lids))
            Do, Io, _, _ = Oracle.ivf_search(0, METRIC_L2, z["centroids"], s2, c2, i2, xqi, int(z["nprobe"]), k)
            check_knn(Do, Io, z["ivfD_" + case["name"]], z["ivfI_" + case["name"]], rtol=1e-4,
                      name="ivf selector " + case["name"])
            n += 1
    assert n >= 7


def test_add_core_lists_vs_reference_golden():
    """IndexIVF::add_core with a caller-supplied assignment (tests/golden/add_core.npz): entries in insertion order in the
    GIVEN lists, -1 left out but counted, PQ codes of the residual against the given list's centroid -- byte for byte what
    the oracle's encoder produces (the GPU test tests/test_gpu_add_core.py holds the device lists against the same oracle)."""
    z = np.load(os.path.join(GOLD, "add_core.npz"))
    d, nt, nb, nq, seed = (int(v) for v in z["gen"])
    _, xb, _ = synthetic_dataset(d, nt, nb, nq, seed=seed)
    assign, ids = z["assign"], z["ids"]
    ok = assign >= 0
    order = np.argsort(np.where(ok, assign, 1 << 30), kind="stable")[: int(ok.sum())]
    want_sizes = np.bincount(assign[ok], minlength=64).astype(np.uint32)
    for tag in ("flat", "pq"):
        assert int(z[tag + "_ntotal"]) == nb
        assert np.array_equal(z[tag + "_sizes"], want_sizes) and np.array_equal(z[tag + "_ids"], ids[order])
    codes = Oracle.pq_encode(z["pq_codebook"], z["pq_coarse"], xb, np.where(ok, assign, 0))
    assert np.array_equal(codes[order], z["pq_codes"])
