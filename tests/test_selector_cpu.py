"""CPU tests of the IDSelector mirror (faiss/impl/IDSelector.h): host-side membership of every selector type against a
numpy restatement, and -- where oracle/_ref is present -- the restatement and the "search with a selector == search of
the selected subset" rule the GPU tests rely on against the REAL reference (IndexFlat::search with
SearchParameters::sel, faiss/IndexFlat.cpp:36-58; IndexIVF::search, faiss/IndexIVF.cpp scan_codes)."""
import numpy as np
import pytest

from compare import check_knn
from oracle.pyoracle import METRIC_INNER_PRODUCT, METRIC_L2, Oracle, Ref, synthetic_dataset
from selector_cases import filter_lists, selector_cases


def test_selector_membership_matches_restatement():
    lo, hi = 1000, 9000
    probe = np.concatenate([np.arange(lo - 50, hi + 50), [-1, -9, 2**40, -2**40, 2**62]]).astype(np.int64)
    for case in selector_cases(lo, hi, seed=2):
        want = case["member"](probe)
        got = np.array([case["sel"].is_member(int(i)) for i in probe])
        assert np.array_equal(got, want), case["name"]


def subset_flat(metric, xb, xq, k, keep):
    rows = np.nonzero(keep)[0]
    D, I = Oracle.flat_search(metric, xb[rows], xq, k)
    return D, np.where(I >= 0, rows[np.maximum(I, 0)] if len(rows) else -1, -1)


@pytest.mark.skipif(not Ref.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("metric", [METRIC_L2, METRIC_INNER_PRODUCT])
def test_reference_flat_selector_is_subset_search(metric):
    d, nb, nq, k = 32, 6000, 40, 100
    _, xb, xq = synthetic_dataset(d, 0, nb, nq, seed=9)
    ref = Ref.index_factory(d, "Flat", metric)
    ref.add(xb)
    labels = np.arange(nb, dtype=np.int64)
    for case in selector_cases(0, nb, seed=4):
        if case["ref"] is None:
            continue
        Dr, Ir = ref.search_sel(xq, k, **case["ref"])
        Do, Io = subset_flat(metric, xb, xq, k, case["member"](labels))
        check_knn(Do, Io, Dr, Ir, rtol=1e-4, name="flat selector " + case["name"])


@pytest.mark.skipif(not Ref.available(), reason="oracle/_ref not built")
def test_reference_ivf_selector_is_filtered_lists():
    d, nb, nq, k, nlist, nprobe = 32, 8000, 40, 50, 16, 4
    xt, xb, xq = synthetic_dataset(d, 2000, nb, nq, seed=10)
    ids = (np.random.RandomState(1).permutation(nb).astype(np.int64) * 5 + 11)
    ref = Ref.index_factory(d, "IVF16,Flat", METRIC_L2)
    ref.set_train_niter(4)
    ref.train(xt)
    ref.add_with_ids(xb, ids)
    cent = ref.centroids()
    sizes, codes, lids = ref.lists()
    for case in selector_cases(11, 11 + 5 * nb, seed=5):
        if case["ref"] is None:
            continue
        Dr, Ir = ref.search_sel(xq, k, nprobe=nprobe, **case["ref"])
        s2, c2, i2 = filter_lists(sizes, codes, lids, case["member"](lids))
        Do, Io, _, _ = Oracle.ivf_search(0, METRIC_L2, cent, s2, c2, i2, xq, nprobe, k)
        check_knn(Do, Io, Dr, Ir, rtol=1e-4, name="ivf selector " + case["name"])
