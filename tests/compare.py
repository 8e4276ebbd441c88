"""tests/compare.py -- result comparison for the parity tests.

Modelled on the reference's compareLists (faiss/gpu/test/TestUtils.cpp:158-296) but stricter,
as the north star asks: labels must be IDENTICAL position by position; the only tolerated
differences are permutations / boundary swaps inside groups of results whose reference
distances are equal to within `tie_rtol` (summation-order noise between MKL/AVX2 and our fixed
fmaf chain), and every such case is counted and reported.  Distances must agree to `rtol`
relative.  With exact=True nothing at all may differ (bitwise distances, identical labels).
"""
import numpy as np


def check_knn(D, I, Dref, Iref, rtol=1e-4, tie_rtol=2e-5, exact=False, max_tie_frac=0.01, name=""):
    D, I, Dref, Iref = map(np.asarray, (D, I, Dref, Iref))
    assert D.shape == Dref.shape and I.shape == Iref.shape, (name, D.shape, Dref.shape)
    n, k = I.shape
    stats = dict(n=n, k=k, label_mismatch=0, tie_swaps=0, boundary_ties=0, max_rel_err=0.0)
    if n == 0:
        return stats
    if exact:
        assert np.array_equal(I, Iref), "%s: labels differ at %d positions" % (name, int((I != Iref).sum()))
        same = (D == Dref) | (np.isnan(D) & np.isnan(Dref))
        assert same.all(), "%s: distances not bit-identical at %d positions (max abs diff %g)" % (
            name, int((~same).sum()), float(np.nanmax(np.abs(D - Dref))))
        return stats
    # padding must coincide (TestUtils.cpp:294-296)
    assert np.array_equal(I == -1, Iref == -1), "%s: -1 padding differs" % name
    valid = Iref != -1
    scale = np.maximum(np.abs(Dref), 1e-30)
    rel = np.where(valid, np.abs(D - Dref) / scale, 0.0)
    # tiny absolute distances (near-duplicates) are compared absolutely against the row scale
    row_scale = np.maximum(np.max(np.abs(np.where(valid, Dref, 0)), axis=1, keepdims=True), 1e-30)
    rel = np.minimum(rel, np.where(valid, np.abs(D - Dref) / row_scale * 1e2, 0.0))
    stats["max_rel_err"] = float(rel.max())
    assert stats["max_rel_err"] <= rtol, "%s: distance error %g > %g" % (name, stats["max_rel_err"], rtol)
    bad_rows = np.nonzero((I != Iref).any(axis=1))[0]
    for r in bad_rows:
        ref_pos = {int(l): j for j, l in enumerate(Iref[r]) if l != -1}
        for j in np.nonzero(I[r] != Iref[r])[0]:
            stats["label_mismatch"] += 1
            lab = int(I[r, j])
            tol = tie_rtol * max(abs(float(Dref[r, j])), float(row_scale[r, 0]) * 1e-3)
            if lab in ref_pos:
                # same label at another rank: must be a near-tie permutation
                j2 = ref_pos[lab]
                assert abs(float(Dref[r, j2]) - float(Dref[r, j])) <= tol, (
                    "%s: row %d label %d at rank %d vs ref rank %d, ref distances %g / %g differ"
                    % (name, r, lab, j, j2, Dref[r, j], Dref[r, j2]))
                stats["tie_swaps"] += 1
            else:
                # label absent from the reference list: only legal as a tie at the k-th boundary
                kth = float(Dref[r][valid[r]][-1])
                assert abs(float(D[r, j]) - kth) <= tol, (
                    "%s: row %d rank %d label %d (dist %g) not in reference list and not a boundary tie (kth %g)"
                    % (name, r, j, lab, D[r, j], kth))
                stats["boundary_ties"] += 1
    frac = stats["label_mismatch"] / float(n * k)
    assert frac <= max_tie_frac, "%s: %.4f of labels sit in near-tie groups (> %.4f)" % (name, frac, max_tie_frac)
    return stats


def recall_at(I, gt, r):
    """faiss's R@r: fraction of queries whose true nearest neighbour is within the first r results
    (benchs/datasets.py:39-43)."""
    return float((I[:, :r] == gt[:, :1]).any(axis=1).mean())
