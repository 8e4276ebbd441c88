"""CPU check of the candidate rule of faiss_amd/csrc/flat_small.hip (one-launch search of a small database, the IVF coarse
quantizer): a numpy restatement of the kernel's bookkeeping -- fp16-rounded scores, 32 groups of disjoint rows (wavefront,
lane half, fragment quarter), bound = minimum of the group maxima, band of 2 e_q below it and below the k-th best score --
must keep the exact top-k inside the candidate set, whatever the data.  (The kernel itself has not run on hardware yet;
this pins the rule it implements.)"""
import numpy as np
import pytest

from oracle.pyoracle import synthetic_dataset


def err_bound(metric_l2, d, xn, yn_max):
    """flat_filter_err_bound (faiss_amd/csrc/kernels.h), exact_inputs = false"""
    nq, ny = np.sqrt(xn), np.sqrt(yn_max)
    e = (9.775e-4 + 2.4e-7 * d) * nq * ny + 3.0e-8 * np.sqrt(d) * (nq + ny) + 1e-30
    if metric_l2:
        e = e + 6.0e-8 * (xn + yn_max) + 6.0e-8 * d * yn_max
    return 1.25 * e


def band(tk, e):
    return tk - 2 * e - 9.6e-7 * np.abs(tk) - 1e-37


@pytest.mark.parametrize("metric_l2", [True, False])
@pytest.mark.parametrize("nb,k,scale", [(4096, 32, 1.0), (2048, 32, 30.0), (8192, 10, 1e-2), (5000, 1, 1.0)])
def test_small_database_candidate_rule_keeps_the_exact_topk(metric_l2, nb, k, scale):
    d, nq = 128, 300
    _, xb, xq = synthetic_dataset(d, 0, nb, nq, seed=nb + k)
    xb, xq = (xb * scale).astype(np.float32), (xq * scale).astype(np.float32)
    yn = (xb.astype(np.float64) ** 2).sum(1)
    bias = (-0.5 * yn if metric_l2 else np.zeros(nb)).astype(np.float32)
    t = xq.astype(np.float16).astype(np.float32) @ xb.astype(np.float16).astype(np.float32).T + bias[None, :]
    exact = xq.astype(np.float64) @ xb.astype(np.float64).T + (-0.5 * yn if metric_l2 else 0.0)
    # the kernel's groups: row r sits in 32-row block r // 32 (wavefront (r // 32) % 4), position p = r % 32 of the MFMA
    # fragment = (quarter g = p // 8, lane half h = (p % 8) // 4, element p % 4)
    r = np.arange(nb)
    group = (((r // 32) % 4) * 2 + (r % 8) // 4) * 4 + (r % 32) // 8
    gm = np.stack([t[:, group == g].max(1) for g in range(32)], axis=1)
    tk_lower = gm.min(1)
    kth = -np.sort(-t, axis=1)[:, k - 1]
    assert (tk_lower <= kth).all(), "the minimum of 32 maxima of disjoint row sets bounds the 32nd best score from below"
    e = err_bound(metric_l2, d, (xq.astype(np.float64) ** 2).sum(1), yn.max())
    cand = t > band(tk_lower, e)[:, None]
    inside = t > band(kth, e)[:, None]
    assert (inside <= cand).all(), "the band below the k-th best score lies inside the candidate set"
    top = np.argsort(-exact, axis=1, kind="stable")[:, :k]
    assert np.take_along_axis(inside, top, axis=1).all(), "an exact top-k row fell outside the band"
    assert cand.sum(1).max() <= 512 or scale != 1.0  # (capacity of the kernel's candidate lists at the data's own scale)
