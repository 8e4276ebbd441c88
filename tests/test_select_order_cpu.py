"""The output-order rule of the small-segment selections (select_kernels.hip small_select_kernel, lmf_select.h lmf_select_tail).

Winners sit in slots ordered by (distance key, scan position); the result rows are ordered by (distance key, label), ties of
both by slot.  Round 5 replaced the all-pairs ranking of the winners by a walk over the run of EQUAL distance keys around a
slot.  This test pins the equivalence of the two rules on a plain-Python model, with heavy ties (the case the GPU tests reach
only through duplicated database rows)."""
import numpy as np


def order_all_pairs(wk, wl):
    n = len(wk)
    out = [None] * n
    for t in range(n):
        r = 0
        for j in range(n):
            if wk[j] < wk[t] or (wk[j] == wk[t] and (wl[j] < wl[t] or (wl[j] == wl[t] and j < t))):
                r += 1
        out[r] = (wk[t], wl[t])
    return out


def order_tie_run(wk, wl):
    n = len(wk)
    out = [None] * n
    for t in range(n):
        lo, hi = t, t + 1
        while lo > 0 and wk[lo - 1] == wk[t]:
            lo -= 1
        while hi < n and wk[hi] == wk[t]:
            hi += 1
        r = lo
        for j in range(lo, hi):
            if wl[j] < wl[t] or (wl[j] == wl[t] and j < t):
                r += 1
        out[r] = (wk[t], wl[t])
    return out


def test_tie_run_rule_equals_all_pairs_ranking():
    rng = np.random.RandomState(11)
    for trial in range(300):
        n = int(rng.randint(1, 130))
        nkeys = int(rng.choice([1, 2, 5, n, 4 * n]))  # from "all distances equal" to "no ties"
        wk = np.sort(rng.randint(0, nkeys, n)).tolist()  # slots are in distance order (positions break the ties)
        wl = rng.randint(0, int(rng.choice([1, 3, 1000])), n).tolist()  # labels, duplicates included
        a, b = order_all_pairs(wk, wl), order_tie_run(wk, wl)
        assert a == b, (trial, wk, wl)
        assert all(x is not None for x in a) and a == sorted(a)  # a permutation, ordered by (key, label)
