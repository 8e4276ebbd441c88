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


def kth_min_relative_radix(vals, k, invalid=0xFFFFFFFF):
    """lmf_bound_kernel / lmf_tighten_kernel (ivf_lm_filter.hip): the k-th smallest valid key through 8-bit digits of key - min over
    the bits in which the valid keys differ; fewer than k valid keys: the invalid key."""
    valid = [v for v in vals if v < invalid]
    if len(valid) < k:
        return invalid
    vmin, vmax = min(valid), max(valid)
    rng = vmax - vmin
    npass = 0 if rng == 0 else (rng.bit_length() + 7) // 8
    prefix, need = 0, k
    for p in range(npass - 1, -1, -1):
        hist = [0] * 256
        for v in valid:
            w = v - vmin
            if p == 3 or (w >> (8 * (p + 1))) == prefix:
                hist[(w >> (8 * p)) & 255] += 1
        before = 0
        for b in range(256):
            if before < need <= before + hist[b]:
                prefix, need = (prefix << 8) | b, need - before
                break
            before += hist[b]
    return vmin + prefix


def test_min_relative_radix_select_finds_the_kth_smallest():
    rng = np.random.RandomState(5)
    for trial in range(300):
        n = int(rng.randint(1, 600))
        spread = int(rng.choice([1, 2, 300, 70000, 1 << 24, 0xFFFFFFF0]))
        base = int(rng.randint(0, 0xFFFFFFFF - spread))
        vals = (base + rng.randint(0, spread, n).astype(np.int64)).tolist()
        for i in rng.choice(n, size=n // 5, replace=False):
            vals[i] = 0xFFFFFFFF  # empty granule slots
        valid = sorted(v for v in vals if v < 0xFFFFFFFF)
        for k in {1, max(1, len(valid) // 2), max(1, len(valid)), len(valid) + 1}:
            want = valid[k - 1] if k <= len(valid) else 0xFFFFFFFF
            assert kth_min_relative_radix(vals, k) == want, (trial, k)
