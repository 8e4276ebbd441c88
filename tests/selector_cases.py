"""tests/selector_cases.py -- IDSelector test matrix shared by the CPU and GPU tests.

A case is (name, build) where build(ids_domain) returns (faiss_amd selector, numpy membership function); the membership
function restates faiss/impl/IDSelector.cpp in numpy (Range: imin <= id < imax; Batch / Array: id in the list;
Bitmap: id // 8 < n and bit id % 8 of byte id // 8, the id taken as an unsigned number; Not / And / Or / XOr).
`ref` describes the same selector for oracle/ref_shim.cpp ref_index_search_sel (kind, a, b, data, negate) or is None.
"""
import numpy as np

import faiss_amd


def _bitmap_member(bits, ids):
    u = ids.astype(np.uint64)
    byte = (u >> np.uint64(3))
    ok = byte < np.uint64(bits.size)
    b = bits[np.minimum(byte, np.uint64(max(bits.size - 1, 0))).astype(np.int64)] if bits.size else np.zeros(len(ids), np.uint8)
    return ok & (((b >> (u & np.uint64(7)).astype(np.uint8)) & 1) != 0)


def selector_cases(lo, hi, seed=0):
    """Selectors over labels drawn from [lo, hi) (hi - lo >= 1000).  Returns a list of dicts: name, sel, member(ids)
    -> bool array, ref (arguments of RefIndex.search_sel or None)."""
    rs = np.random.RandomState(seed)
    span = hi - lo
    a, b = lo + span // 4, lo + (3 * span) // 4
    some = lo + rs.permutation(span)[: max(1, span * 3 // 10)]          # 30 % of the labels
    few = lo + rs.permutation(span)[:60]                                 # fewer than k = 100
    nbits = (lo + span * 9 // 10 + 7) // 8                               # labels beyond the bitmap are excluded
    bits = rs.randint(0, 256, size=nbits).astype(np.uint8)
    cases = []

    def add(name, sel, member, ref=None, keep=()):
        cases.append(dict(name=name, sel=sel, member=member, ref=ref, keep=keep))

    add("all", faiss_amd.IDSelectorAll(), lambda ids: np.ones(len(ids), bool))
    add("range", faiss_amd.IDSelectorRange(a, b), lambda ids: (ids >= a) & (ids < b), dict(kind=0, a=a, b=b))
    add("batch30", faiss_amd.IDSelectorBatch(some), lambda ids: np.isin(ids, some), dict(kind=1, data=some))
    add("array_few", faiss_amd.IDSelectorArray(few), lambda ids: np.isin(ids, few), dict(kind=2, data=few))
    add("bitmap", faiss_amd.IDSelectorBitmap(bits), lambda ids: _bitmap_member(bits, ids), dict(kind=3, data=bits))
    s_range, s_batch, s_bm = cases[1]["sel"], cases[2]["sel"], cases[4]["sel"]
    add("not_batch", faiss_amd.IDSelectorNot(s_batch), lambda ids: ~np.isin(ids, some), dict(kind=1, data=some, negate=True),
        keep=(s_batch,))
    add("range_and_not_batch", s_range & ~s_batch, lambda ids: (ids >= a) & (ids < b) & ~np.isin(ids, some),
        dict(kind=4, a=a, b=b, data=some))
    add("range_xor_bitmap", s_range ^ s_bm, lambda ids: ((ids >= a) & (ids < b)) ^ _bitmap_member(bits, ids))
    add("batch_or_range", s_batch | s_range, lambda ids: np.isin(ids, some) | ((ids >= a) & (ids < b)))
    add("empty", faiss_amd.IDSelectorRange(lo, lo), lambda ids: np.zeros(len(ids), bool), dict(kind=0, a=lo, b=lo))
    return cases


def filter_lists(sizes, codes, ids, keep):
    """the inverted lists (sizes [nlist], codes / ids concatenated in list order) restricted to the entries with
    keep[entry] -- entries keep their order, so the scan order of the remaining ones is unchanged"""
    owner = np.repeat(np.arange(len(sizes)), sizes.astype(np.int64))
    new_sizes = np.bincount(owner[keep], minlength=len(sizes)).astype(np.uint32)
    return new_sizes, np.ascontiguousarray(codes[keep]), np.ascontiguousarray(ids[keep])
