// faiss_amd/csrc/lmf_select.h -- the k-selection a rerank workgroup of the list-major filter path runs on its own candidates
// (shared by ivf_lm_filter.hip: IVFFlat / IVFPQ, and ivf_fused.hip: scalar quantizer).
#pragma once
#include "kernels.h"

namespace faiss_amd {

// The k best of a query's re-derived candidates, by the workgroup that holds them (IvfLmParams::fin_dis): keys kl[0 .. n) in
// LDS (ordkey << 32 | scan position, all distinct), their probe numbers in cpr.  Winners = the k smallest keys (rank by
// counting: every thread reads the same kl[j], an LDS broadcast); labels from the stored ids; output order by (distance,
// label) among the winners, ties of both by their winner slot -- what select_k_kernel / wave_select_kernel produce.
// wk / wl: LDS room for kLmfFusedSelectK winners.  Called by all threads of the workgroup.
template <int THREADS>
__device__ __forceinline__ void lmf_select_tail(const IvfLmParams& p, int q, int n, const unsigned long long* kl, const uint16_t* cpr, uint32_t* wk,
                                                int64_t* wl) {
    const int tid = threadIdx.x, np = p.nprobe, k = p.k;
    __syncthreads();
    for (int i = tid; i < n; i += THREADS) {
        const unsigned long long ki = kl[i];
        int r = 0;
        for (int j = 0; j < n; ++j) r += kl[j] < ki ? 1 : 0;
        if (r < k) {
            const uint32_t pos = (uint32_t)ki;
            const int pr = (int)cpr[i];
            wk[r] = (uint32_t)(ki >> 32);
            wl[r] = p.arena_ids[p.row_base[(int64_t)q * np + pr] + (int64_t)pos];
        }
    }
    __syncthreads();
    const int nwin = min(n, k);
    const float pad = neutral_distance(p.metric);
    float* od = p.fin_dis + (int64_t)q * k;
    int64_t* oi = p.fin_ids + (int64_t)q * k;
    for (int i = tid; i < k; i += THREADS) {
        if (i >= nwin) { // fewer candidates than k: the tail is padding
            od[i] = pad;
            oi[i] = -1;
            continue;
        }
        // (the slots are in (distance, position) order: the winners with a smaller distance are the slots before the run of equal
        // distances around i, and only that run -- one slot unless distances tie -- is ranked by (label, slot))
        const uint32_t a = wk[i];
        const int64_t ia = wl[i];
        int lo = i, hi = i + 1;
        while (lo > 0 && wk[lo - 1] == a) --lo;
        while (hi < nwin && wk[hi] == a) ++hi;
        int r = lo;
        for (int j = lo; j < hi; ++j) {
            const int64_t ib = wl[j];
            r += (ib < ia || (ib == ia && j < i)) ? 1 : 0;
        }
        const bool real = a < kInvalidOrdKey;
        od[r] = real ? unordkey_rt(p.metric, a) : pad;
        oi[r] = real ? ia : -1;
    }
}


} // namespace faiss_amd
