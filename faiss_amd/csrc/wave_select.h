// faiss_amd/csrc/wave_select.h -- wavefront-level (64 lanes, no workgroup barrier) exact selection
// over 64-bit keys held in global memory: the per-(query, split) reservoirs of the flat scans.
// 64-lane replacement for the reference's WarpSelect (faiss/gpu/utils/Select.cuh:337-560).
#pragma once
#include "common.h"

namespace faiss_amd {

typedef unsigned long long u64;

// ---------------------------------------------------------------------------------
// wave-level reservoir compaction (one 64-lane wavefront, no workgroup barriers)
// ---------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() {
    return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

__device__ __forceinline__ void wave_mem_sync() {
    // orders this wave's LDS/global accesses (s_waitcnt) and stops compiler reordering
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ u64 wave_max_u64(u64 v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        u64 o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

// k-th smallest (1-based, 1 <= k <= n) of n unique 64-bit keys in global memory.
// MSB-first radix select, 8 bits per pass, histogram in this wave's private LDS `hist[256]`.
__device__ inline u64 wave_select_kth(const u64* __restrict__ keys, int n, int k, unsigned* hist) {
    const int lane = lane_id();
    u64 prefix = 0, mask = 0;
    int need = k;
    for (int shift = 56; shift >= 0; shift -= 8) {
        // zero histogram
        *(uint4*)(hist + 4 * lane) = make_uint4(0, 0, 0, 0);
        wave_mem_sync();
        for (int i = lane; i < n; i += 64) {
            u64 key = keys[i];
            if ((key & mask) == prefix) {
                atomicAdd(&hist[(unsigned)(key >> shift) & 255u], 1u);
            }
        }
        wave_mem_sync();
        uint4 c;
        {
            lds_volatile_u32* hv = lds_volatile(hist + 4 * lane); // (typed to LDS: a generic volatile read is a FLAT load, common.h)
            c.x = hv[0]; c.y = hv[1]; c.z = hv[2]; c.w = hv[3];
        }
        unsigned s = c.x + c.y + c.z + c.w;
        const unsigned incl = wave_incl_scan(s);
        u64 ge = __ballot(incl >= (unsigned)need);
        int src = __ffsll((long long)ge) - 1; // first lane whose inclusive count reaches need
        unsigned excl_l = __shfl(incl - s, src, 64);
        unsigned c0 = __shfl(c.x, src, 64), c1 = __shfl(c.y, src, 64), c2 = __shfl(c.z, src, 64),
                 c3 = __shfl(c.w, src, 64);
        unsigned rem = (unsigned)need - excl_l; // 1-based rank inside this lane's 4 bins
        unsigned digit, cnt_b;
        if (rem <= c0) {
            digit = 0; cnt_b = c0;
        } else if (rem <= c0 + c1) {
            digit = 1; cnt_b = c1; rem -= c0;
        } else if (rem <= c0 + c1 + c2) {
            digit = 2; cnt_b = c2; rem -= c0 + c1;
        } else {
            digit = 3; cnt_b = c3; rem -= c0 + c1 + c2;
        }
        digit += 4u * (unsigned)src;
        prefix |= (u64)digit << shift;
        mask |= (u64)255u << shift;
        need = (int)rem;
        if ((unsigned)need == cnt_b) {
            // the k-th key is the largest key of the selected bucket: one max pass
            u64 best = 0;
            for (int i = lane; i < n; i += 64) {
                u64 key = keys[i];
                if ((key & mask) == prefix && key > best) best = key;
            }
            return wave_max_u64(best);
        }
    }
    return prefix;
}

// keep keys <= kth (in place, stable within chunks); returns the number kept
__device__ inline int wave_compact(u64* keys, int n, u64 kth) {
    const int lane = lane_id();
    int out = 0;
    for (int base = 0; base < n; base += 64) {
        int i = base + lane;
        u64 key = i < n ? keys[i] : ~0ull;
        bool keep = (i < n) && key <= kth;
        u64 m = __ballot(keep);
        wave_mem_sync(); // every load of this chunk has returned before any store below
        int pos = out + __popcll(m & ((1ull << lane) - 1ull));
        if (keep) keys[pos] = key;
        out += __popcll(m);
    }
    wave_mem_sync();
    return out;
}


} // namespace faiss_amd
