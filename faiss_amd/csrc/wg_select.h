// faiss_amd/csrc/wg_select.h -- workgroup-level exact k-selection over 64-bit keys held in LDS.
//
// These helpers are the gfx950 replacement for the reference's BlockSelect
// (faiss/gpu/utils/Select.cuh:139-335): instead of per-thread queues merged through bitonic
// networks, candidates that beat the running threshold are appended to an LDS reservoir and the
// reservoir is cut back to its k smallest keys by an MSB-first radix select whenever it fills.
// Keys are unique ((ordkey(distance) << 32) | position), so "k smallest keys" is a set, and the
// result does not depend on the order in which lanes append.
#pragma once
#include "common.h"

namespace faiss_amd {

typedef unsigned long long u64;

struct WgSelCtl {
    u64 kth;       // scratch for the bucket-maximum reduction
    unsigned cnt;  // live reservoir entries
    unsigned digit, rem, cnt_b;
};

__device__ __forceinline__ int wgs_lane() {
    return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// k-th smallest (1-based) of n unique keys in LDS, n > k >= 1.  Every thread of the workgroup
// calls it (uniform control flow, BLOCK threads); hist = 256 LDS words.
// Round 5 / 6 finding (DESIGN.md 6b "the wg_select_kth fault"): as a plain `__device__` template this function was NOT inlined --
// hipcc emitted one out-of-line copy per BLOCK and 112 `s_swappc_b64` call sites -- and an out-of-line function sees `keys`, `hist`
// and `ctl` as GENERIC pointers: every access to the LDS histogram became a FLAT instruction (flat_store / flat_atomic_add /
// flat_load through the aperture check) guarded before each s_barrier by `s_waitcnt lgkmcnt(0)` only.  On gfx950 that is not
// enough for a no-return flat_atomic_add that lands in LDS: under the right timing the wavefront that scans the 256 bins reads
// them before every increment has been performed, picks a later bucket, returns a key ABOVE the k-th smallest, and the
// compaction keeps more than k keys -- a partial result of n > k keys then overruns its slot of part_keys and the merge launch
// faults (tools/wgs_fault_repro.sh reproduces it: 100 % with the out-of-line build whose zeroing is written as a loop, 0 % with an
// extra `s_waitcnt vmcnt(0)` in front of the barriers, 0 % inlined).  Inlined, the pointers keep their LDS address space and the
// accesses are DS instructions, which lgkmcnt does cover.  tests/test_isa_lint_cpu.py asserts that no kernel file contains a
// call or a FLAT access at all where LDS is meant.
#ifdef FAISS_AMD_WGS_OUTOFLINE_REPRO
#define FA_WGS_INLINE __attribute__((noinline))
#else
#define FA_WGS_INLINE __forceinline__
#endif
#ifdef FAISS_AMD_WGS_VMCNT_REPRO
#define FA_WGS_SYNC()                                        \
    do {                                                     \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     \
        __syncthreads();                                     \
    } while (0)
#else
#define FA_WGS_SYNC() __syncthreads()
#endif
template <int BLOCK>
__device__ FA_WGS_INLINE u64 wg_select_kth(const u64* keys, int n, int k, unsigned* hist, WgSelCtl* ctl) {
    const int tid = threadIdx.x;
    u64 prefix = 0, mask = 0;
    int need = k;
    for (int shift = 56; shift >= 0; shift -= 8) {
        for (int i = tid; i < 256; i += BLOCK) hist[i] = 0; // (the form that faulted out of line in round 5: see above)
        FA_WGS_SYNC();
        for (int i = tid; i < n; i += BLOCK) {
            const u64 key = keys[i];
            if ((key & mask) == prefix) atomicAdd(&hist[(unsigned)(key >> shift) & 255u], 1u);
        }
        FA_WGS_SYNC();
        if (tid < 64) {
            // one wavefront scans the 256 bins: lane l owns bins 4l..4l+3
            const int lane = tid;
            const uint4 c = *(const uint4*)(hist + 4 * lane);
            const unsigned s = c.x + c.y + c.z + c.w;
            const unsigned incl = wave_incl_scan(s);
            const u64 ge = __ballot(incl >= (unsigned)need);
            const int src = __ffsll((long long)ge) - 1;
            if (lane == src) {
                unsigned rem = (unsigned)need - (incl - s);
                unsigned digit, cnt_b;
                if (rem <= c.x) {
                    digit = 0; cnt_b = c.x;
                } else if (rem <= c.x + c.y) {
                    digit = 1; cnt_b = c.y; rem -= c.x;
                } else if (rem <= c.x + c.y + c.z) {
                    digit = 2; cnt_b = c.z; rem -= c.x + c.y;
                } else {
                    digit = 3; cnt_b = c.w; rem -= c.x + c.y + c.z;
                }
                ctl->digit = digit + 4u * (unsigned)lane;
                ctl->rem = rem;
                ctl->cnt_b = cnt_b;
                ctl->kth = 0;
            }
        }
        FA_WGS_SYNC();
        prefix |= (u64)ctl->digit << shift;
        mask |= (u64)255u << shift;
        need = (int)ctl->rem;
        if ((unsigned)need == ctl->cnt_b) {
            // the k-th key is the largest key of the selected bucket
            u64 best = 0;
            for (int i = tid; i < n; i += BLOCK) {
                const u64 key = keys[i];
                if ((key & mask) == prefix && key > best) best = key;
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const u64 o = __shfl_xor(best, off, 64);
                best = o > best ? o : best;
            }
            if ((tid & 63) == 0 && best) atomicMax(&ctl->kth, best);
            FA_WGS_SYNC();
            const u64 r = ctl->kth;
            FA_WGS_SYNC(); // ctl is rewritten by the next call
            return r;
        }
        FA_WGS_SYNC(); // hist / ctl are rewritten by the next pass
    }
    return prefix;
}

// Keep the keys <= kth (in place).  Leaves ctl->cnt = number kept.  Works in rounds of BLOCK
// keys (one key per thread in registers): survivors of round r land in slots below the number
// of keys examined so far, i.e. only on positions that have already been read.
template <int BLOCK>
__device__ void wg_compact(u64* keys, int n, u64 kth, WgSelCtl* ctl) {
    const int tid = threadIdx.x;
    if (tid == 0) ctl->cnt = 0;
    for (int base = 0; base < n; base += BLOCK) {
        const int i = base + tid;
        const u64 held = i < n ? keys[i] : ~0ull;
        __syncthreads(); // every key of this round is in registers (and ctl->cnt is initialised)
        const bool keep = i < n && held <= kth;
        const u64 m = __ballot(keep);
        if (m) {
            const int lane = wgs_lane();
            const int first = __ffsll((long long)m) - 1;
            unsigned b = 0;
            if (lane == first) b = atomicAdd(&ctl->cnt, (unsigned)__popcll(m));
            b = __shfl(b, first, 64);
            if (keep) keys[b + __popcll(m & ((1ull << lane) - 1ull))] = held;
        }
    }
    __syncthreads();
}

// Wave-aggregated append of `key` (when pass) to the LDS reservoir; the caller guarantees room.
__device__ __forceinline__ void wg_append(u64* keys, WgSelCtl* ctl, bool pass, u64 key) {
    const u64 m = __ballot(pass);
    if (m) {
        const int lane = wgs_lane();
        const int first = __ffsll((long long)m) - 1;
        unsigned base = 0;
        if (lane == first) base = atomicAdd(&ctl->cnt, (unsigned)__popcll(m));
        base = __shfl(base, first, 64);
        if (pass) keys[base + __popcll(m & ((1ull << lane) - 1ull))] = key;
    }
}

// Bitonic sort of kp (power of two) (key32, id64) pairs in LDS, ascending by (key, id).
template <int BLOCK>
__device__ void wg_bitonic_sort(unsigned* w_key, int64_t* w_id, int kp) {
    const int tid = threadIdx.x;
    for (int size = 2; size <= kp; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < (kp >> 1); t += BLOCK) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool up = ((lo & size) == 0);
                const unsigned ka = w_key[lo], kb = w_key[hi];
                const int64_t ia = w_id[lo], ib = w_id[hi];
                const bool a_gt_b = (ka > kb) || (ka == kb && ia > ib);
                if (a_gt_b == up) {
                    w_key[lo] = kb; w_key[hi] = ka;
                    w_id[lo] = ib; w_id[hi] = ia;
                }
            }
            __syncthreads();
        }
    }
}

} // namespace faiss_amd
