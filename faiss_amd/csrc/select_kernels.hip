// faiss_amd/csrc/select_kernels.hip -- exact k-selection for gfx950, one workgroup per query.
//
// Replaces the reference's WarpSelect/BlockSelect family (faiss/gpu/utils/Select.cuh:139-335,
// BlockSelectKernel.cuh:15-132, MergeNetwork*.cuh) and the IVF two-pass selection
// (faiss/gpu/impl/IVFUtilsSelect1.cu:24-99, IVFUtilsSelect2.cu:49-155) for the merge stage.
// Unlike the reference (which leaves ties unordered, MergeNetworkWarp.cuh:41-78) the result is
// the k smallest 64-bit keys exactly, then ordered by (distance, label): the total order the
// CPU reference produces for IndexFlat (faiss/impl/ResultHandler.h:276-281, 439-453).
//
// Algorithm: MSB-first 8-bit radix select over the candidate keys (histogram in LDS, early
// exit when the k-th key is the maximum of its bucket), gather of the <= k winners into LDS,
// payload -> label translation, bitonic sort of the winners, coalesced write-out.
#include "kernels.h"
#include <type_traits>
#include "wave_select.h"
#include "wg_select.h"

namespace faiss_amd {

constexpr int SEL_THREADS = 256;

struct SelShared {
    unsigned hist[256];
    unsigned scan[256];
    unsigned wsum[4];
    u64 prefix;
    u64 mask;
    u64 kth;
    int need;
    int done;
    unsigned total;
    unsigned nwin;
};

template <typename F>
__device__ __forceinline__ void for_each_key(const SelectParams& p, int q, F f) {
    const u64* base = p.keys + (p.q_off ? p.q_off[q] : (int64_t)q * p.q_stride);
    for (int s = 0; s < p.nseg; ++s) {
        const unsigned cnt = p.seg_cnt[(int64_t)q * p.nseg + s];
        const u64* seg = base + (int64_t)s * p.seg_stride;
        for (unsigned i = threadIdx.x; i < cnt; i += SEL_THREADS) f(seg[i]);
    }
}

__global__ void __launch_bounds__(SEL_THREADS) select_k_kernel(SelectParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    SelShared* sh = (SelShared*)smem;
    // winners: ordkey (u32) + label (i64); kp = pow2 >= k entries
    int kp = 1;
    while (kp < p.k) kp <<= 1;
    int64_t* w_id = (int64_t*)(smem + ((sizeof(SelShared) + 15) & ~15));
    unsigned* w_key = (unsigned*)(w_id + kp);

    const int q = blockIdx.x;
    const int tid = threadIdx.x;

    // ---- total candidate count
    if (tid == 0) {
        sh->total = 0;
        sh->nwin = 0;
        sh->prefix = 0;
        sh->mask = 0;
        sh->need = p.k;
        sh->done = 0;
        sh->kth = ~0ull;
    }
    __syncthreads();
    {
        unsigned loc = 0;
        for (int s = tid; s < p.nseg; s += SEL_THREADS) loc += p.seg_cnt[(int64_t)q * p.nseg + s];
        if (loc) atomicAdd(&sh->total, loc);
    }
    __syncthreads();
    const unsigned total = sh->total;

    // ---- radix select of the k-th smallest key (only when there are more than k)
    if (total > (unsigned)p.k) {
        for (int shift = 56; shift >= 0; shift -= 8) {
            sh->hist[tid] = 0;
            __syncthreads();
            const u64 prefix = sh->prefix, mask = sh->mask;
            for_each_key(p, q, [&](u64 key) {
                if ((key & mask) == prefix) atomicAdd(&sh->hist[(unsigned)(key >> shift) & 255u], 1u);
            });
            __syncthreads();
            // inclusive scan of the 256 bins: inside each wavefront by shuffles, then the three wavefront totals before
            // it (one barrier; the Hillis-Steele ladder in LDS it replaces took 16 per radix pass -- most of the kernel's
            // time on the ~2000-key segments of the list-major IVF scan)
            const unsigned v = sh->hist[tid];
            {
                unsigned inc = wave_incl_scan(v);
                const int ln = tid & 63;
                if (ln == 63) sh->wsum[tid >> 6] = inc;
                __syncthreads();
                for (int w = 0; w < (tid >> 6); ++w) inc += sh->wsum[w];
                sh->scan[tid] = inc;
            }
            const unsigned incl = sh->scan[tid];
            const unsigned excl = incl - v;
            const unsigned need = (unsigned)sh->need;
            __syncthreads();
            if (excl < need && need <= incl) {
                // exactly one bin satisfies this
                sh->prefix = prefix | ((u64)tid << shift);
                sh->mask = mask | ((u64)255u << shift);
                sh->need = (int)(need - excl);
                sh->done = ((need - excl) == v) ? 1 : 0;
            }
            __syncthreads();
            if (sh->done || shift == 0) break;
        }
        if (sh->done) {
            // k-th key = max key of the selected bucket
            if (tid == 0) sh->kth = 0;
            __syncthreads();
            const u64 prefix = sh->prefix, mask = sh->mask;
            u64 best = 0;
            for_each_key(p, q, [&](u64 key) {
                if ((key & mask) == prefix && key > best) best = key;
            });
            if (best) atomicMax(&sh->kth, best);
            __syncthreads();
        } else {
            if (tid == 0) sh->kth = sh->prefix;
            __syncthreads();
        }
    }
    const u64 kth = sh->kth;
    if (p.kth_out) {
        // threshold-only use (list-major IVF scan): kth_out = the bound (with no more than k keys every distance
        // qualifies: kth = ~0), and -- cnt_out given -- the segment (one per query) is cut back to its keys <= kth, which
        // is all of it that can still be part of the answer
        if (tid == 0) p.kth_out[q] = (uint32_t)(kth >> 32);
        if (p.cnt_out && total > (unsigned)p.k) {
            u64* keep = (u64*)w_id; // [kp]
            u64* seg = const_cast<u64*>(p.keys) + (p.q_off ? p.q_off[q] : (int64_t)q * p.q_stride);
            for (unsigned i = tid; i < total; i += SEL_THREADS) {
                const u64 key = seg[i];
                if (key <= kth) {
                    const unsigned slot = atomicAdd(&sh->nwin, 1u);
                    if (slot < (unsigned)kp) keep[slot] = key;
                }
            }
            __syncthreads(); // every read of the segment precedes the writes below
            const unsigned nw = min(sh->nwin, (unsigned)p.k);
            for (unsigned i = tid; i < nw; i += SEL_THREADS) seg[i] = keep[i];
            if (tid == 0) p.cnt_out[q] = nw;
        }
        return;
    }

    // ---- gather winners (keys <= kth): exactly min(total, k) because keys are unique
    for (int i = tid; i < kp; i += SEL_THREADS) {
        w_key[i] = 0xffffffffu;
        w_id[i] = INT64_MAX;
    }
    __syncthreads();
    {
        const uint32_t* pre = p.mode == 1 ? p.ivf_prefix + (int64_t)q * (p.nprobe + 1) : nullptr;
        for_each_key(p, q, [&](u64 key) {
            if (key <= kth) {
                unsigned slot = atomicAdd(&sh->nwin, 1u);
                if (slot < (unsigned)kp) {
                    uint32_t payload = (uint32_t)key;
                    int64_t label;
                    if (p.mode == 0) {
                        label = (int64_t)payload + p.id_base;
                    } else if (p.mode == 2) {
                        const int s = (int)(payload / (uint32_t)p.k), r = (int)(payload % (uint32_t)p.k);
                        label = p.merge_ids[((int64_t)s * p.nq + q) * p.k + r] +
                                (p.merge_base ? p.merge_base[s] : 0);
                    } else {
                        // binary search: largest pr with pre[pr] <= payload
                        int lo = 0, hi = p.nprobe; // invariant pre[lo] <= payload < pre[hi]
                        while (hi - lo > 1) {
                            int mid = (lo + hi) >> 1;
                            if (pre[mid] <= payload) lo = mid;
                            else hi = mid;
                        }
                        const int64_t list = p.coarse_ids[(int64_t)q * p.nprobe + lo];
                        label = p.arena_ids[p.list_start[list] + (payload - pre[lo])];
                    }
                    w_key[slot] = (uint32_t)(key >> 32);
                    w_id[slot] = label;
                }
            }
        });
    }
    __syncthreads();
    const int nwin = min((int)sh->nwin, p.k);

    // ---- bitonic sort of kp entries by (ordkey, label) ascending
    for (int size = 2; size <= kp; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < (kp >> 1); t += SEL_THREADS) {
                int lo = 2 * t - (t & (stride - 1));
                int hi = lo + stride;
                bool up = ((lo & size) == 0);
                unsigned ka = w_key[lo], kb = w_key[hi];
                int64_t ia = w_id[lo], ib = w_id[hi];
                bool a_gt_b = (ka > kb) || (ka == kb && ia > ib);
                if (a_gt_b == up) {
                    w_key[lo] = kb; w_key[hi] = ka;
                    w_id[lo] = ib; w_id[hi] = ia;
                }
            }
            __syncthreads();
        }
    }

    // ---- write-out
    const float pad = neutral_distance(p.metric);
    for (int i = tid; i < p.k; i += SEL_THREADS) {
        float dis;
        int64_t id;
        if (i < nwin && w_key[i] < kInvalidOrdKey) {
            dis = unordkey_rt(p.metric, w_key[i]);
            id = w_id[i];
        } else {
            dis = pad;
            id = -1;
        }
        p.out_dis[(int64_t)q * p.k + i] = dis;
        p.out_ids[(int64_t)q * p.k + i] = id;
    }
}

// ---------------------------------------------------------------------------------
// One WAVEFRONT per query (round 3): the bound and the final selection of the list-major IVF scan choose k of a few
// thousand keys per query, 10 000 times per search; the workgroup-per-query radix select above spends its time in
// barriers and in LDS atomics that pile onto one histogram bin (the keys of a query share their leading bytes).  Here
// a query's keys sit in the registers of one wavefront (64 per lane); the k-th smallest distance is found by bisection
// on the 32-bit ordkey between the segment's minimum and maximum -- a compare-and-count sweep over the registers and a
// wave reduction per step, no barrier, no atomic -- ties at that distance are resolved by a second bisection on the
// scan position (the key's lower half; positions are unique), which yields EXACTLY the k smallest 64-bit keys, the set
// select_k_kernel selects.  Bound mode (kth_out / cnt_out): the segment is cut back to those k keys.  Selection mode
// (mode 1): winners -> labels -> bitonic sort by (distance, label) in a per-wave LDS slice -> best first.
// A query with more keys than the registers hold streams them from memory in every step (rare by construction).
// ---------------------------------------------------------------------------------
constexpr int WS_KPL = 64; // keys per lane held in registers
__device__ __forceinline__ unsigned ws_sum(unsigned v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ unsigned ws_min(unsigned v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = min(v, (unsigned)__shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ unsigned ws_max(unsigned v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = max(v, (unsigned)__shfl_xor(v, off, 64));
    return v;
}

template <bool BOUND>
__global__ void __launch_bounds__(256) wave_select_kernel(SelectParams p, int kp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + wave;
    if (q >= p.nq) return; // (no barrier below: wavefronts are independent)
    const unsigned n = p.seg_cnt[q];
    if (p.small_done && n <= 256u) return; // (served by small_select_kernel)
    u64* seg = const_cast<u64*>(p.keys) + (p.q_off ? p.q_off[q] : (int64_t)q * p.q_stride);
    const unsigned k = (unsigned)p.k;
    const bool resident = n <= 64u * WS_KPL;
    const int nrow8 = (int)((n + 511u) >> 9); // groups of 8 register slots (512 keys) that hold keys
    // slot i of lane l = key l + 64 i; slots past the end hold ~0 (hi = lo = 0xffffffff: no real key, positions stay below)
    unsigned hi[WS_KPL], lo[WS_KPL];
    if (resident) {
#pragma unroll
        for (int i = 0; i < WS_KPL; ++i) {
            const unsigned idx = (unsigned)lane + 64u * i;
            const u64 key = idx < n ? seg[idx] : ~0ull;
            hi[i] = (unsigned)(key >> 32);
            lo[i] = (unsigned)key;
        }
    }
    // f(hi, lo) over this lane's keys (exact: slots past the end are skipped)
    auto for_keys = [&](auto f) __attribute__((always_inline)) {
        if (resident) {
#pragma unroll
            for (int i = 0; i < WS_KPL; ++i)
                if ((unsigned)lane + 64u * i < n) f(hi[i], lo[i]);
        } else {
            for (unsigned idx = lane; idx < n; idx += 64) {
                const u64 key = seg[idx];
                f((unsigned)(key >> 32), (unsigned)key);
            }
        }
    };
    // wave-wide count of the keys with pred(hi, lo), for predicates that are false on the ~0 filler: one compare per
    // slot into a lane mask, counted on the scalar unit -- no cross-lane traffic
    auto count = [&](auto pred) __attribute__((always_inline)) -> unsigned {
        unsigned c = 0;
        if (resident) {
#pragma unroll
            for (int g = 0; g < WS_KPL / 8; ++g) {
                if (g < nrow8) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) c += (unsigned)__builtin_popcountll(__ballot(pred(hi[8 * g + e], lo[8 * g + e])));
                }
            }
        } else {
            for (unsigned base = 0; base < n; base += 64) {
                const unsigned idx = base + lane;
                const u64 key = idx < n ? seg[idx] : ~0ull;
                c += (unsigned)__builtin_popcountll(__ballot(pred((unsigned)(key >> 32), (unsigned)key)));
            }
        }
        return c;
    };
    unsigned T = 0xffffffffu, P = 0xffffffffu; // winners: hi < T, or hi == T and lo <= P
    bool ranked = false;
    if (n > k && n <= 256u) {
        // small segments (round 5: the filter path hands ~ k + a few dozen re-derived candidates per query): the k-th smallest
        // key by COUNTING -- the keys go through this wave's LDS slice, every lane ranks its (at most four) keys against all n
        // with broadcast reads; the key of rank k - 1 is the bound.  The bisection below costs up to 64 rounds of ballots over the
        // key space whatever n is: 20 us per query for 140 keys.  (Keys are distinct: the payload is a scan position / an id;
        // should two ever coincide no key has rank k - 1 and the bisection decides.)
        u64* sk = (u64*)(smem + (size_t)4 * kp * (BOUND ? 8 : 12)) + (size_t)wave * 256;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned idx = (unsigned)lane + 64u * i;
            if (idx < n) sk[idx] = ((u64)hi[i] << 32) | lo[i];
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        // (two keys of the lane per walk over the slice: the 128 key registers leave room for little else at three waves per SIMD)
        bool found = false;
        unsigned fh = 0u, fl = 0u;
#pragma unroll
        for (int i0 = 0; i0 < 4; i0 += 2) {
            if (64u * i0 >= n) break; // (wave-uniform)
            unsigned r0 = 0u, r1 = 0u;
            for (unsigned j = 0; j < n; ++j) {
                const u64 kj = sk[j];
                const unsigned jh = (unsigned)(kj >> 32), jl = (unsigned)kj;
                r0 += (jh < hi[i0] || (jh == hi[i0] && jl < lo[i0])) ? 1u : 0u;
                r1 += (jh < hi[i0 + 1] || (jh == hi[i0 + 1] && jl < lo[i0 + 1])) ? 1u : 0u;
            }
            if ((unsigned)lane + 64u * i0 < n && r0 == k - 1u) found = true, fh = hi[i0], fl = lo[i0];
            if ((unsigned)lane + 64u * (i0 + 1) < n && r1 == k - 1u) found = true, fh = hi[i0 + 1], fl = lo[i0 + 1];
        }
        const unsigned long long fb = __ballot(found);
        if (fb) {
            const int src = __builtin_ctzll(fb);
            T = (unsigned)__shfl((int)fh, src, 64);
            P = (unsigned)__shfl((int)fl, src, 64);
            ranked = true;
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier(); // (the slice is not reused before every lane has read it)
    }
    if (n > k && !ranked) {
        unsigned mn = 0xffffffffu, mx = 0u;
        for_keys([&](unsigned h, unsigned) {
            mn = min(mn, h);
            mx = max(mx, h);
        });
        unsigned lo_b = ws_min(mn), hi_b = ws_max(mx);
        while (lo_b < hi_b) { // smallest T with #(hi <= T) >= k; mid < hi_b <= 0xffffffff: the filler never counts
            const unsigned mid = lo_b + ((hi_b - lo_b) >> 1);
            if (count([&](unsigned h, unsigned) { return h <= mid; }) >= k) hi_b = mid;
            else lo_b = mid + 1;
        }
        T = lo_b;
        unsigned cl = 0, ct = 0;
        for_keys([&](unsigned h, unsigned) {
            cl += h < T ? 1u : 0u;
            ct += h == T ? 1u : 0u;
        });
        cl = ws_sum(cl);
        ct = ws_sum(ct);
        const unsigned need = k - cl; // 1 <= need <= ct
        if (need < ct) {
            unsigned pl = 0u, ph = 0xffffffffu; // smallest P with #(hi == T, lo <= P) >= need (mid < 0xffffffff)
            while (pl < ph) {
                const unsigned mid = pl + ((ph - pl) >> 1);
                if (count([&](unsigned h, unsigned l) { return h == T && l <= mid; }) >= need) ph = mid;
                else pl = mid + 1;
            }
            P = pl;
        }
    }
    auto wins = [&](unsigned h, unsigned l) { return h < T || (h == T && l <= P); };
    // slots of this lane's winners: exclusive scan of the per-lane counts
    unsigned mine = 0;
    for_keys([&](unsigned h, unsigned l) { mine += wins(h, l) ? 1u : 0u; });
    const unsigned inc = wave_incl_scan(mine);
    unsigned at = inc - mine;
    const unsigned nwin = (unsigned)__shfl(inc, 63, 64); // = min(n, k)

    if (BOUND) {
        // kth_out = the bound (with no more than k keys every distance qualifies: 0xffffffff); the segment keeps the k keys.
        // They pass through this wave's LDS slice: every key has been read before the first one is overwritten, also
        // when the keys are streamed from memory.
        if (lane == 0) p.kth_out[q] = n > k ? T : 0xffffffffu;
        if (p.cnt_out && n > k) {
            u64* keep = (u64*)smem + (size_t)wave * kp;
            for_keys([&](unsigned h, unsigned l) {
                if (wins(h, l)) keep[at++] = ((u64)h << 32) | l;
            });
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            for (unsigned t = lane; t < nwin; t += 64) seg[t] = keep[t];
            if (lane == 0) p.cnt_out[q] = nwin;
        }
        return;
    }

    // ---- winners -> this wave's LDS slice (ordkey, payload), then payload -> label one winner per lane
    int64_t* w_id = (int64_t*)smem + (size_t)wave * kp;
    unsigned* w_key = (unsigned*)(smem + (size_t)4 * kp * 8) + (size_t)wave * kp;
    for (int i = lane; i < kp; i += 64) {
        w_key[i] = 0xffffffffu;
        w_id[i] = INT64_MAX;
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    for_keys([&](unsigned h, unsigned l) {
        if (wins(h, l)) {
            w_key[at] = h;
            w_id[at] = (int64_t)l;
            ++at;
        }
    });
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    {
        const uint32_t* pre = p.mode == 1 ? p.ivf_prefix + (int64_t)q * (p.nprobe + 1) : nullptr;
        // up to 63 probes: lane pr keeps pre[pr] and the arena row of the first entry of probe pr's list; the binary search
        // of a winner then reads other lanes' registers instead of walking memory (five dependent loads per winner)
        const bool in_regs = p.mode == 1 && p.nprobe < 64;
        uint32_t pre_l = 0xffffffffu;
        int64_t base_l = 0;
        if (in_regs) {
            if (lane <= p.nprobe) pre_l = pre[lane];
            if (lane < p.nprobe) {
                const int64_t list = p.coarse_ids[(int64_t)q * p.nprobe + lane];
                base_l = list >= 0 ? p.list_start[list] : 0;
            }
        }
        for (unsigned t0 = 0; t0 < nwin; t0 += 64) { // (uniform trip count: the shuffles below need every lane)
            const unsigned t = t0 + lane;
            const bool live = t < nwin;
            const uint32_t l = live ? (uint32_t)w_id[t] : 0u;
            int64_t label = -1;
            if (p.mode == 0) {
                label = (int64_t)l + p.id_base;
            } else if (in_regs) {
                int a = 0, b = p.nprobe; // invariant pre[a] <= payload < pre[b]
                for (int step = 0; step < 6; ++step) { // 2^6 >= 64 probes
                    const int mid = (a + b) >> 1;
                    const uint32_t pm = (uint32_t)__shfl((int)pre_l, mid, 64);
                    if (b - a > 1) {
                        if (pm <= l) a = mid;
                        else b = mid;
                    }
                }
                const uint32_t pa = (uint32_t)__shfl((int)pre_l, a, 64);
                const int64_t base = ((int64_t)__shfl((int)(base_l >> 32), a, 64) << 32) | (uint32_t)__shfl((int)base_l, a, 64);
                if (live) label = p.arena_ids[base + (l - pa)];
            } else if (live) {
                // binary search: largest pr with pre[pr] <= payload
                int a = 0, b = p.nprobe; // invariant pre[a] <= payload < pre[b]
                while (b - a > 1) {
                    const int mid = (a + b) >> 1;
                    if (pre[mid] <= l) a = mid;
                    else b = mid;
                }
                const int64_t list = p.coarse_ids[(int64_t)q * p.nprobe + a];
                label = p.arena_ids[p.list_start[list] + (l - pre[a])];
            }
            if (live) w_id[t] = label;
        }
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    // ---- bitonic sort of kp entries by (ordkey, label) ascending, inside the wavefront
    for (int size = 2; size <= kp; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = lane; t < (kp >> 1); t += 64) {
                const int a = 2 * t - (t & (stride - 1));
                const int b = a + stride;
                const bool up = ((a & size) == 0);
                const unsigned ka = w_key[a], kb = w_key[b];
                const int64_t ia = w_id[a], ib = w_id[b];
                const bool a_gt_b = (ka > kb) || (ka == kb && ia > ib);
                if (a_gt_b == up) {
                    w_key[a] = kb;
                    w_key[b] = ka;
                    w_id[a] = ib;
                    w_id[b] = ia;
                }
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
    }
    const float pad = neutral_distance(p.metric);
    for (int i = lane; i < p.k; i += 64) {
        float dis;
        int64_t id;
        if (i < (int)nwin && w_key[i] < kInvalidOrdKey) {
            dis = unordkey_rt(p.metric, w_key[i]);
            id = w_id[i];
        } else {
            dis = pad;
            id = -1;
        }
        p.out_dis[(int64_t)q * p.k + i] = dis;
        p.out_ids[(int64_t)q * p.k + i] = id;
    }
}
// ---------------------------------------------------------------------------------
// Small segments (round 5): one wavefront per query for n <= 256 keys, k <= 256, label modes 0 / 1 with fewer than 64 probes --
// what the filter path of IVFPQ hands over: ~ k + a few dozen re-derived candidates per query.  At most four keys per lane,
// both rankings by COUNTING through the wave's LDS slice (broadcast reads): (distance, position) picks the k winners -- rank =
// winner slot, no scan --, (distance, label) orders them, ties of both by the winner slot: the result of wave_select_kernel /
// select_k_kernel.  Few registers: 8 wavefronts per SIMD, the whole batch of 10 000 queries resident at once (the general
// kernel keeps 64 keys per lane in registers: 2 wavefronts per SIMD, 20 us per query whatever n is -- 0.10 ms per search).
// Queries with more keys are left to the general kernel (SelectParams::small_done tells it which were served).
// ---------------------------------------------------------------------------------
constexpr int SS_N = 256;
__global__ void __launch_bounds__(256) small_select_kernel(SelectParams p) {
    __shared__ u64 s_keys[4][SS_N];
    __shared__ unsigned s_wk[4][SS_N];
    __shared__ int64_t s_wl[4][SS_N];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + wave;
    if (q >= p.nq) return; // (wavefronts are independent: no workgroup barrier below)
    const unsigned n = p.seg_cnt[q];
    if (n > (unsigned)SS_N) return;
    const u64* seg = p.keys + (p.q_off ? p.q_off[q] : (int64_t)q * p.q_stride);
    const unsigned k = (unsigned)p.k;
    u64* sk = s_keys[wave];
    unsigned* wk = s_wk[wave];
    int64_t* wl = s_wl[wave];
    u64 key[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned idx = (unsigned)lane + 64u * i;
        key[i] = idx < n ? seg[idx] : ~0ull;
        if (idx < n) sk[idx] = key[i];
    }
    // probes of the query in the lanes' registers (mode 1): a winner's position -> probe by bisection over other lanes' values
    uint32_t pre_l = 0xffffffffu;
    int64_t base_l = 0;
    if (p.mode == 1) {
        const uint32_t* pre = p.ivf_prefix + (int64_t)q * (p.nprobe + 1);
        if (lane <= p.nprobe) pre_l = pre[lane];
        if (lane < p.nprobe) {
            const int64_t list = p.coarse_ids[(int64_t)q * p.nprobe + lane];
            base_l = list >= 0 ? p.list_start[list] : 0;
        }
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    unsigned r[4] = {0u, 0u, 0u, 0u};
    // (the kernel is VALU-bound -- 10 000 waves x a few thousand instructions --, so only the register slots that hold keys are
    // ranked: n = 140 fills three of the four)
    auto rank_slots = [&](auto ns_c) __attribute__((always_inline)) {
        constexpr int NS = decltype(ns_c)::value;
        for (unsigned j = 0; j < n; ++j) {
            const u64 kj = sk[j];
#pragma unroll
            for (int i = 0; i < NS; ++i) r[i] += kj < key[i] ? 1u : 0u;
        }
    };
    if (n <= 64u) rank_slots(std::integral_constant<int, 1>{});
    else if (n <= 128u) rank_slots(std::integral_constant<int, 2>{});
    else if (n <= 192u) rank_slots(std::integral_constant<int, 3>{});
    else rank_slots(std::integral_constant<int, 4>{});
    const unsigned nwin = min(n, k);
    // winners -> slot = rank, with their labels
#pragma unroll
    for (int i = 0; i < 4; ++i) { // (uniform trip count: the shuffles need every lane)
        const bool win = (unsigned)lane + 64u * i < n && r[i] < k;
        const uint32_t l = (uint32_t)key[i];
        int64_t label;
        if (p.mode == 0) {
            label = (int64_t)l + p.id_base;
        } else {
            int a = 0, b = p.nprobe; // invariant pre[a] <= payload < pre[b]
            for (int step = 0; step < 6; ++step) {
                const int mid = (a + b) >> 1;
                const uint32_t pm = (uint32_t)__shfl((int)pre_l, mid, 64);
                if (b - a > 1) {
                    if (pm <= l) a = mid;
                    else b = mid;
                }
            }
            const uint32_t pa = (uint32_t)__shfl((int)pre_l, a, 64);
            const int64_t base = ((int64_t)__shfl((int)(base_l >> 32), a, 64) << 32) | (uint32_t)__shfl((int)base_l, a, 64);
            label = win ? p.arena_ids[base + (l - pa)] : -1;
        }
        if (win) {
            wk[r[i]] = (unsigned)(key[i] >> 32);
            wl[r[i]] = label;
        }
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const float pad = neutral_distance(p.metric);
    float* od = p.out_dis + (int64_t)q * k;
    int64_t* oi = p.out_ids + (int64_t)q * k;
    for (unsigned t = lane; t < k; t += 64) {
        if (t >= nwin) { // fewer keys than k: the tail is padding
            od[t] = pad;
            oi[t] = -1;
            continue;
        }
        // The slots are in (distance, position) order, so the winners with a smaller distance are exactly the slots before the
        // run of equal distances around t: only that run -- one slot unless distances tie -- is ranked by (label, slot).
        const unsigned a = wk[t];
        const int64_t ia = wl[t];
        unsigned lo = t, hi = t + 1u;
        while (lo > 0u && wk[lo - 1u] == a) --lo;
        while (hi < nwin && wk[hi] == a) ++hi;
        unsigned r2 = lo;
        for (unsigned j = lo; j < hi; ++j) {
            const int64_t ib = wl[j];
            r2 += (ib < ia || (ib == ia && j < t)) ? 1u : 0u;
        }
        const bool real = a < kInvalidOrdKey;
        od[r2] = real ? unordkey_rt(p.metric, a) : pad;
        oi[r2] = real ? ia : -1;
    }
}
static bool small_select_serves(const SelectParams& p) {
    return !p.kth_out && p.nseg == 1 && p.k <= SS_N && (p.mode == 0 || (p.mode == 1 && p.nprobe < 64)) && p.max_cnt > 0;
}

static bool wave_select_serves(const SelectParams& p) {
    static const char* e = experiment_env("FAISS_AMD_WAVE_SELECT"); // timing experiments: 0 = the radix kernel everywhere
    if (e && atoi(e) == 0) return false;
    if (p.nseg != 1 || p.k > 256 || p.max_cnt <= 0) return false;
    // (segments longer than the 4096 keys the registers hold are streamed from memory in every bisection step: fine for
    // the odd query, not for a launch made of them)
    return (p.kth_out || p.mode == 0 || p.mode == 1) && p.max_cnt <= 2 * 64 * WS_KPL;
}

__global__ void pack_merge_keys_kernel(int metric, const float* __restrict__ all_d,
                                       const int64_t* __restrict__ all_i, int nshard, int nq, int k,
                                       u64* __restrict__ keys, uint32_t* __restrict__ cnt) {
    const int64_t total = (int64_t)nq * nshard * k;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(t / ((int64_t)nshard * k));
        const int rem = (int)(t - (int64_t)q * nshard * k);
        const int s = rem / k, r = rem - s * k;
        const int64_t src = ((int64_t)s * nq + q) * k + r;
        const uint32_t ok = all_i[src] >= 0 ? ordkey_rt(metric, all_d[src]) : 0xffffffffu;
        keys[t] = ((u64)ok << 32) | (uint32_t)rem;
        if (rem == 0) cnt[q] = (uint32_t)(nshard * k);
    }
}

void launch_pack_merge_keys(int metric, const float* all_d, const int64_t* all_i, int nshard, int nq, int k,
                            u64* keys, uint32_t* cnt, hipStream_t stream) {
    if (nq == 0) return;
    const int64_t total = (int64_t)nq * nshard * k;
    unsigned grid = (unsigned)std::min<int64_t>(div_up(total, 256), 65535 * 8);
    hipLaunchKernelGGL(pack_merge_keys_kernel, dim3(grid), dim3(256), 0, stream, metric, all_d, all_i, nshard,
                       nq, k, keys, cnt);
    HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------
// stand-alone exercise of the three selection primitives (test hook, faiss/gpu/test/TestGpuSelect.cu:23-198 style):
// k best of every row of a [rows][cols] matrix
// ---------------------------------------------------------------------------------
__global__ void pack_row_keys_kernel(int metric, const float* __restrict__ vals, int rows, int cols, u64* __restrict__ keys,
                                     uint32_t* __restrict__ cnt) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)rows * cols) return;
    const int c = (int)(t % cols);
    keys[t] = ((u64)ordkey_rt(metric, vals[t]) << 32) | (uint32_t)c;
    if (c == 0) cnt[t / cols] = (uint32_t)cols;
}

// wg_select.h: the row streamed through an LDS reservoir exactly as the fused IVF scans do (threshold, append, cut back
// to k when full, final cut + sort)
template <int FB>
__global__ void __launch_bounds__(FB) wg_reservoir_test_kernel(int metric, const u64* __restrict__ keys, int cols, int k,
                                                                int kp, int cap, float* __restrict__ out_dis,
                                                                int64_t* __restrict__ out_ids) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64* res = (u64*)smem;                          // [cap]
    int64_t* w_id = (int64_t*)(res + cap);          // [kp]
    unsigned* w_key = (unsigned*)(w_id + kp);       // [kp]
    unsigned* hist = w_key + ((kp + 1) & ~1);       // [256] (8-byte aligned: WgSelCtl holds a 64-bit atomic)
    WgSelCtl* ctl = (WgSelCtl*)(hist + 256);
    const int tid = threadIdx.x, r = blockIdx.x;
    const u64* row = keys + (int64_t)r * cols;
    if (tid == 0) ctl->cnt = 0;
    __syncthreads();
    u64 tau = ~0ull;
    for (int base = 0; base < cols; base += FB) {
        const int n_ = (int)ctl->cnt;
        __syncthreads();
        if (n_ + FB > cap) {
            const u64 kth = wg_select_kth<FB>(res, n_, k, hist, ctl);
            wg_compact<FB>(res, n_, kth, ctl);
            tau = kth;
        }
        const int i = base + tid;
        const u64 key = i < cols ? row[i] : ~0ull;
        wg_append(res, ctl, i < cols && key < tau, key);
        __syncthreads();
    }
    int n = (int)ctl->cnt;
    if (n > k) {
        const u64 kth = wg_select_kth<FB>(res, n, k, hist, ctl);
        wg_compact<FB>(res, n, kth, ctl);
        n = (int)ctl->cnt;
    }
    for (int i = tid; i < kp; i += FB) {
        w_key[i] = i < n ? (uint32_t)(res[i] >> 32) : 0xffffffffu;
        w_id[i] = i < n ? (int64_t)(uint32_t)res[i] : INT64_MAX;
    }
    __syncthreads();
    wg_bitonic_sort<FB>(w_key, w_id, kp);
    for (int i = tid; i < k; i += FB) {
        const bool ok = i < n && w_key[i] < kInvalidOrdKey;
        out_dis[(int64_t)r * k + i] = ok ? unordkey_rt(metric, w_key[i]) : neutral_distance(metric);
        out_ids[(int64_t)r * k + i] = ok ? w_id[i] : -1;
    }
}

// wave_select.h: one wavefront per row, keys in global memory (the per-(query, split) reservoirs of the flat scan);
// the winners are written in whatever order the compaction leaves them
__global__ void __launch_bounds__(64) wave_select_test_kernel(int metric, u64* __restrict__ keys, int cols, int k,
                                                              float* __restrict__ out_dis, int64_t* __restrict__ out_ids) {
    __shared__ unsigned hist[256];
    const int r = blockIdx.x, lane = threadIdx.x;
    u64* row = keys + (int64_t)r * cols;
    int n = cols;
    if (n > k) {
        const u64 kth = wave_select_kth(row, n, k, hist);
        n = wave_compact(row, n, kth);
    }
    for (int i = lane; i < k; i += 64) {
        const bool ok = i < n && (uint32_t)(row[i] >> 32) < kInvalidOrdKey;
        out_dis[(int64_t)r * k + i] = ok ? unordkey_rt(metric, (uint32_t)(row[i] >> 32)) : neutral_distance(metric);
        out_ids[(int64_t)r * k + i] = ok ? (int64_t)(uint32_t)row[i] : -1;
    }
}

void launch_select_test(int which, int metric, const float* vals, int rows, int cols, int k, float* out_dis,
                        int64_t* out_ids, unsigned long long* keys, uint32_t* cnt, hipStream_t stream) {
    if (rows == 0) return;
    FA_THROW_IF_NOT(k >= 1 && k <= kMaxSelectionK && cols >= 1);
    hipLaunchKernelGGL(pack_row_keys_kernel, dim3((unsigned)div_up((size_t)rows * cols, 256)), dim3(256), 0, stream, metric,
                       vals, rows, cols, keys, cnt);
    if (which == 0) {
        SelectParams sp{};
        sp.metric = metric;
        sp.nq = rows;
        sp.k = k;
        sp.keys = keys;
        sp.q_stride = cols;
        sp.nseg = 1;
        sp.seg_cnt = cnt;
        sp.mode = 0;
        sp.out_dis = out_dis;
        sp.out_ids = out_ids;
        launch_select_k(sp, stream);
    } else if (which == 1) {
        constexpr int FB = 512;
        int kp = 1;
        while (kp < k) kp <<= 1;
        int cap = 1024;
        while (cap < k + FB) cap <<= 1;
        const size_t lds = (size_t)cap * 8 + (size_t)kp * 8 + (size_t)((kp + 1) & ~1) * 4 + 1024 + 64;
        HIP_CHECK(hipFuncSetAttribute((const void*)wg_reservoir_test_kernel<FB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds));
        hipLaunchKernelGGL((wg_reservoir_test_kernel<FB>), dim3((unsigned)rows), dim3(FB), lds, stream, metric, keys, cols, k,
                           kp, cap, out_dis, out_ids);
    } else {
        hipLaunchKernelGGL(wave_select_test_kernel, dim3((unsigned)rows), dim3(64), 0, stream, metric, keys, cols, k, out_dis,
                           out_ids);
    }
    HIP_CHECK(hipGetLastError());
}

void launch_select_k(const SelectParams& p, hipStream_t stream) {
    if (p.nq == 0) return;
    FA_THROW_IF_NOT(p.k >= 1 && p.k <= kMaxSelectionK);
    int kp = 1;
    while (kp < p.k) kp <<= 1;
    if (wave_select_serves(p)) {
        const dim3 grid((unsigned)div_up(p.nq, 4));
        SelectParams pw = p;
        pw.small_done = 0; // (set here only: a caller's uninitialised field must not make the general kernel skip queries)
        if (small_select_serves(p)) {
            hipLaunchKernelGGL(small_select_kernel, grid, dim3(256), 0, stream, p);
            pw.small_done = 1; // the general kernel below takes the queries with more than 256 keys
        }
        const SelectParams& p = pw;
        // (+ 4 x 256 keys: the counting rank of small segments)
        if (p.kth_out) hipLaunchKernelGGL((wave_select_kernel<true>), grid, dim3(256), (size_t)4 * kp * 8 + 4 * 256 * 8, stream, p, kp);
        else hipLaunchKernelGGL((wave_select_kernel<false>), grid, dim3(256), (size_t)4 * kp * 12 + 4 * 256 * 8, stream, p, kp);
        HIP_CHECK(hipGetLastError());
        return;
    }
    size_t lds = ((sizeof(SelShared) + 15) & ~(size_t)15) + (size_t)kp * (8 + 4);
    hipLaunchKernelGGL(select_k_kernel, dim3((unsigned)p.nq), dim3(SEL_THREADS), lds, stream, p);
    HIP_CHECK(hipGetLastError());
}

} // namespace faiss_amd
