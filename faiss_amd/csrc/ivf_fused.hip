// faiss_amd/csrc/ivf_fused.hip -- fused inverted-list search for gfx950: one workgroup walks the
// probed lists of one query, builds the PQ lookup table in LDS, scans the list codes against
// it and keeps the running top-k in an LDS reservoir.  Neither the table (64 KB per (query,
// probe) at M = 64) nor the per-code distances ever reach HBM; the only global traffic is
// the code stream itself plus k results per query.
//
// Replaces, in one launch, the reference chain
//   runPQCodeDistances        faiss/gpu/impl/PQCodeDistances-inl.cuh:29-285, 590-736  (table -> global)
//   pqScanNoPrecomputedMultiPass  faiss/gpu/impl/PQScanMultiPassNoPrecomputed-inl.cuh:173-270 (distances -> global)
//   runCalcListOffsets / runPass1SelectLists / runPass2SelectLists
//                             faiss/gpu/impl/IVFUtils.cu:131-186, IVFUtilsSelect1.cu, IVFUtilsSelect2.cu
// and, for IVFFlat, ivfInterleavedScan + ivfInterleavedScan2 (faiss/gpu/impl/IVFInterleaved.cuh:33-224,
// IVFInterleaved.cu:18-177).
//
// Arithmetic contract: identical to ivf_kernels.hip (restated by oracle/faiss_oracle.c):
//   IVFPQ   : lut[m][c] = chain_j fmaf(q_mj, pq[m][c][j], acc), rounded to the query's power-of-two grid
//             (kernels.h pq_lut_grid); S = sum_m lut[m][code_m] is then exact in fp32 in ANY order
//             L2: dis = fmaf(-2, S, coarse_l2 + t2(row));  IP: dis = coarse_ip + S
//   IVFFlat : dis = chain_k fmaf(q[k]-y[k], q[k]-y[k], acc) (L2) / fmaf(q[k], y[k], acc) (IP)
//   selection: k smallest keys (ordkey(dis) << 32 | position in probe order) = "first scanned wins"
//             among equal distances (faiss/IndexIVF.cpp:642-655 + strict heap admission); the k winners
//             are then ordered by (distance, id) (faiss/impl/ResultHandler.h:439-453).
#include "kernels.h"
#include "lmf_select.h"
#include "wg_select.h"

namespace faiss_amd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// threads per workgroup, a template parameter of the kernels (FB inside them): 1024 = 16 wavefronts, one workgroup per
// CU (IVFFlat: 256 KB of row loads in flight per iteration); 512 for IVFPQ, whose 64 KB table then leaves room for
// TWO workgroups per CU -- one query's table build / reservoir selects / final sort overlap the other's gathers
constexpr int FB_MAX = 1024;

// fp32 at an absolute LDS byte address (the lookup table starts at LDS address 0: the kernels here have no static
// LDS, checked once per workgroup) -- spares the "+ table base" VALU add per gather that pointer arithmetic costs
__device__ __forceinline__ float lds_f32(unsigned byte_addr) {
    return *(const __attribute__((address_space(3))) float*)(size_t)byte_addr;
}
// LDS byte address of table entry lut[c][m] (M = 64: c * 256 + 4 m) in ONE VALU instruction: v_perm_b32 puts code
// byte I of `w` (= c) into byte 1 and byte I of `rot` (= 4 m for this lane and step, precomputed once) into byte 0.
template <int I>
__device__ __forceinline__ unsigned lut_addr64(unsigned w, unsigned rot) {
    return __builtin_amdgcn_perm(w, rot, 0x0c0c0000u | ((4u + I) << 8) | (unsigned)I);
}
constexpr int FMAXR = 8;  // reservoir capacity <= FMAXR * FB keys
// IDSelector: bit `row` of the per-arena-row mask (launch_selector_mask).  Tested only for rows whose key beats the
// running threshold, i.e. after the distance is known: the mask (1 bit per row) stays in L2, the code stream is untouched.
__device__ __forceinline__ bool sel_bit(const uint32_t* __restrict__ mask, int64_t row) {
    return ((mask[row >> 5] >> (row & 31)) & 1u) != 0u;
}

struct FusedLds {
    char* lut;        // [M][256] fp32 | at the end: winners
    float* rs;        // [dpad] residual (L2) or query (IP)
    u64* res;         // [cap]
    uint32_t* pre;    // [nprobe + 1] exclusive prefix of the probed lists' lengths (scan positions)
    uint32_t* bpre;   // [nprobe + 1] exclusive prefix of their 64-row block counts (IVFPQ)
    unsigned* colmax; // [M] + 4: bits of max_c |lut[m][c]|, then the table grid {delta, 1/delta, on}
    int* lst;         // [nprobe]
    int64_t* lstart;  // [nprobe] first arena row of each probed list
    unsigned* hist;   // [256]
    WgSelCtl* ctl;
};

// region 0: the PQ lookup table(s) / the scalar quantizer's M table rows of dpad (= dsq) floats + one coarse distance
// per probe; reused for the winners once the scan is over
static size_t fused_region0_bytes(int kind, int M, int kp, int nlut, int dpad, int nprobe) {
    size_t lut = kind == 1 ? (size_t)M * 1024 * nlut : kind == 2 ? (size_t)M * dpad * 4 + (size_t)nprobe * 4 : 0;
    return round_up(std::max<size_t>(std::max<size_t>(lut, (size_t)kp * 12), 16), 16);
}
// (kind 2: dpad = the quantizer's padded dimension sq_dsq, M = table rows)
size_t ivf_fused_lds_bytes(int kind, int M, int dpad, int kp, int cap, int nprobe, int nlut) {
    return fused_region0_bytes(kind, M, kp, nlut, dpad, nprobe) + 2 * round_up((size_t)dpad * 4, 16) + (size_t)cap * 8 +
           2 * round_up((size_t)(nprobe + 1) * 4, 16) + round_up((size_t)(M + 4) * 4, 16) +
           round_up((size_t)nprobe * 4, 16) + (size_t)nprobe * 8 + 1024 + 64;
}

__device__ __forceinline__ FusedLds fused_carve(char* smem, const IvfFusedParams& p) {
    FusedLds L;
    size_t o = 0;
    L.lut = smem;
    const int dq = p.kind == 2 ? p.sq_dsq : p.dpad;
    {
        size_t lut = p.kind == 1 ? (size_t)p.M * 1024 * p.nlut
                   : p.kind == 2 ? (size_t)p.M * dq * 4 + (size_t)p.nprobe * 4 : 0;
        size_t r0 = lut > (size_t)p.kp * 12 ? lut : (size_t)p.kp * 12;
        if (r0 < 16) r0 = 16;
        o = (r0 + 15) & ~(size_t)15;
    }
    L.rs = (float*)(smem + o); // two buffers of rs_stride floats
    o += 2 * (((size_t)dq * 4 + 15) & ~(size_t)15);
    L.res = (u64*)(smem + o);
    o += (size_t)p.cap * 8;
    L.pre = (uint32_t*)(smem + o);
    o += ((size_t)(p.nprobe + 1) * 4 + 15) & ~(size_t)15;
    L.bpre = (uint32_t*)(smem + o);
    o += ((size_t)(p.nprobe + 1) * 4 + 15) & ~(size_t)15;
    L.colmax = (unsigned*)(smem + o);
    o += ((size_t)(p.M + 4) * 4 + 15) & ~(size_t)15;
    L.lst = (int*)(smem + o);
    o += ((size_t)p.nprobe * 4 + 15) & ~(size_t)15;
    L.lstart = (int64_t*)(smem + o);
    o += (size_t)p.nprobe * 8;
    L.hist = (unsigned*)(smem + o);
    o += 1024;
    L.ctl = (WgSelCtl*)(smem + o);
    return L;
}

// probed lists of query q -> LDS (list ids, exclusive prefix of their lengths)
template <int FB>
__device__ __forceinline__ void fused_load_probes(const IvfFusedParams& p, int q, const FusedLds& L) {
    const int tid = threadIdx.x;
    for (int t = tid; t < p.nprobe; t += FB) {
        const int64_t l = p.coarse_ids[(int64_t)q * p.nprobe + t];
        L.lst[t] = (int)l;
        if (p.probe_len) {
            L.pre[t + 1] = p.probe_len[(int64_t)q * p.nprobe + t];
            L.lstart[t] = p.probe_start[(int64_t)q * p.nprobe + t];
        } else {
            L.pre[t + 1] = l >= 0 ? p.list_len[l] : 0u;
            L.lstart[t] = l >= 0 ? p.list_start[l] : 0;
        }
    }
    for (int m = tid; m < p.M; m += FB) L.colmax[m] = 0u;
    if (tid == 0) {
        L.pre[0] = 0;
        L.bpre[0] = 0;
        L.ctl->cnt = 0;
    }
    __syncthreads();
    if (tid < 64) {
        unsigned carry = 0, bcarry = 0;
        for (int base = 0; base < p.nprobe; base += 64) {
            const int t = base + tid;
            unsigned v = t < p.nprobe ? L.pre[t + 1] : 0u;
            unsigned bv = (v + 63u) >> 6;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned o = __shfl_up(v, off, 64);
                const unsigned bo = __shfl_up(bv, off, 64);
                if (tid >= off) {
                    v += o;
                    bv += bo;
                }
            }
            v += carry;
            bv += bcarry;
            if (t < p.nprobe) {
                L.pre[t + 1] = v;
                L.bpre[t + 1] = bv;
            }
            carry = __shfl(v, 63, 64);
            bcarry = __shfl(bv, 63, 64);
        }
    }
    __syncthreads();
}

// Cut the reservoir back to its k smallest keys when the next chunk might not fit.  `bound` is
// a wave-uniform upper bound of the live count kept in a register, so the shared counter is
// only consulted (read by everybody, THEN a barrier, so that no early lane appends in between)
// when the bound says the reservoir could be full.
#define FUSED_MAKE_ROOM(chunk)                                                         \
    do {                                                                               \
        if (bound + FB > p.cap) {                                                      \
            const int n_ = (int)L.ctl->cnt;                                            \
            __syncthreads();                                                           \
            bound = n_;                                                                \
            if (n_ + FB > p.cap) {                                                     \
                const u64 kth_ = wg_select_kth<FB>(L.res, n_, p.k, L.hist, L.ctl);     \
                wg_compact<FB>(L.res, n_, kth_, L.ctl);                         \
                tau = kth_;                                                            \
                bound = p.k;                                                           \
            }                                                                          \
        }                                                                              \
        bound += (int)(chunk);                                                         \
    } while (0)

// The same for a loop WITHOUT a barrier per iteration (the wavefronts of the workgroup drift apart between two cuts):
// everybody's appends are complete before the counter is read, and nobody appends again before everybody has read it.
#define FUSED_MAKE_ROOM_DRIFT(chunk)                                                   \
    do {                                                                               \
        if (bound + FB > p.cap) {                                                      \
            __syncthreads();                                                           \
            const int n_ = (int)L.ctl->cnt;                                            \
            __syncthreads();                                                           \
            bound = n_;                                                                \
            if (n_ + FB > p.cap) {                                                     \
                const u64 kth_ = wg_select_kth<FB>(L.res, n_, p.k, L.hist, L.ctl);     \
                wg_compact<FB>(L.res, n_, kth_, L.ctl);                                \
                tau = kth_;                                                            \
                bound = p.k;                                                           \
            }                                                                          \
        }                                                                              \
        bound += (int)(chunk);                                                         \
    } while (0)

// final k-selection, position -> user id, ordering, write-out (or partial keys when G > 1)
template <int FB>
__device__ __forceinline__ void fused_finish(const IvfFusedParams& p, int q, int g, const FusedLds& L) {
    const int tid = threadIdx.x;
    __syncthreads();
    int n = (int)L.ctl->cnt;
    if (p.defer_finish) {
        // the reservoir as it is: the selection runs in its own launch at many workgroups per CU instead of on the
        // critical path of this one (33 k of 146 k cycles per workgroup at nb = 1M, tools/ivfpq_phases.py)
        u64* out = p.part_keys + (int64_t)q * p.cap;
        for (int i = tid; i < n; i += FB) out[i] = L.res[i];
        if (tid == 0) p.part_cnt[q] = (uint32_t)n;
        for (int t = tid; t <= p.nprobe; t += FB) p.prefix_out[(int64_t)q * (p.nprobe + 1) + t] = L.pre[t];
        return;
    }
    if (n > p.k) {
        const u64 kth = wg_select_kth<FB>(L.res, n, p.k, L.hist, L.ctl);
        wg_compact<FB>(L.res, n, kth, L.ctl);
        n = (int)L.ctl->cnt;
    }
    if (p.G > 1) {
        u64* out = p.part_keys + ((int64_t)q * p.G + g) * p.k;
        for (int i = tid; i < n; i += FB) out[i] = L.res[i];
        if (tid == 0) p.part_cnt[(int64_t)q * p.G + g] = (uint32_t)n;
        if (g == 0)
            for (int t = tid; t <= p.nprobe; t += FB) p.prefix_out[(int64_t)q * (p.nprobe + 1) + t] = L.pre[t];
        return;
    }
    int64_t* w_id = (int64_t*)L.lut;
    unsigned* w_key = (unsigned*)(w_id + p.kp);
    for (int i = tid; i < p.kp; i += FB) {
        unsigned wk = 0xffffffffu;
        int64_t wi = INT64_MAX;
        if (i < n) {
            const u64 key = L.res[i];
            const uint32_t payload = (uint32_t)key;
            int lo = 0, hi = p.nprobe; // invariant pre[lo] <= payload < pre[hi]
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (L.pre[mid] <= payload) lo = mid;
                else hi = mid;
            }
            wi = p.arena_ids[L.lstart[lo] + (payload - L.pre[lo])];
            wk = (uint32_t)(key >> 32);
        }
        w_key[i] = wk;
        w_id[i] = wi;
    }
    __syncthreads();
    const float pad = neutral_distance(p.metric);
    if (p.kp <= 256) {
        // n <= 256 winners: every thread ranks one of them against all others (broadcast LDS reads) and writes it
        // straight to its place -- no barrier ladder of a sorting network
        for (int i = tid; i < p.k; i += FB) {
            if (i < n) {
                const unsigned ki = w_key[i];
                const int64_t ii = w_id[i];
                int rank = 0;
                for (int j2 = 0; j2 < n; ++j2) {
                    const unsigned kj = w_key[j2];
                    const int64_t ij = w_id[j2];
                    // (duplicate user ids are legal: the slot index breaks the last tie)
                    rank += (kj < ki || (kj == ki && (ij < ii || (ij == ii && j2 < i)))) ? 1 : 0;
                }
                const bool ok = ki < kInvalidOrdKey;
                p.out_dis[(int64_t)q * p.k + rank] = ok ? unordkey_rt(p.metric, ki) : pad;
                p.out_ids[(int64_t)q * p.k + rank] = ok ? ii : -1;
            } else {
                p.out_dis[(int64_t)q * p.k + i] = pad;
                p.out_ids[(int64_t)q * p.k + i] = -1;
            }
        }
        return;
    }
    wg_bitonic_sort<FB>(w_key, w_id, p.kp);
    for (int i = tid; i < p.k; i += FB) {
        float dis = pad;
        int64_t id = -1;
        if (i < n && w_key[i] < kInvalidOrdKey) {
            dis = unordkey_rt(p.metric, w_key[i]);
            id = w_id[i];
        }
        p.out_dis[(int64_t)q * p.k + i] = dis;
        p.out_ids[(int64_t)q * p.k + i] = id;
    }
}

// ---------------------------------------------------------------------------------
// IVFPQ.  One lookup table per QUERY, resident in LDS for all of its probes:
//   lut[c][m] = <q_m, pq[m][c]> rounded to the query's grid (pq_lut_grid)        (both metrics)
//   IP :  dis = coarse_ip(q, list) + S               S = sum_m lut[code_m][m]  (exact, order-free)
//   L2 :  |q - c - r^|^2 = |q - c|^2 + (|r^|^2 + 2 <c, r^>) - 2 <q, r^>
//         dis = fmaf(-2, S, coarse_l2(q, list) + t2(y))
// with r^ the decoded residual of the stored vector and t2(y) = |r^|^2 + 2 <c, r^> a per-vector
// constant computed once at add time (ivfpq_t2 kernels).  This is the decomposition the reference
// CPU index uses by default (faiss/impl/pq_code_distance/IVFPQ_QueryTables.cpp:126-192 term 1-3,
// use_precomputed_table) and the reference GPU index offers as usePrecomputedTables
// (faiss/gpu/impl/IVFPQ.cu:362-489); keeping the list-dependent term per VECTOR instead of per
// (list, m, code) costs 4 bytes per vector and removes the 256 MB term-2 table and, above all, the
// rebuild of a 64 KB table for each of the 32 probes of a query.
//
// Scan: the probed lists are a stream of 64-row code BLOCKS (kernels.h pq_code_offset); every wavefront takes one
// block per iteration, lane l = row l of the block.  The block layout hands lane l its code rotated by l, and the
// table is laid out [256][M]: at step j the 32 lanes of an LDS access group read 32 different sub-quantizers =
// 32 different banks, so the 64 gathers per code are conflict-free (round 1: [M][256] table, every lane the same
// sub-quantizer, random codes -> 3.5-way conflicts, 60 % of all LDS cycles, profiles/r02_a_*).  The table grid
// makes the sum exact, so the rotated order changes nothing.
// M64: the sub-quantizer count is the compile-time constant 64 (four coalesced 1 KB loads per block; table address
// of a gather = ONE v_perm_b32).
// ---------------------------------------------------------------------------------
template <int METRIC, bool M64, int FB, bool SEL>
__global__ void __launch_bounds__(FB, 4) ivfpq_fused_kernel(IvfFusedParams p) { // 4 waves per SIMD: two 512-thread workgroups per CU
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const FusedLds L = fused_carve(smem, p);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NWV = FB / 64;
    const int q = blockIdx.x / p.G, g = blockIdx.x - q * p.G;
    const int p0 = g * p.npc, p1 = min(p.nprobe, p0 + p.npc);
    float* lut = (float*)L.lut;
    const int M = M64 ? 64 : p.M, d = p.d, dsub = p.dsub;
    float* grid = (float*)(L.colmax + M);

    long long t_ph = p.phase_ticks ? (long long)__builtin_readcyclecounter() : 0;
    auto phase_mark = [&](int ph) {
        if (p.phase_ticks && tid == 0) {
            const long long now = (long long)__builtin_readcyclecounter();
            atomicAdd(&p.phase_ticks[ph], (unsigned long long)(now - t_ph));
            t_ph = now;
        }
    };
    // (the query row is requested before the probe tables: its latency overlaps theirs; so is the first half of this
    // lane's codebook entries, which depend on nothing)
    const float q_first = tid < d ? p.xq[(int64_t)q * p.ldq + tid] : 0.f;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    constexpr int NE = 16384 / FB; // table entries per lane (M64, dsub == 2)
    const bool fast_build = M64 && dsub == 2;
    f32x2 cb0[NE / 2];
    if (fast_build) {
#pragma unroll
        for (int u = 0; u < NE / 2; ++u) cb0[u] = *(const f32x2*)(p.pq_t + (size_t)(tid + u * FB) * 2);
    }
    fused_load_probes<FB>(p, q, L);
    if (tid < d) L.rs[tid] = q_first;
    for (int cc = tid + FB; cc < d; cc += FB) L.rs[cc] = p.xq[(int64_t)q * p.ldq + cc];
    __syncthreads();
    // ---- block stream of this workgroup's probes [p0, p1): blocks [bpre[p0], bpre[p1])
    const unsigned blk_begin = L.bpre[p0], blk_end = L.bpre[p1];
    constexpr int NW = 4; // 16-byte code words per row (M64)
    // one block of one wavefront on its way from HBM to the gathers: row `lane` of the block
    struct Stage {
        uint4 w[NW];        // code words (M64)
        float t2;           // per-row L2 term
        // wave-uniform (kept in scalar registers): coarse term of the block's list, scan position of the block's
        // first row, rows of the list from that row on (0 = no block)
        float dis0;
        unsigned pos0, rem;
        const uint8_t* bp;  // block base (generic M: codes are read in the gather loop)
        int64_t row0;       // SEL: arena row of the block's first row
    };
    int tcur = p0; // probe of the block fetched last (blocks are visited in increasing order)
    auto fetch = [&](unsigned blk, Stage& st) {
        st.rem = 0;
        if (blk < blk_end) { // wave-uniform
            while (L.bpre[tcur + 1] <= blk) ++tcur;
            const unsigned b = blk - L.bpre[tcur];
            const int64_t row0 = L.lstart[tcur] + (int64_t)b * 64;
            st.bp = p.arena_codes + row0 * M;
            if (SEL) st.row0 = row0;
            st.rem = __builtin_amdgcn_readfirstlane(L.pre[tcur + 1] - L.pre[tcur] - b * 64u);
            // only the rows the list really holds are requested: the last block of a list is on average half empty,
            // 13 % of the code traffic at nb / nlist = 244 rows per list
            if ((unsigned)lane < st.rem) {
                if (M64) {
#pragma unroll
                    for (int k = 0; k < NW; ++k) st.w[k] = *(const uint4*)(st.bp + k * 1024 + lane * 16);
                }
                if (METRIC == METRIC_L2) st.t2 = p.arena_t2[row0 + lane];
            }
            st.dis0 = __uint_as_float(__builtin_amdgcn_readfirstlane(
                    __float_as_uint(p.coarse_dis[(int64_t)q * p.nprobe + tcur])));
            st.pos0 = __builtin_amdgcn_readfirstlane(L.pre[tcur] + b * 64u);
        }
    };
    // the first block of every wavefront is requested BEFORE the table is built: its HBM latency hides behind the
    // codebook reads and the rounding
    Stage s0, s1;
    fetch(blk_begin + wave, s0);
    phase_mark(0);
    // ---- the query's table (transposed codebook read through L2 once per query), its grid, the rounding
    const int ne = M * 256;
    auto publish_grid = [&]() {
        // B = sum_m max_c |lut[m][c]| in sub-quantizer order (the oracle's order), by one lane
        if (tid == 0) {
            float B = 0.f;
            for (int m = 0; m < M; ++m) B = B + __uint_as_float(L.colmax[m]);
            float delta = 0.f, inv = 0.f;
            const bool on = pq_lut_grid(B, &delta, &inv);
            grid[0] = delta;
            grid[1] = inv;
            grid[2] = on ? 1.f : 0.f;
        }
    };
    if (fast_build) {
        // entries e = tid + u * FB of the [256][64] table: sub-quantizer e & 63 = tid & 63 for all of them (FB is a
        // multiple of 64), so the query slice and the running maximum stay in registers
        const float r0 = L.rs[2 * (tid & 63)], r1 = L.rs[2 * (tid & 63) + 1];
        float v[NE];
        float mx = 0.f;
        // the codebook entries of this lane arrive in two waves of NE / 2 loads (32 + 32 live registers instead of
        // 96): the first went out at kernel entry, the second goes out now and lands while the first is consumed
        f32x2 cb1[NE / 2];
#pragma unroll
        for (int u = 0; u < NE / 2; ++u) cb1[u] = *(const f32x2*)(p.pq_t + (size_t)(tid + (NE / 2 + u) * FB) * 2);
#pragma unroll
        for (int u = 0; u < NE; ++u) {
            const f32x2 c = u < NE / 2 ? cb0[u] : cb1[u - NE / 2];
            v[u] = __fmaf_rn(r1, c[1], __fmaf_rn(r0, c[0], 0.f));
            const float a = fabsf(v[u]);
            mx = (a > mx || a != a) ? a : mx; // NaN sticks
        }
        atomicMax(&L.colmax[tid & 63], __float_as_uint(mx));
        __syncthreads();
        publish_grid();
        __syncthreads();
        const float delta = grid[0], inv = grid[1];
        const bool on = grid[2] != 0.f;
#pragma unroll
        for (int u = 0; u < NE; ++u) lut[tid + u * FB] = on ? __builtin_rintf(v[u] * inv) * delta : v[u];
    } else {
        for (int e = tid; e < ne; e += FB) {
            const int m = e % M;
            const float* cen = p.pq_t + (size_t)e * dsub;
            const float* r = L.rs + m * dsub;
            float acc = 0.f;
            for (int jd = 0; jd < dsub; ++jd) acc = __fmaf_rn(r[jd], cen[jd], acc);
            lut[e] = acc;
            atomicMax(&L.colmax[m], __float_as_uint(fabsf(acc)));
        }
        __syncthreads();
        publish_grid();
        __syncthreads();
        if (grid[2] != 0.f) {
            const float delta = grid[0], inv = grid[1];
            for (int e = tid; e < ne; e += FB) lut[e] = __builtin_rintf(lut[e] * inv) * delta;
        }
    }
    __syncthreads();

    // per-lane table columns: rot[kk] byte i = 4 * ((4 kk + i + lane) mod 64) for dword kk of the code
    unsigned rot[16];
    if (M64) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            unsigned r = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) r |= ((4u * (unsigned)(4 * k + i + lane)) & 255u) << (8 * i);
            rot[k] = r;
        }
        if ((unsigned)(size_t)(const __attribute__((address_space(3))) char*)lut != 0u) __builtin_trap(); // see lds_f32
    }

    phase_mark(1);
    u64 tau = ~0ull;
    int bound = 0;
    // gathers + key + append of one staged block
    auto scan = [&](const Stage& st) {
        bool pass = false;
        u64 key = 0;
        if ((unsigned)lane < st.rem) {
            float sum;
            if (M64) {
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
                for (int k = 0; k < NW; ++k) {
                    const unsigned w4[4] = {st.w[k].x, st.w[k].y, st.w[k].z, st.w[k].w};
#pragma unroll
                    for (int wd = 0; wd < 4; ++wd) {
                        const unsigned ro = rot[4 * k + wd];
                        a0 += lds_f32(lut_addr64<0>(w4[wd], ro));
                        a1 += lds_f32(lut_addr64<1>(w4[wd], ro));
                        a2 += lds_f32(lut_addr64<2>(w4[wd], ro));
                        a3 += lds_f32(lut_addr64<3>(w4[wd], ro));
                    }
                }
                sum = (a0 + a1) + (a2 + a3);
            } else {
                const int ch = pq_chunk_bytes(M);
                sum = 0.f;
                for (int j = 0; j < M; ++j) {
                    const unsigned c = st.bp[(j / ch) * 64 * ch + lane * ch + (j % ch)];
                    int m = j + lane;
                    m -= (m / M) * M;
                    sum += lut[c * M + m];
                }
            }
            const float dis = METRIC == METRIC_L2 ? __fmaf_rn(-2.f, sum, st.dis0 + st.t2) : st.dis0 + sum;
            key = ((u64)ordkey<METRIC>(dis) << 32) | (u64)(st.pos0 + (unsigned)lane);
            pass = key < tau;
            if (SEL) {
                if (pass) pass = sel_bit(p.sel_mask, st.row0 + lane);
            }
        }
        wg_append(L.res, L.ctl, pass, key);
    };
    // Two stages per wavefront: while block i is gathered, block i + 1 is in flight.  (A third stage was measured
    // and dropped: no gain -- with both workgroups of a CU scanning the kernel already moves ~5.2 TB/s of actual traffic,
    // the fabric's limit -- and it needed the table columns recomputed per gather to fit 128 registers.)
    unsigned base = blk_begin;
    for (;;) {
        if (base >= blk_end) break;
        FUSED_MAKE_ROOM(FB);
        fetch(base + NWV + wave, s1);
        scan(s0);
        __syncthreads();
        base += NWV;
        if (base >= blk_end) break;
        FUSED_MAKE_ROOM(FB);
        fetch(base + NWV + wave, s0);
        scan(s1);
        __syncthreads();
        base += NWV;
    }
    phase_mark(2);
    fused_finish<FB>(p, q, g, L);
    phase_mark(3);
    if (p.phase_ticks && tid == 0) atomicAdd(&p.phase_ticks[4], 1ull);
}

// ---------------------------------------------------------------------------------
// IVFFlat: same reservoir machinery, exact distances straight from the fp32 rows of the lists.
// The probed lists of a query are treated as ONE array of rows addressed by scan position (the
// payload of the keys): every iteration takes the next 512 positions whatever list they fall in
// (position -> (probe, offset) by a 5-step search of the prefix table in LDS).  Eight adjacent
// lanes share a row: lane j reads the 16-byte chunks j, j+8, ... (each load instruction covers
// one full 128-byte segment per row), keeps a partial fmaf chain over them and the eight partial
// sums meet in an xor butterfly (order restated by oracle/faiss_oracle.c).  With 4 rows per lane
// group in flight a 1024-thread workgroup keeps 256 KB of loads outstanding per iteration.
// HBM-bound: nprobe * (nb / nlist) * d * 4 bytes per query, no reuse across queries.
// ---------------------------------------------------------------------------------
constexpr int FF_ROWS = 4;                   // rows per 8-lane group and iteration
constexpr int FF_POS = FF_ROWS * (FB_MAX / 8); // positions per iteration (512)
constexpr int FF_QCH = 4;                    // query chunks kept in registers per lane (dpad <= 128)

template <int METRIC, bool SEL>
__global__ void __launch_bounds__(FB_MAX) ivfflat_fused_kernel(IvfFusedParams p) {
    constexpr int FB = FB_MAX;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const FusedLds L = fused_carve(smem, p);
    const int tid = threadIdx.x;
    const int q = blockIdx.x / p.G, g = blockIdx.x - q * p.G;
    const int p0 = g * p.npc, p1 = min(p.nprobe, p0 + p.npc);
    const int ln = tid & 7;   // lane inside the row group
    const int rg = tid >> 3;  // row group 0..127

    fused_load_probes<FB>(p, q, L);
    for (int cc = tid; cc < p.dpad; cc += FB) L.rs[cc] = p.xq[(int64_t)q * p.ldq + cc];
    __syncthreads();
    const int nch = p.dpad >> 2;               // 16-byte chunks per row
    const bool qreg = nch <= 8 * FF_QCH;       // this lane's query chunks fit the register copy
    f32x4 qv[FF_QCH];
#pragma unroll
    for (int t = 0; t < FF_QCH; ++t) {
        const int c4 = ln + 8 * t;
        qv[t] = c4 < nch ? *(const f32x4*)(L.rs + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }

    const unsigned pos_begin = L.pre[p0], pos_end = L.pre[p1];
    u64 tau = ~0ull;
    int bound = 0;
    for (unsigned base = pos_begin; base < pos_end; base += FF_POS) {
        FUSED_MAKE_ROOM(min((unsigned)FF_POS, pos_end - base));
        float part[FF_ROWS];
        unsigned posv[FF_ROWS];
        const float* rowp[FF_ROWS];
        int64_t rowi[FF_ROWS]; // SEL: arena row index
        // ---- rows of this iteration: position -> arena row
#pragma unroll
        for (int u = 0; u < FF_ROWS; ++u) {
            const unsigned pos = base + u * (FB / 8) + rg;
            posv[u] = pos;
            rowp[u] = nullptr;
            if (pos < pos_end) {
                int lo = p0, hi = p1; // invariant pre[lo] <= pos < pre[hi]
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (L.pre[mid] <= pos) lo = mid;
                    else hi = mid;
                }
                const int64_t row = L.lstart[lo] + (pos - L.pre[lo]);
                rowp[u] = p.arena_vecs + row * p.ldv;
                if (SEL) rowi[u] = row;
            }
        }
        // ---- partial chains
        if (qreg) {
            f32x4 yv[FF_ROWS][FF_QCH];
#pragma unroll
            for (int u = 0; u < FF_ROWS; ++u)
#pragma unroll
                for (int t = 0; t < FF_QCH; ++t) {
                    const int c4 = ln + 8 * t;
                    yv[u][t] = (rowp[u] && c4 < nch) ? *(const f32x4*)(rowp[u] + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
            for (int u = 0; u < FF_ROWS; ++u) {
                float a = 0.f;
#pragma unroll
                for (int t = 0; t < FF_QCH; ++t) {
                    if (ln + 8 * t < nch) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (METRIC == METRIC_L2) {
                                const float tt = qv[t][e] - yv[u][t][e];
                                a = __fmaf_rn(tt, tt, a);
                            } else {
                                a = __fmaf_rn(qv[t][e], yv[u][t][e], a);
                            }
                        }
                    }
                }
                part[u] = a;
            }
        } else {
#pragma unroll
            for (int u = 0; u < FF_ROWS; ++u) {
                float a = 0.f;
                if (rowp[u]) {
                    for (int c4 = ln; c4 < nch; c4 += 8) {
                        const f32x4 y4 = *(const f32x4*)(rowp[u] + 4 * c4);
                        const f32x4 q4 = *(const f32x4*)(L.rs + 4 * c4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (METRIC == METRIC_L2) {
                                const float tt = q4[e] - y4[e];
                                a = __fmaf_rn(tt, tt, a);
                            } else {
                                a = __fmaf_rn(q4[e], y4[e], a);
                            }
                        }
                    }
                }
                part[u] = a;
            }
        }
        // ---- ((p0+p1)+(p2+p3)) + ((p4+p5)+(p6+p7)) in every lane of the group, append from lane 0
#pragma unroll
        for (int u = 0; u < FF_ROWS; ++u) {
            float a = part[u];
            a = a + __shfl_xor(a, 1, 64);
            a = a + __shfl_xor(a, 2, 64);
            a = a + __shfl_xor(a, 4, 64);
            const u64 key = ((u64)ordkey<METRIC>(a) << 32) | (u64)posv[u];
            bool pass = rowp[u] != nullptr && ln == 0 && key < tau;
            if (SEL) {
                if (pass) pass = sel_bit(p.sel_mask, rowi[u]);
            }
            wg_append(L.res, L.ctl, pass, key);
        }
        __syncthreads();
    }
    fused_finish<FB>(p, q, g, L);
}

// ---------------------------------------------------------------------------------
// IVF scalar quantizer (kind 2).  The block stream of the IVFPQ scan over rows of CODES: the arena is a sequence of
// 64-row blocks (lists start on block boundaries), a block holds its rows chunk-major -- [chunk][64 rows][chunk bytes],
// a chunk = 16 components = 16 bytes of 8-bit codes / 8 of 4-bit / 12 of 6-bit / 32 of fp16 (kernels.h sq_code_offset)
// -- so that a wavefront reads one chunk of all 64 rows with ONE coalesced load instruction and lane l owns row l:
// no cross-lane reduction, no per-row address search (one prefix lookup per block and wavefront).  A lane decodes its
// chunk in registers and folds it into the distance with the per-dimension scale s and a query-side table row a,
// both read from LDS as wave-wide broadcasts (every row of a block belongs to the same list):
//   L2: tt = fmaf(-code, s_i, a_i), a_i = (q_i [- centroid_i]) - b_i   (fp16: tt = a_i - half)   acc = fmaf(tt, tt, acc)
//   IP: acc = fmaf(w_i, code, acc), w_i = q_i * s_i                     (fp16: w_i = q_i);  + <q, b> (+ coarse term)
// as two sequential chains, one over the even and one over the odd dimensions (packed fp32 math), added at the end:
// the distance to the reconstruction b_i + s_i * code_i of
// faiss::ScalarQuantizer (quantizers.h:92-150: vmin + (code + 0.5) / 255 * vdiff) without materialising it.  With
// residual encoding the L2 row a changes with the list: one row per probe of the workgroup (built once per query,
// nprobe x d x 4 bytes).  Two stages per wavefront: the next (block, chunk group) is in flight while the current one is
// folded; rows longer than a stage (32 registers: 128 8-bit components) take several chunk groups per block.
// The code stream is the only HBM traffic: d bytes per scanned vector for 8-bit codes.
// Reference: IVFSQScannerL2 / IVFSQScannerIP (faiss/impl/scalar_quantizer/scanners.h:34-140), on the GPU
// IVFFlatScan with a Codec (faiss/gpu/impl/IVFFlatScan.cu, GpuScalarQuantizer.cuh).
// ---------------------------------------------------------------------------------
constexpr int SQ_FB = 512;
template <int CT>
struct SqChunk {
    static constexpr int WORDS = CT == SQ_U8 ? 4 : CT == SQ_U4 ? 2 : CT == SQ_U6 ? 3 : 8;
    static constexpr int CG = 32 / WORDS > 8 ? 8 : 32 / WORDS; // chunks per stage (<= 32 registers)
};
// component E (0..15) of a chunk, as a float
template <int CT, int E>
__device__ __forceinline__ float sq_comp(const unsigned (&w)[SqChunk<CT>::WORDS]) {
    if constexpr (CT == SQ_U8) {
        return (float)((w[E >> 2] >> (8 * (E & 3))) & 255u);
    } else if constexpr (CT == SQ_U4) {
        return (float)((w[E >> 3] >> (4 * (E & 7))) & 15u);
    } else if constexpr (CT == SQ_U6) {
        constexpr int off = 6 * E, wi = off >> 5, sh = off & 31;
        if constexpr (sh <= 26) return (float)((w[wi] >> sh) & 63u);
        else return (float)(((w[wi] >> sh) | (w[wi + 1] << (32 - sh))) & 63u);
    } else {
        const unsigned short hw = (unsigned short)(w[E >> 1] >> (16 * (E & 1)));
        return (float)__builtin_bit_cast(_Float16, hw);
    }
}
// components E, E + 1 of a chunk folded into the (even, odd) pair of chains with packed fp32 math (v_pk_fma_f32: two
// IEEE fmas per instruction): acc[0] runs over the even dimensions, acc[1] over the odd ones
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int METRIC, int CT, int E>
__device__ __forceinline__ void sq_fold(const unsigned (&w)[SqChunk<CT>::WORDS], const float* sv, const float* av, f32x2& acc) {
    if constexpr (E < 16) {
        const f32x2 cf = {sq_comp<CT, E>(w), sq_comp<CT, E + 1>(w)};
        const f32x2 a2 = {av[E], av[E + 1]};
        if (METRIC == METRIC_L2) {
            f32x2 tt;
            if (CT == SQ_F16) {
                tt = a2 - cf;
            } else {
                const f32x2 s2 = {sv[E], sv[E + 1]};
                tt = __builtin_elementwise_fma(-cf, s2, a2);
            }
            acc = __builtin_elementwise_fma(tt, tt, acc);
        } else {
            acc = __builtin_elementwise_fma(a2, cf, acc);
        }
        sq_fold<METRIC, CT, E + 2>(w, sv, av, acc);
    }
}

template <int METRIC, int CT, bool SEL>
__global__ void __launch_bounds__(SQ_FB, 4) ivfsq_fused_kernel(IvfFusedParams p) { // two 512-thread workgroups per CU
    constexpr int FB = SQ_FB;
    constexpr int NWV = FB / 64;
    constexpr int W = SqChunk<CT>::WORDS;
    constexpr int CG = SqChunk<CT>::CG;
    constexpr int CHB = W * 4; // bytes per chunk
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const FusedLds L = fused_carve(smem, p);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = blockIdx.x / p.G, g = blockIdx.x - q * p.G;
    const int p0 = g * p.npc, p1 = min(p.nprobe, p0 + p.npc);
    const int dsq = p.sq_dsq, nch = dsq >> 4;
    const int ngrp = (nch + CG - 1) / CG;
    const bool per_probe = METRIC == METRIC_L2 && p.sq_by_residual;
    float* tab_s = (float*)L.lut;
    float* tab_a = tab_s + dsq;
    float* tab_c = tab_a + (size_t)(p.M - 1) * dsq; // [npc] coarse distances (IP with residual encoding)

    fused_load_probes<FB>(p, q, L);
    // ---- block stream of this workgroup's probes [p0, p1): blocks [bpre[p0], bpre[p1])
    const unsigned blk_begin = L.bpre[p0], blk_end = L.bpre[p1];
    struct Stage {
        unsigned w[CG][W];  // code words of this lane's row, chunks grp * CG ..
        // wave-uniform: probe (relative to p0) of the block's list, scan position of the block's first row, rows of the
        // list from that row on (0 = no block)
        int t;
        unsigned pos0, rem;
        int64_t row0; // SEL: arena row of the block's first row
    };
    int tcur = p0; // probe of the block fetched last (blocks are visited in non-decreasing order)
    auto fetch = [&](unsigned blk, int grp, Stage& st) {
        st.rem = 0;
        if (blk < blk_end) { // wave-uniform
            while (L.bpre[tcur + 1] <= blk) ++tcur;
            const unsigned b = blk - L.bpre[tcur];
            const int64_t row0 = L.lstart[tcur] + (int64_t)b * 64;
            const uint8_t* bp = p.arena_codes + row0 * (int64_t)p.sq_ld + lane * CHB;
            if (SEL) st.row0 = row0;
            st.rem = __builtin_amdgcn_readfirstlane(L.pre[tcur + 1] - L.pre[tcur] - b * 64u);
            st.t = __builtin_amdgcn_readfirstlane(tcur - p0);
            st.pos0 = __builtin_amdgcn_readfirstlane(L.pre[tcur] + b * 64u);
            // only the rows the list really holds are requested
            if ((unsigned)lane < st.rem) {
#pragma unroll
                for (int j = 0; j < CG; ++j) {
                    const int c = grp * CG + j;
                    if (c < nch) { // wave-uniform
                        const unsigned* src = (const unsigned*)(bp + (int64_t)c * 64 * CHB);
                        if constexpr (W == 4) {
                            const uint4 v = *(const uint4*)src;
                            st.w[j][0] = v.x, st.w[j][1] = v.y, st.w[j][2] = v.z, st.w[j][3] = v.w;
                        } else if constexpr (W == 2) {
                            const uint2 v = *(const uint2*)src;
                            st.w[j][0] = v.x, st.w[j][1] = v.y;
                        } else if constexpr (W == 3) {
#pragma unroll
                            for (int i = 0; i < 3; ++i) st.w[j][i] = src[i];
                        } else {
                            const uint4 v0 = *(const uint4*)src, v1 = *(const uint4*)(src + 4);
                            st.w[j][0] = v0.x, st.w[j][1] = v0.y, st.w[j][2] = v0.z, st.w[j][3] = v0.w;
                            st.w[j][4] = v1.x, st.w[j][5] = v1.y, st.w[j][6] = v1.z, st.w[j][7] = v1.w;
                        }
                    }
                }
            }
        }
    };
    // the first stage of every wavefront is requested BEFORE the tables are built: its HBM latency hides behind them
    Stage s0, s1;
    fetch(blk_begin + wave, 0, s0);

    // ---- tables
    const float* xq = p.xq + (int64_t)q * p.ldq;
    for (int i = tid; i < dsq; i += FB) tab_s[i] = CT == SQ_F16 ? 0.f : p.sq_s[i];
    if (per_probe) {
        const int cnt = (p1 - p0) * dsq;
        for (int idx = tid; idx < cnt; idx += FB) {
            const int t = idx / dsq, i = idx - t * dsq;
            const int l = L.lst[p0 + t];
            float a = 0.f;
            if (i < p.d && l >= 0) {
                a = xq[i] - p.centroids[(int64_t)l * p.ldc + i];
                if (CT != SQ_F16) a = a - p.sq_b[i];
            }
            tab_a[idx] = a;
        }
    } else {
        for (int i = tid; i < dsq; i += FB) {
            float a = 0.f;
            if (i < p.d) {
                if (METRIC == METRIC_L2) a = CT == SQ_F16 ? xq[i] : xq[i] - p.sq_b[i];
                else a = CT == SQ_F16 ? xq[i] : xq[i] * p.sq_s[i];
            }
            tab_a[i] = a;
        }
    }
    if (METRIC != METRIC_L2) {
        for (int t = tid; t < p1 - p0; t += FB) tab_c[t] = p.sq_by_residual ? p.coarse_dis[(int64_t)q * p.nprobe + p0 + t] : 0.f;
        if (tid == 0) { // <q, b>: one sequential chain
            float acc = 0.f;
            if (CT != SQ_F16)
                for (int i = 0; i < p.d; ++i) acc = __fmaf_rn(xq[i], p.sq_b[i], acc);
            L.rs[0] = acc;
        }
    }
    __syncthreads();
    const float qb = METRIC != METRIC_L2 ? L.rs[0] : 0.f;

    u64 tau = ~0ull;
    int bound = 0;
    f32x2 acc = {0.f, 0.f}; // this lane's row (even / odd dimensions), carried over the chunk groups of a block
    // fold one staged chunk group; behind the last group of a block: key + append
    auto scan = [&](const Stage& st, int grp) {
        if (grp == 0) acc = f32x2{0.f, 0.f};
        if ((unsigned)lane < st.rem) {
            const float* arow = tab_a + (per_probe ? st.t * dsq : 0);
#pragma unroll
            for (int j = 0; j < CG; ++j) {
                const int c = grp * CG + j;
                if (c < nch) { // wave-uniform
                    float sv[16], av[16];
                    const f32x4* sp = (const f32x4*)(tab_s + 16 * c);
                    const f32x4* ap = (const f32x4*)(arow + 16 * c);
#pragma unroll
                    for (int v4 = 0; v4 < 4; ++v4) { // (wave-wide broadcast reads)
                        const f32x4 x4 = ap[v4];
                        av[4 * v4] = x4[0], av[4 * v4 + 1] = x4[1], av[4 * v4 + 2] = x4[2], av[4 * v4 + 3] = x4[3];
                        if (METRIC == METRIC_L2 && CT != SQ_F16) {
                            const f32x4 s4 = sp[v4];
                            sv[4 * v4] = s4[0], sv[4 * v4 + 1] = s4[1], sv[4 * v4 + 2] = s4[2], sv[4 * v4 + 3] = s4[3];
                        }
                    }
                    sq_fold<METRIC, CT, 0>(st.w[j], sv, av, acc);
                }
            }
        }
        if (grp == ngrp - 1) {
            bool pass = false;
            u64 key = 0;
            if ((unsigned)lane < st.rem) {
                float dis = acc[0] + acc[1];
                if (METRIC != METRIC_L2) dis = (dis + qb) + tab_c[st.t];
                key = ((u64)ordkey<METRIC>(dis) << 32) | (u64)(st.pos0 + (unsigned)lane);
                pass = key < tau;
                if (SEL) {
                    if (pass) pass = sel_bit(p.sel_mask, st.row0 + lane);
                }
            }
            wg_append(L.res, L.ctl, pass, key);
        }
    };
    // work item k of a wavefront: block blk_begin + wave + NWV * (k / ngrp), chunk group k % ngrp; all wavefronts of the
    // workgroup walk the same item sequence (the reservoir cuts are workgroup-wide), without a barrier per item
    unsigned base = blk_begin;
    int grp = 0;
    auto next_item = [&](unsigned& nb, int& ng) {
        ng = grp + 1;
        nb = base;
        if (ng == ngrp) {
            ng = 0;
            nb = base + NWV;
        }
    };
    for (;;) {
        if (base >= blk_end) break;
        {
            if (grp == 0) FUSED_MAKE_ROOM_DRIFT(FB);
            unsigned nb;
            int ng;
            next_item(nb, ng);
            fetch(nb + wave, ng, s1);
            scan(s0, grp);
            base = nb, grp = ng;
        }
        if (base >= blk_end) break;
        {
            if (grp == 0) FUSED_MAKE_ROOM_DRIFT(FB);
            unsigned nb;
            int ng;
            next_item(nb, ng);
            fetch(nb + wave, ng, s0);
            scan(s1, grp);
            base = nb, grp = ng;
        }
    }
    fused_finish<FB>(p, q, g, L);
}

// ---------------------------------------------------------------------------------
// Rerank of the list-major scan behind the f16 filter, scalar quantizer (ivf_lm_filter.hip, round 5): the exact distance of
// every collected candidate with the arithmetic of ivfsq_fused_kernel above -- the same sq_comp / sq_fold, the same table
// entries (a_i, s_i), the same two chains over the even and the odd dimensions -- so that a large batch returns, bit for
// bit, what the query-major scan returns.  One workgroup per query, one LANE per candidate (the query-major kernel gives a
// lane a row too); the table entries of a candidate's probe are recomputed per lane from the query, the centroid row and
// the decoder tables (all cache resident).  Components beyond d: a = 0 and s = 0 there, tt * tt adds exactly nothing.
// ---------------------------------------------------------------------------------
template <int METRIC, int CT>
__global__ void __launch_bounds__(256) lmf_rerank_sq_kernel(IvfLmParams p) {
    constexpr int W = SqChunk<CT>::WORDS;
    constexpr int CHB = W * 4;
    __shared__ float s_qb;
    __shared__ u64 sel_k[kLmfFusedSelectN];
    __shared__ uint32_t sel_wk[kLmfFusedSelectK];
    __shared__ int64_t sel_wl[kLmfFusedSelectK];
    const bool fin = p.fin_dis != nullptr;
    const int q = blockIdx.x, tid = threadIdx.x;
    const int np = p.nprobe;
    int n = (int)min((int64_t)p.cnt[q], p.stride);
    if (fin) n = min(n, kLmfFusedSelectN); // (more: the tightening launch listed the query for the redo)
    u64* kq = p.keys + (int64_t)q * p.stride;
    const uint16_t* cpr = p.cand_pr + (int64_t)q * p.stride;
    const float* xq = p.xq + (int64_t)q * p.ldq;
    const int nch = (p.d + 15) >> 4;
    const bool per_probe = METRIC == METRIC_L2 && p.sq_by_residual;
    if (METRIC != METRIC_L2) {
        if (tid == 0) { // <q, b>: one sequential chain (ivfsq_fused_kernel)
            float acc = 0.f;
            if (CT != SQ_F16)
                for (int i = 0; i < p.d; ++i) acc = __fmaf_rn(xq[i], p.sq_b_plain[i], acc);
            s_qb = acc;
        }
        __syncthreads();
    }
    const float qb = METRIC != METRIC_L2 ? s_qb : 0.f;
    // The table entries (a_j, s_j) of the query -- per probe when the codes are residuals -- are computed ONCE per workgroup into
    // LDS when they fit (nprobe x 16 nch <= 4096 floats: nprobe <= 32 at d = 128).  (Every lane recomputing them from the query,
    // the decoder tables and its probe's centroid cost 16 sixteen-byte loads per lane and chunk, 136 per candidate: the
    // texture path, not HBM, paced the kernel -- 0.18 ms at nb = 1M.)  Same expressions, evaluated once instead of per candidate.
    // Rows of s_a are dp + 4 floats apart: lanes of different probes read the same chunk of different rows at once, and rows a
    // multiple of 128 bytes apart would all start in the same bank.
    __shared__ __attribute__((aligned(16))) float s_a[4096 + 4 * 64];
    __shared__ __attribute__((aligned(16))) float s_s[512];
    const int dp = 16 * nch, lda = dp + 4;
    const int arows = per_probe ? np : 1;
    const bool staged = arows * dp <= 4096 && arows <= 64 && dp <= 512; // (workgroup-uniform)
    if (staged) {
        // one (row, chunk) per thread and round: its loads leave together
        for (int t = tid; t < arows * nch; t += 256) {
            const int pr = t / nch, c = t - pr * nch;
            const int64_t l = per_probe ? p.coarse_ids[(int64_t)q * np + pr] : -1;
            const float* cen = p.centroids + (l >= 0 ? l : 0) * p.ldc;
            float av[16];
            if (16 * c + 16 <= p.d) {
#pragma unroll
                for (int v4 = 0; v4 < 4; ++v4) {
                    const f32x4 x4 = *(const f32x4*)(xq + 16 * c + 4 * v4);
                    f32x4 s4 = f32x4{0.f, 0.f, 0.f, 0.f}, b4 = s4, c4 = s4;
                    if (CT != SQ_F16) {
                        s4 = *(const f32x4*)(p.sq_s + 16 * c + 4 * v4);
                        b4 = *(const f32x4*)(p.sq_b_plain + 16 * c + 4 * v4);
                    }
                    if (per_probe && l >= 0) c4 = *(const f32x4*)(cen + 16 * c + 4 * v4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float a;
                        if (per_probe) {
                            a = x4[e] - c4[e];
                            if (CT != SQ_F16) a = a - b4[e];
                        } else if (METRIC == METRIC_L2) {
                            a = CT == SQ_F16 ? x4[e] : x4[e] - b4[e];
                        } else {
                            a = CT == SQ_F16 ? x4[e] : x4[e] * s4[e];
                        }
                        av[4 * v4 + e] = a;
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int jx = 16 * c + e;
                    float a = 0.f;
                    if (jx < p.d) {
                        if (per_probe) {
                            a = xq[jx] - (l >= 0 ? cen[jx] : 0.f);
                            if (CT != SQ_F16) a = a - p.sq_b_plain[jx];
                        } else if (METRIC == METRIC_L2) {
                            a = CT == SQ_F16 ? xq[jx] : xq[jx] - p.sq_b_plain[jx];
                        } else {
                            a = CT == SQ_F16 ? xq[jx] : xq[jx] * p.sq_s[jx];
                        }
                    }
                    av[e] = a;
                }
            }
#pragma unroll
            for (int v4 = 0; v4 < 4; ++v4)
                *(f32x4*)(s_a + pr * lda + 16 * c + 4 * v4) = f32x4{av[4 * v4], av[4 * v4 + 1], av[4 * v4 + 2], av[4 * v4 + 3]};
        }
        for (int t = tid; t < dp; t += 256) s_s[t] = (CT != SQ_F16 && t < p.d) ? p.sq_s[t] : 0.f;
        __syncthreads();
    }
    for (int i = tid; i < n; i += 256) {
        const uint32_t pos = (uint32_t)kq[i];
        const int pr = (int)cpr[i];
        const int64_t l = p.coarse_ids[(int64_t)q * np + pr];
        const int64_t row = p.row_base[(int64_t)q * np + pr] + (int64_t)pos;
        const uint8_t* rp = p.arena_codes + (row >> 6) * 64 * (int64_t)p.sq_ld + (row & 63) * CHB;
        const float* cen = p.centroids + l * p.ldc;
        f32x2 acc = {0.f, 0.f};
        auto chunk = [&](int c, const unsigned (&w)[W]) __attribute__((always_inline)) {
            // the table entries of this chunk for the candidate's probe: the expressions of ivfsq_fused_kernel's table build.
            // Whole chunks through 16-byte loads (the rows of xq / centroids / sq_s / sq_b are padded to 8 floats and more)
            float sv[16], av[16];
            if (staged) {
                const float* ar = s_a + (per_probe ? pr : 0) * lda + 16 * c;
#pragma unroll
                for (int v4 = 0; v4 < 4; ++v4) {
                    const f32x4 a4 = *(const f32x4*)(ar + 4 * v4), s4 = *(const f32x4*)(s_s + 16 * c + 4 * v4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) av[4 * v4 + e] = a4[e], sv[4 * v4 + e] = s4[e];
                }
            } else if (16 * c + 16 <= p.d) { // (workgroup-uniform)
#pragma unroll
                for (int v4 = 0; v4 < 4; ++v4) {
                    const f32x4 x4 = *(const f32x4*)(xq + 16 * c + 4 * v4);
                    f32x4 s4 = f32x4{0.f, 0.f, 0.f, 0.f}, b4 = s4, c4 = s4;
                    if (CT != SQ_F16) {
                        s4 = *(const f32x4*)(p.sq_s + 16 * c + 4 * v4);
                        b4 = *(const f32x4*)(p.sq_b_plain + 16 * c + 4 * v4);
                    }
                    if (per_probe) c4 = *(const f32x4*)(cen + 16 * c + 4 * v4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float a;
                        if (per_probe) {
                            a = x4[e] - c4[e];
                            if (CT != SQ_F16) a = a - b4[e];
                        } else if (METRIC == METRIC_L2) {
                            a = CT == SQ_F16 ? x4[e] : x4[e] - b4[e];
                        } else {
                            a = CT == SQ_F16 ? x4[e] : x4[e] * s4[e];
                        }
                        av[4 * v4 + e] = a;
                        sv[4 * v4 + e] = s4[e];
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int jx = 16 * c + e;
                    float a = 0.f, sj = 0.f;
                    if (jx < p.d) {
                        if (CT != SQ_F16) sj = p.sq_s[jx];
                        if (per_probe) {
                            a = xq[jx] - cen[jx];
                            if (CT != SQ_F16) a = a - p.sq_b_plain[jx];
                        } else if (METRIC == METRIC_L2) {
                            a = CT == SQ_F16 ? xq[jx] : xq[jx] - p.sq_b_plain[jx];
                        } else {
                            a = CT == SQ_F16 ? xq[jx] : xq[jx] * p.sq_s[jx];
                        }
                    }
                    av[e] = a;
                    sv[e] = sj;
                }
            }
            sq_fold<METRIC, CT, 0>(w, sv, av, acc);
        };
        // d <= 128 with codes of <= 16 bytes per chunk: all of the row's chunks are loaded before the first is folded (the row is
        // a random read from HBM: one round trip per candidate instead of one per chunk -- 0.18 ms of rerank at nb = 1M were eight
        // of them in a row)
        constexpr bool PRE = W <= 4;
        if (PRE && nch <= 8) { // (workgroup-uniform)
            unsigned wall[8][W];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const unsigned* src = (const unsigned*)(rp + (int64_t)min(c, nch - 1) * 64 * CHB);
#pragma unroll
                for (int u = 0; u < W; ++u) wall[c][u] = src[u];
            }
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (c < nch) chunk(c, wall[c]);
        } else {
            for (int c = 0; c < nch; ++c) {
                unsigned w[W];
                const unsigned* src = (const unsigned*)(rp + (int64_t)c * 64 * CHB);
#pragma unroll
                for (int u = 0; u < W; ++u) w[u] = src[u];
                chunk(c, w);
            }
        }
        float dis = acc[0] + acc[1];
        if (METRIC != METRIC_L2) dis = (dis + qb) + (p.sq_by_residual ? p.coarse_dis[(int64_t)q * np + pr] : 0.f);
        const u64 key = ((u64)ordkey<METRIC>(dis) << 32) | (u64)pos;
        if (fin) sel_k[i] = key;
        else kq[i] = key;
    }
    if (fin) lmf_select_tail<256>(p, q, n, sel_k, cpr, sel_wk, sel_wl);
}
void launch_ivf_lmf_rerank_sq(const IvfLmParams& p, hipStream_t stream) {
    if (p.nq == 0) return;
    FA_THROW_IF_NOT(p.kind == 2 && p.arena_codes && p.sq_s && p.sq_b_plain && p.centroids && p.keys && p.cand_pr && p.cnt);
    FA_THROW_IF_NOT(!p.fin_dis || (p.fin_ids && p.arena_ids && p.k <= kLmfFusedSelectK));
    FA_THROW_IF_NOT(p.row_base != nullptr); // (written by launch_ivf_lm_plan)
    const dim3 grid((unsigned)p.nq), block(256);
    const bool l2 = p.metric == METRIC_L2;
#define FA_RRSQ(CT_)                                                                                        \
    do {                                                                                                    \
        if (l2) hipLaunchKernelGGL((lmf_rerank_sq_kernel<METRIC_L2, CT_>), grid, block, 0, stream, p);     \
        else hipLaunchKernelGGL((lmf_rerank_sq_kernel<METRIC_INNER_PRODUCT, CT_>), grid, block, 0, stream, p); \
    } while (0)
    switch (p.sq_ct) {
        case SQ_U8: FA_RRSQ(SQ_U8); break;
        case SQ_U4: FA_RRSQ(SQ_U4); break;
        case SQ_U6: FA_RRSQ(SQ_U6); break;
        default: FA_RRSQ(SQ_F16); break;
    }
#undef FA_RRSQ
    HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------
// Deferred finish (IvfFusedParams::defer_finish): the reservoir a scan workgroup left in part_keys[q][0..n) is cut to
// the k best, translated to user ids and ordered by a small workgroup of its own -- 256 threads, ~11 KB of LDS, many
// per CU -- instead of on the critical path of a 75 KB scan workgroup.  Same code as the in-kernel finish.
// ---------------------------------------------------------------------------------
constexpr int FIN_THREADS = 256;
static size_t ivf_finish_lds_bytes(int kp, int cap, int nprobe) {
    return round_up((size_t)kp * 12, 16) + (size_t)cap * 8 + round_up((size_t)(nprobe + 1) * 4, 16) + (size_t)nprobe * 8 + 1024 +
           64;
}
__global__ void __launch_bounds__(FIN_THREADS) ivf_finish_kernel(IvfFusedParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    FusedLds L{};
    size_t o = 0;
    L.lut = smem; // winners (kp x 12 bytes)
    o += ((size_t)p.kp * 12 + 15) & ~(size_t)15;
    L.res = (u64*)(smem + o);
    o += (size_t)p.cap * 8;
    L.pre = (uint32_t*)(smem + o);
    o += ((size_t)(p.nprobe + 1) * 4 + 15) & ~(size_t)15;
    L.lstart = (int64_t*)(smem + o);
    o += (size_t)p.nprobe * 8;
    L.hist = (unsigned*)(smem + o);
    o += 1024;
    L.ctl = (WgSelCtl*)(smem + o);
    const int q = blockIdx.x, tid = threadIdx.x;
    const int n = (int)p.part_cnt[q];
    const u64* src = p.part_keys + (int64_t)q * p.cap;
    for (int i = tid; i < n; i += FIN_THREADS) L.res[i] = src[i];
    for (int t = tid; t <= p.nprobe; t += FIN_THREADS) L.pre[t] = p.prefix_out[(int64_t)q * (p.nprobe + 1) + t];
    for (int t = tid; t < p.nprobe; t += FIN_THREADS) L.lstart[t] = p.probe_start[(int64_t)q * p.nprobe + t];
    if (tid == 0) L.ctl->cnt = (unsigned)n;
    p.defer_finish = 0;
    fused_finish<FIN_THREADS>(p, q, 0, L); // (starts with a barrier)
}
void launch_ivf_finish(const IvfFusedParams& p, hipStream_t stream) {
    if (p.nq == 0) return;
    FA_THROW_IF_NOT(p.G == 1 && p.part_keys && p.part_cnt && p.prefix_out && p.probe_start);
    const size_t lds = ivf_finish_lds_bytes(p.kp, p.cap, p.nprobe);
    FA_THROW_IF_NOT(lds <= 64 * 1024);
    hipLaunchKernelGGL(ivf_finish_kernel, dim3((unsigned)p.nq), dim3(FIN_THREADS), lds, stream, p);
    HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------
int ivf_fused_threads(int kind) {
    if (kind == 2) return SQ_FB;
    if (kind != 1) return FB_MAX;
    if (const char* e = experiment_env("FAISS_AMD_IVFPQ_FB")) return atoi(e) == 1024 ? 1024 : 512; // timing experiments only
    return 512;
}
bool ivf_fused_supported(int kind, int M, int dpad, int k, int nprobe, int* cap_out, int* kp_out, int* nlut_out) {
    const int fb = ivf_fused_threads(kind);
    int kp = 1;
    while (kp < k) kp <<= 1;
    int cap = 1024;
    while (cap < k + fb) cap <<= 1;
    if (cap > FMAXR * fb) return false;
    if (cap_out) *cap_out = cap;
    if (kp_out) *kp_out = kp;
    const int nlut = 1; // one lookup table per query (see ivfpq_fused_kernel)
    if (nlut_out) *nlut_out = nlut;
    return ivf_fused_lds_bytes(kind, M, dpad, kp, cap, nprobe, nlut) <= 160 * 1024;
}

template <typename K>
static void launch_one(K kern, const IvfFusedParams& p, size_t lds, int fb, hipStream_t stream) {
    HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)(p.nq * p.G)), dim3(fb), lds, stream, p);
}

void launch_ivf_fused(const IvfFusedParams& p, hipStream_t stream) {
    if (p.nq == 0) return;
    FA_THROW_IF_NOT(p.G >= 1 && p.npc >= 1 && p.G * p.npc >= p.nprobe);
    const int fb = ivf_fused_threads(p.kind);
    FA_THROW_IF_NOT(p.cap >= p.k + fb && p.cap <= FMAXR * fb);
    FA_THROW_IF_NOT(p.nlut == 1 || p.nlut == 2);
    const size_t lds = ivf_fused_lds_bytes(p.kind, p.M, p.kind == 2 ? p.sq_dsq : p.dpad, p.kp, p.cap, p.nprobe, p.nlut);
    FA_THROW_IF_NOT_MSG(lds <= 160 * 1024, "fused IVF scan does not fit the LDS");
    const bool l2 = p.metric == METRIC_L2;
    const bool sel = p.sel_mask != nullptr; // IDSelector: the instantiations that test the row mask
    if (p.kind == 0) {
        if (sel) {
            if (l2) launch_one(ivfflat_fused_kernel<METRIC_L2, true>, p, lds, fb, stream);
            else launch_one(ivfflat_fused_kernel<METRIC_INNER_PRODUCT, true>, p, lds, fb, stream);
        } else {
            if (l2) launch_one(ivfflat_fused_kernel<METRIC_L2, false>, p, lds, fb, stream);
            else launch_one(ivfflat_fused_kernel<METRIC_INNER_PRODUCT, false>, p, lds, fb, stream);
        }
    } else if (p.kind == 2) {
        const int nch = p.sq_dsq / 16;
        FA_THROW_IF_NOT_MSG(p.sq_dsq % 16 == 0 && nch >= 1 && nch <= 64, "scalar-quantizer scan: d <= 1024");
        FA_THROW_IF_NOT(p.M == sq_table_rows(p.metric, p.sq_by_residual != 0, p.npc) && p.arena_codes && p.sq_s && p.sq_b);
        FA_THROW_IF_NOT(p.sq_ld % 4 == 0 && p.sq_ld >= nch * sq_chunk_bytes(p.sq_ct));
#define FA_SQ_LAUNCH(CT_)                                                                                  \
    do {                                                                                                   \
        if (sel) {                                                                                         \
            if (l2) launch_one(ivfsq_fused_kernel<METRIC_L2, CT_, true>, p, lds, fb, stream);              \
            else launch_one(ivfsq_fused_kernel<METRIC_INNER_PRODUCT, CT_, true>, p, lds, fb, stream);      \
        } else {                                                                                           \
            if (l2) launch_one(ivfsq_fused_kernel<METRIC_L2, CT_, false>, p, lds, fb, stream);             \
            else launch_one(ivfsq_fused_kernel<METRIC_INNER_PRODUCT, CT_, false>, p, lds, fb, stream);     \
        }                                                                                                  \
    } while (0)
        switch (p.sq_ct) {
            case SQ_U8: FA_SQ_LAUNCH(SQ_U8); break;
            case SQ_U4: FA_SQ_LAUNCH(SQ_U4); break;
            case SQ_U6: FA_SQ_LAUNCH(SQ_U6); break;
            default: FA_SQ_LAUNCH(SQ_F16); break;
        }
#undef FA_SQ_LAUNCH
    } else {
        FA_THROW_IF_NOT_MSG(p.metric != METRIC_L2 || p.arena_t2, "IVFPQ L2 needs the per-vector t2 terms");
#define FA_PQ_LAUNCH(M64_, FB_)                                                                            \
    do {                                                                                                   \
        if (sel) {                                                                                         \
            if (l2) launch_one(ivfpq_fused_kernel<METRIC_L2, M64_, FB_, true>, p, lds, fb, stream);        \
            else launch_one(ivfpq_fused_kernel<METRIC_INNER_PRODUCT, M64_, FB_, true>, p, lds, fb, stream); \
        } else {                                                                                           \
            if (l2) launch_one(ivfpq_fused_kernel<METRIC_L2, M64_, FB_, false>, p, lds, fb, stream);       \
            else launch_one(ivfpq_fused_kernel<METRIC_INNER_PRODUCT, M64_, FB_, false>, p, lds, fb, stream); \
        }                                                                                                  \
    } while (0)
        if (p.M == 64) {
            if (fb == 512) FA_PQ_LAUNCH(true, 512);
            else FA_PQ_LAUNCH(true, 1024);
        } else {
            if (fb == 512) FA_PQ_LAUNCH(false, 512);
            else FA_PQ_LAUNCH(false, 1024);
        }
#undef FA_PQ_LAUNCH
    }
    HIP_CHECK(hipGetLastError());
}

} // namespace faiss_amd
