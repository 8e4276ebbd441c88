// faiss_amd/csrc/ivf_listmajor.hip -- list-major inverted-list search for gfx950 (round 3).
//
// The query-major scans (ivf_fused.hip; the reference's IVFInterleaved.cuh:33-224 and
// PQScanMultiPassNoPrecomputed-inl.cuh:173-270 are query-major too) stream every probed list once per query: a batch
// of 10 000 queries x 32 probes over 4096 lists reads each list ~78 times.  Here a workgroup takes one list and up to
// 128 of the queries that probe it: the queries sit in registers as MFMA B operands, the list's rows pass through LDS
// in 64-row tiles ONCE (IVFPQ: decoded from the codes on the way in), and the 64 x 128 distance block is computed on
// the f32 matrix pipe (v_mfma_f32_32x32x2_f32 = a k-ordered fmaf chain, bit for bit that of flat_scan_kernel).
// The bound moves from the code / vector stream (HBM, fabric) to the f32 MFMA rate: 2 * nq * nprobe * (nb / nlist) * d
// flop per batch (20 GFLOP at nb = 1M, 2 TFLOP at nb = 100M) against 157 TFLOP/s.
//
// Per-query state cannot live in a workgroup any more (a query's probes are spread over many of them), so selection is
// two passes over a per-query key segment in HBM (kernels.h IvfLmParams): pass 1 writes every distance of the query's
// leading probes (>= k rows) at its scan position; the k-th smallest of those is a bound no better result can exceed;
// pass 2 appends the rows of the remaining probes that are at or below the bound.  The result is the k smallest keys
// of the segment under (distance, scan position) -- the same rule as every other scan here -- whatever order the
// workgroups ran in.
#include "kernels.h"
#include <type_traits>
#include "wg_select.h"

namespace faiss_amd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned long long u64;

constexpr int LM_THREADS = 256;
constexpr int LM_TR = 64;                       // rows per tile
constexpr int LM_ROWB = 512;                    // LDS bytes per tile row (128 floats, whatever dpad is)
constexpr int LM_TILE_BYTES = LM_TR * LM_ROWB;  // 32768
// IVFPQ (tile decoded by the threads): one tile + 64 row norms.  IVFFlat: TWO tiles + 2 x 64 row norms -- the rows of
// tile t + 1 arrive by LDS-DMA (global_load_lds_dwordx4: no staging registers) while tile t is multiplied.
constexpr int LM_LDS_RN = LM_TILE_BYTES;        // 64 row norms (single-tile layout)
constexpr int LM2_LDS_RN = 2 * LM_TILE_BYTES;   // 2 x 64 row norms (two-tile layout)
// pass 2: every wave parks its candidates (key + query) in an LDS slice of its own and hands them to the segments in
// HBM only when the slice fills up (and when the kernel ends): one atomic round trip per ~hundred candidates instead of
// one per 32-row block -- a returning atomic inside the tile loop stalls its wave for a memory round trip per tile,
// as long as the tile's MFMAs take, and drains the prefetch with it
constexpr int LM_PARK = 192;                                   // parked candidates per wave
constexpr int LM_PARK_BYTES = 4 * LM_PARK * (8 + 4);           // 4 waves x (u64 key + u32 query)
constexpr int LM_LDS_PARK = LM_TILE_BYTES + LM_TR * 4;         // single-tile layout
constexpr int LM2_LDS_PARK = 2 * LM_TILE_BYTES + 2 * LM_TR * 4; // two-tile layout
constexpr int LM_LDS_TOTAL_P = LM_LDS_PARK + LM_PARK_BYTES;
constexpr int LM2_LDS_TOTAL = LM2_LDS_PARK + LM_PARK_BYTES;

// LDS-DMA issued from inline asm (the helpers of flat_filter.hip): hipcc would make every ds_read that follows a
// __builtin_amdgcn_global_load_lds wait for vmcnt(0), draining the prefetch before the tile in hand is even read.
// Hidden in asm the DMA is invisible to the compiler's counters; completion is enforced by our own counted
// s_waitcnt vmcnt(N) + barrier before a tile is consumed (cdna_hip_programming.md 5.7).  Extra vector-memory operations
// of the compiler's (the epilogue's key stores, the query loads of the next item) only make both sides' counted waits
// conservative: loads complete in order, so "at most N outstanding" always covers everything older than the last N.
__device__ __forceinline__ void lm_glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
            "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(gsrc), "s"(lds_dst)
            : "memory");
}
__device__ __forceinline__ void lm_glds4(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
            "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(gsrc), "s"(lds_dst)
            : "memory");
}
// Workgroup barrier WITHOUT the fences of __syncthreads(): with key stores / atomics of the epilogue pending, the
// workgroup-scope release / acquire around s_barrier makes hipcc wait for vmcnt(0), which also drains the prefetch DMAs.
// In the DMA pipeline no thread writes LDS (the tile arrives by DMA, covered by the counted vmcnt wait before this
// barrier) and a wave that arrives has consumed every LDS read it issued; the "memory" clobber keeps the compiler from
// moving LDS reads across it.
__device__ __forceinline__ void lm_barrier() {
    asm volatile("s_barrier" ::: "memory");
}
// same, scalar 64-bit base + 32-bit per-lane byte offset: the per-tile address arithmetic is one scalar add.  The leading
// s_nop covers the SALU-write -> VMEM-read hazard on the base SGPRs, which hipcc cannot see inside an asm statement
// (cdna_hip_programming.md 5.7)
__device__ __forceinline__ void lm_glds16_s(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
            "s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(voff), "s"(sbase), "s"(lds_dst)
            : "memory");
}
__device__ __forceinline__ void lm_glds4_s(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
            "s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(voff), "s"(sbase), "s"(lds_dst)
            : "memory");
}
__device__ __forceinline__ const char* lm_uniform_ptr(const char* ptr) {
    const unsigned long long v = (unsigned long long)ptr;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const char*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ unsigned lm_lds_addr(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}

bool ivf_lm_supported(int kind, int dpad, int M, int d) {
    if (dpad > 128 || (dpad & 7)) return false;
    if (kind == 0) return true;
    if (kind == 1) return M >= 1 && d % M == 0;
    // scalar quantizer: every code type (6-bit fields: the 24 bits of an operand group straddle dwords and are cut out of
    // a dword pair with one v_alignbit)
    if (kind == 2) return M == SQ_U8 || M == SQ_U4 || M == SQ_U6 || M == SQ_F16;
    return false;
}

// ------------------------------------------------------------------ IVF scalar quantizer: decode helpers
// bytes of one 16-component chunk
template <int CT>
struct LmSq {
    static constexpr int CHB = CT == SQ_U8 ? 16 : CT == SQ_U4 ? 8 : CT == SQ_U6 ? 12 : 32;
    // bytes of the 4 components 8 s + 4 h + e of a lane's MFMA operand group (6-bit codes: 3 bytes, not addressable as
    // such -- see lm_sq_piece_off / lm_sq_piece_shift)
    static constexpr int PIECE = CHB / 4;
    static constexpr float MID = CT == SQ_U8 ? 127.5f : CT == SQ_U4 ? 7.5f : CT == SQ_U6 ? 31.5f : 0.f;
};
// piece g (0..3) of a row's chunk = its components 4 g .. 4 g + 3.  Byte offset of the load that fetches it and, for 6-bit
// codes, the bit position of the piece inside the loaded dword pair: pieces start at bytes 0, 3, 6, 9 of the 12-byte
// chunk, i.e. in dwords 0, 0, 1, 2 at bits 0, 24, 16, 8.  (The pair of piece 3 ends one dword behind the chunk: the next
// row's bytes, the next chunk of the block or the arena's padding -- loaded, shifted out.)
template <int CT>
__device__ __forceinline__ int lm_sq_piece_off(int g) {
    if constexpr (CT == SQ_U6) return g < 2 ? 0 : 4 * (g - 1);
    else return g * LmSq<CT>::PIECE;
}
__device__ __forceinline__ int lm_sq_piece_shift(int g) { // 6-bit codes only
    return (32 - 8 * g) & 31;
}
// component e (0..3) of a 4-component piece (8-bit: a dword; 4-bit: 16 bits; 6-bit: 24 bits at bit `sh` of a dword pair;
// fp16: two dwords) as the matrix pipe sees it: integer codes CENTRED on the middle of their range (code - 127.5 / 7.5 /
// 31.5, exact in fp32; the offset b the query operand carries is moved by the same amount, kernels.h
// IvfLmParams::sq_b).  Uncentred, |a|^2 and |s o code|^2 are ~10 x the distance they cancel to (both vectors sit half a
// range away from the origin) and the rounding of the three terms shows at 4e-5 of the distances at the bench shape;
// centred it is the rounding of the distance itself.
template <int CT>
__device__ __forceinline__ float lm_sq_comp(const uint2 w, int e, int sh = 0) {
    if constexpr (CT == SQ_U8) {
        return __fsub_rn((float)((w.x >> (8 * e)) & 255u), 127.5f);
    } else if constexpr (CT == SQ_U4) {
        return __fsub_rn((float)((w.x >> (4 * e)) & 15u), 7.5f);
    } else if constexpr (CT == SQ_U6) {
        const unsigned v = __builtin_amdgcn_alignbit(w.y, w.x, (unsigned)sh); // ({w.y, w.x} >> sh)[31:0]
        return __fsub_rn((float)((v >> (6 * e)) & 63u), 31.5f);
    } else {
        const unsigned v = e < 2 ? w.x : w.y;
        return (float)__builtin_bit_cast(_Float16, (unsigned short)(v >> (16 * (e & 1))));
    }
}
// the load of a piece: `ptr` = the row's chunk bytes + lm_sq_piece_off(g)
struct __attribute__((packed, aligned(4))) LmDwordPair {
    unsigned x, y;
};
template <int CT>
__device__ __forceinline__ uint2 lm_sq_load_piece(const uint8_t* ptr) {
    if constexpr (CT == SQ_U8) {
        return uint2{*(const unsigned*)ptr, 0u};
    } else if constexpr (CT == SQ_U4) {
        return uint2{(unsigned)*(const unsigned short*)ptr, 0u};
    } else if constexpr (CT == SQ_U6) {
        const LmDwordPair v = *(const LmDwordPair*)ptr; // (dword-aligned, not 8-byte-aligned)
        return uint2{v.x, v.y};
    } else {
        return *(const uint2*)ptr;
    }
}
// The B operands of a (query, list) pair and the scalar that goes with them (kernels.h IvfLmParams, kind 2).  Lane (h, j)
// owns the coordinates 8 s + 4 h + e of query j.  L2: bq = a o s with a = (q [- centroid]) - b, xn = |a|^2; inner product:
// bq = q o s, xn = <q, b> + coarse.  |a|^2 and <q, b>: each lane of the pair (h = 0 / 1) runs a sequential fmaf chain over
// its own coordinates, the two sums are added (oracle: orc_ivfsq_search, arith 1).
template <int METRIC, bool FULL>
__device__ __forceinline__ void lm_sq_query(const IvfLmParams& p, int ns, int h, const float* qrow, const float* cen,
                                            float coarse, f32x4 (&bq)[16], float& xn) {
    // branch-free: `cen` is the list's centroid or a row of zeros (no residual encoding: x - 0 == x), fp16 codes come
    // with s = 1, b = 0 (a * 1 == a, fmaf(q, 0, acc) == acc)
    float acc = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        // (four coordinate groups at a time: left alone hipcc hoists all 64 loads -- query, centroid, scale, offset --
        // above the arithmetic and spills)
        if ((s & 3) == 0 && s > 0) __builtin_amdgcn_sched_barrier(0);
        if (FULL || s < ns) {
            const f32x4 v = *(const f32x4*)(qrow + 8 * s + 4 * h);
            const f32x4 s4 = *(const f32x4*)(p.sq_s + 8 * s + 4 * h);
            const f32x4 b4 = *(const f32x4*)(p.sq_b + 8 * s + 4 * h);
            if (METRIC == METRIC_L2) {
                const f32x4 c4 = *(const f32x4*)(cen + 8 * s + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a = __fsub_rn(__fsub_rn(v[e], c4[e]), b4[e]);
                    acc = __fmaf_rn(a, a, acc);
                    bq[s][e] = __fmul_rn(a, s4[e]);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc = __fmaf_rn(v[e], b4[e], acc);
                    bq[s][e] = __fmul_rn(v[e], s4[e]);
                }
            }
        } else {
            bq[s] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const float both = acc + __shfl_xor(acc, 32, 64);
    xn = METRIC == METRIC_L2 ? both : both + coarse;
}

// |s o code|^2 of arena rows (the list-major scan's second L2 term for the scalar quantizer)
template <int CT>
__global__ void ivfsq_row_norms_kernel(const uint8_t* __restrict__ arena, int ld, int d, const float* __restrict__ sq_s,
                                       const int64_t* __restrict__ dest, int64_t row0, int64_t n, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t row = dest ? dest[i] : row0 + i;
    if (row < 0) return;
    constexpr int CHB = LmSq<CT>::CHB;
    const uint8_t* rp = arena + (row >> 6) * 64 * (int64_t)ld + (row & 63) * CHB;
    float acc = 0.f;
    for (int c = 0; 16 * c < d; ++c) {
        const uint8_t* ch = rp + (size_t)c * 64 * CHB;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint2 w = lm_sq_load_piece<CT>(ch + lm_sq_piece_off<CT>(g));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = 16 * c + 4 * g + e;
                if (j < d) {
                    const float cf = lm_sq_comp<CT>(w, e, lm_sq_piece_shift(g));
                    const float v = CT == SQ_F16 ? cf : __fmul_rn(sq_s[j], cf);
                    acc = __fmaf_rn(v, v, acc);
                }
            }
        }
    }
    out[row] = acc;
}
void launch_ivfsq_row_norms(const uint8_t* arena, int ct, int ld, int d, const float* sq_s, const int64_t* dest, int64_t row0,
                            int64_t n, float* out, hipStream_t stream) {
    if (n == 0) return;
    const dim3 grid((unsigned)div_up(n, 256)), block(256);
    switch (ct) {
        case SQ_U8: hipLaunchKernelGGL(ivfsq_row_norms_kernel<SQ_U8>, grid, block, 0, stream, arena, ld, d, sq_s, dest, row0, n, out); break;
        case SQ_U4: hipLaunchKernelGGL(ivfsq_row_norms_kernel<SQ_U4>, grid, block, 0, stream, arena, ld, d, sq_s, dest, row0, n, out); break;
        case SQ_U6: hipLaunchKernelGGL(ivfsq_row_norms_kernel<SQ_U6>, grid, block, 0, stream, arena, ld, d, sq_s, dest, row0, n, out); break;
        case SQ_F16: hipLaunchKernelGGL(ivfsq_row_norms_kernel<SQ_F16>, grid, block, 0, stream, arena, ld, d, sq_s, dest, row0, n, out); break;
        default: FA_THROW_MSG("list-major scan: scalar-quantizer code type without row norms");
    }
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------ plan
// one wavefront per query: scan positions of its probes (exclusive prefix sums over the probe list, 64 probes per round),
// the probes of pass 1, pairs per (pass, list)
__global__ void __launch_bounds__(256) lm_plan_kernel(IvfLmParams p) {
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (q >= p.nq) return;
    const int np = p.nprobe;
    const int64_t* ids = p.coarse_ids + (int64_t)q * np;
    uint32_t* pre = p.prefix + (int64_t)q * (np + 1);
    uint32_t* pre1 = p.prefix1 + (int64_t)q * (np + 1);
    // pass 1 sees the first rows_per_item rows of a list (its first row chunk); the rest of a longer list waits for
    // pass 2 like the lists of the later probes
    const uint32_t r1max = p.force_all ? 0xffffffffu : (uint32_t)p.rows_per_item;
    uint32_t cum = 0, cum1 = 0; // (wave-uniform) totals of the probes before this round
    // f16 filter scan (ivf_lm_filter.hip): granule slots of the probes, 2 per 32 * gran_blocks rows of a list
    uint32_t* preg = p.filter ? p.prefixg + (int64_t)q * (np + 1) : nullptr;
    const uint32_t grows = 32u * (uint32_t)max(p.gran_blocks, 1);
    uint32_t cumg = 0;
    int p0 = np;
    for (int base = 0; base < np; base += 64) {
        const int pr = base + lane;
        const int64_t l = pr < np ? ids[pr] : -1;
        const uint32_t len = l >= 0 ? p.list_len[l] : 0u;
        const uint32_t len1 = min(len, r1max);
        uint32_t inc = len, inc1 = len1; // inclusive scans over the lanes
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(inc, off, 64), o1 = __shfl_up(inc1, off, 64);
            if (lane >= off) {
                inc += o;
                inc1 += o1;
            }
        }
        if (pr < np) {
            pre[pr] = cum + inc - len;
            pre1[pr] = cum1 + inc1 - len1;
            if (p.row_base) p.row_base[(int64_t)q * np + pr] = l >= 0 ? p.list_start[l] - (int64_t)(cum + inc - len) : 0;
        }
        if (p.filter) {
            const uint32_t lg = 2u * ivf_lmf_list_granules(len, (uint32_t)p.rows_per_item, grows, (uint32_t)p.sample_rows);
            uint32_t incg = lg;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t o = __shfl_up(incg, off, 64);
                if (lane >= off) incg += o;
            }
            if (pr < np) preg[pr] = cumg + incg - lg;
            cumg += __shfl(incg, 63, 64);
        }
        // first probe after which pass 1 has seen k rows
        const unsigned long long reach = __ballot(pr < np && cum1 + inc1 >= (uint32_t)p.k);
        if (p0 == np && reach) p0 = base + __ffsll((long long)reach);
        cum += __shfl(inc, 63, 64);
        cum1 += __shfl(inc1, 63, 64);
    }
    // at least min_p1 probes in pass 1: the k-th best of the nearest FEW lists is a far tighter bound than that of the
    // nearest one alone (a query near a cell border finds most of its neighbours next door)
    if (p0 < p.min_p1) p0 = min(np, p.min_p1);
    if (p.force_all) p0 = np;
    if (p.filter) p0 = 0; // no exact sample: every pair belongs to the sweeps' one class of items
    if (lane == 0) {
        pre[np] = cum;
        pre1[np] = cum1;
        p.p0[q] = (uint32_t)p0;
        if (p.filter) preg[np] = cumg;
    }
    // rows of pass 1 = pre1[p0] (p0 == np: the total)
    uint32_t c1 = cum1;
    if (p0 < np) {
        // recompute from the lens of the probes before p0 (a second sweep is cheaper than a round trip through memory)
        uint32_t s1 = 0;
        for (int base = 0; base < p0; base += 64) {
            const int pr = base + lane;
            const int64_t l = pr < p0 ? ids[pr] : -1;
            uint32_t v = l >= 0 ? min(p.list_len[l], r1max) : 0u;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
            s1 += v;
        }
        c1 = s1;
    }
    if (lane == 0) p.cnt[q] = c1;
    for (int pr = lane; pr < np; pr += 64) {
        const int64_t l = ids[pr];
        if (l >= 0 && p.list_len[l] > 0) atomicAdd(&p.bucket_cnt[2 * (int)l + (pr >= p0 ? 1 : 0)], 1u);
    }
}

// one workgroup: bucket_start = exclusive scan of bucket_cnt; work items.  Bucket 2 l holds the pass-1 pairs of list l,
// bucket 2 l + 1 its other pairs (next to each other in `pairs`).  Pass 1 runs (pass-1 bucket, row chunk 0); pass 2 runs row
// chunk 0 for the other bucket, and the row chunks 1.. of a list (lists longer than a chunk) for BOTH buckets as one range
// of pairs (item.both): past the first chunk all queries of a list are candidates against the bound alike, and
// grouping them together fills the 32-query blocks (10 + 68 queries: 3 blocks instead of 1 + 3).  Items are listed pass 1
// first, list by list, row chunk by row chunk, query group innermost (consecutive items share their rows).
__global__ void __launch_bounds__(1024) lm_items_kernel(IvfLmParams p) {
    __shared__ uint32_t part_pairs[1024];
    __shared__ uint32_t part_i1[1024];
    __shared__ uint32_t part_i2[1024];
    __shared__ uint32_t tot[2];
    const int t = threadIdx.x;
    const int n = p.nlist;
    const int per = (n + 1023) / 1024;
    const int a = min(n, t * per), b = min(n, a + per);
    auto shape = [&](int l, uint32_t& c1, uint32_t& c2, int& nrt) {
        c1 = p.bucket_cnt[2 * l];
        c2 = p.bucket_cnt[2 * l + 1];
        nrt = p.force_all ? 1 : (int)((p.list_len[l] + p.rows_per_item - 1) / p.rows_per_item);
    };
    auto groups = [&](uint32_t c) { return (c + p.qpi - 1) / p.qpi; };
    // (force_all: one item takes all rows of its list: the redo of a few queries is not worth balancing)
    uint32_t sp = 0, s1 = 0, s2 = 0;
    for (int i = a; i < b; ++i) {
        uint32_t c1, c2;
        int nrt;
        shape(i, c1, c2, nrt);
        sp += c1 + c2;
        s1 += groups(c1);
        s2 += groups(c2) + groups(c1 + c2) * (uint32_t)max(nrt - 1, 0);
    }
    // exclusive scans of the three per-thread sums over the 1024 threads: inclusive scan inside every wavefront
    // (shuffles), then the 16 wavefront totals by the first wavefront
    {
        const int ln = t & 63, wv = t >> 6;
        uint32_t ip = sp, i1s = s1, i2s = s2;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t op = __shfl_up(ip, off, 64), o1 = __shfl_up(i1s, off, 64), o2 = __shfl_up(i2s, off, 64);
            if (ln >= off) {
                ip += op;
                i1s += o1;
                i2s += o2;
            }
        }
        if (ln == 63) {
            part_pairs[wv] = ip;
            part_i1[wv] = i1s;
            part_i2[wv] = i2s;
        }
        __syncthreads();
        if (t < 64) {
            uint32_t vp = t < 16 ? part_pairs[t] : 0u, v1 = t < 16 ? part_i1[t] : 0u, v2 = t < 16 ? part_i2[t] : 0u;
            const uint32_t wp = vp, w1 = v1, w2 = v2;
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
                const uint32_t op = __shfl_up(vp, off, 64), o1 = __shfl_up(v1, off, 64), o2 = __shfl_up(v2, off, 64);
                if (t >= off) {
                    vp += op;
                    v1 += o1;
                    v2 += o2;
                }
            }
            if (t < 16) { // exclusive offsets of the wavefronts (written behind the totals read above)
                part_pairs[64 + t] = vp - wp;
                part_i1[64 + t] = v1 - w1;
                part_i2[64 + t] = v2 - w2;
            }
            if (t == 15) {
                p.bucket_start[2 * n] = vp;
                tot[0] = v1;
                tot[1] = v2;
            }
        }
        __syncthreads();
        sp = part_pairs[64 + wv] + ip - sp; // exclusive prefix of this thread
        s1 = part_i1[64 + wv] + i1s - s1;
        s2 = part_i2[64 + wv] + i2s - s2;
    }
    uint32_t rp = sp, i1 = s1, i2 = tot[0] + s2;
    auto put = [&](uint32_t at, int bk, int qt, int rt, int both) {
        if (at < (uint32_t)p.max_items) p.items[at] = IvfLmItem{bk, qt, rt, both};
    };
    for (int i = a; i < b; ++i) {
        uint32_t c1, c2;
        int nrt;
        shape(i, c1, c2, nrt);
        p.bucket_start[2 * i] = rp;
        p.bucket_start[2 * i + 1] = rp + c1;
        for (uint32_t qt = 0; qt < groups(c1); ++qt) put(i1++, 2 * i, (int)qt, 0, 0);
        for (uint32_t qt = 0; qt < groups(c2); ++qt) put(i2++, 2 * i + 1, (int)qt, 0, 0);
        for (int rt = 1; rt < nrt; ++rt)
            for (uint32_t qt = 0; qt < groups(c1 + c2); ++qt) put(i2++, 2 * i, (int)qt, rt, 1);
        rp += c1 + c2;
    }
    if (t == 0) {
        const uint32_t all = tot[0] + tot[1];
        p.item_bounds[0] = 0;
        p.item_bounds[1] = min(tot[0], (uint32_t)p.max_items);
        p.item_bounds[2] = min(all, (uint32_t)p.max_items);
        // (max_items is an upper bound computed on the host from the list lengths; exceeding it would lose work, so the
        // host checks item_bounds[3] == 0 whenever it reads the overflow word)
        p.item_bounds[3] = all > (uint32_t)p.max_items ? 1u : 0u;
        p.item_bounds[4] = 0; // item counter of pass 2 (the kernels whose waves draw their items: IVFFlat pass 2, IVFPQ)
        p.item_bounds[5] = 0; // ... of pass 1 (IVFPQ)
    }
    // the filter sweeps draw their items from one counter per XCD (ivf_lm_filter.hip)
    if (p.filter && t < 16) p.item_bounds[kLmXcdCtr + t * 32] = 0;
}

// one thread per (query, probe): its place among the pairs of its bucket (any order inside a bucket)
__global__ void lm_fill_kernel(IvfLmParams p) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)p.nq * p.nprobe) return;
    const int64_t l = p.coarse_ids[i];
    if (l < 0 || p.list_len[l] == 0) return;
    const int q = (int)(i / p.nprobe), pr = (int)(i - (int64_t)q * p.nprobe);
    const int bk = 2 * (int)l + (pr >= (int)p.p0[q] ? 1 : 0);
    const uint32_t slot = p.bucket_start[bk] + atomicAdd(&p.bucket_fill[bk], 1u);
    p.pairs[slot] = (uint32_t)i;
}

void launch_ivf_lm_plan(const IvfLmParams& p, hipStream_t stream) {
    if (p.nq == 0) return;
    FA_THROW_IF_NOT(p.rows_per_item % LM_TR == 0 && p.rows_per_item > 0);
    FA_THROW_IF_NOT(!p.filter || (p.prefixg && p.gran_blocks >= 1 && p.rows_per_item % (32 * p.gran_blocks) == 0 && !p.force_all));
    FA_THROW_IF_NOT(!p.filter || (p.sample_rows >= 0 && p.sample_rows % (32 * p.gran_blocks) == 0));
    // bucket_cnt and bucket_fill are one allocation [2][2 nlist]
    FA_THROW_IF_NOT(p.bucket_fill == p.bucket_cnt + 2 * p.nlist);
    if (!p.pre_cleared) HIP_CHECK(hipMemsetAsync(p.bucket_cnt, 0, (size_t)4 * p.nlist * 4, stream));
    hipLaunchKernelGGL(lm_plan_kernel, dim3((unsigned)div_up(p.nq, 4)), dim3(256), 0, stream, p);
    hipLaunchKernelGGL(lm_items_kernel, dim3(1), dim3(1024), 0, stream, p);
    hipLaunchKernelGGL(lm_fill_kernel, dim3((unsigned)div_up((size_t)p.nq * p.nprobe, 256)), dim3(256), 0, stream, p);
    HIP_CHECK(hipGetLastError());
}

__global__ void lm_clamp_kernel(IvfLmParams p) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= p.nq) return;
    if ((int64_t)p.cnt[q] > p.stride) {
        p.cnt[q] = (uint32_t)p.stride;
        const uint32_t s = atomicAdd(&p.ovf[0], 1u);
        p.ovf[1 + s] = (uint32_t)q;
    }
}
void launch_ivf_lm_clamp(const IvfLmParams& p, hipStream_t stream) {
    if (p.nq == 0) return;
    hipLaunchKernelGGL(lm_clamp_kernel, dim3((unsigned)div_up(p.nq, 256)), dim3(256), 0, stream, p);
    HIP_CHECK(hipGetLastError());
}

__global__ void l2_norms_scatter_kernel(const float* __restrict__ x, int64_t ld, int64_t n, int d,
                                        const int64_t* __restrict__ dest, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t row = dest[i];
    if (row < 0) return;
    const float* r = x + i * ld;
    float acc = 0.f;
    for (int k = 0; k < d; ++k) {
        const float v = r[k];
        acc = __fmaf_rn(v, v, acc);
    }
    out[row] = acc;
}
void launch_l2_norms_scatter(const float* x, int64_t ld, int64_t n, int d, const int64_t* dest, float* out,
                             hipStream_t stream) {
    if (n == 0) return;
    hipLaunchKernelGGL(l2_norms_scatter_kernel, dim3((unsigned)div_up(n, 256)), dim3(256), 0, stream, x, ld, n, d, dest, out);
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------ scan
// Workgroup = 4 waves over one 64-row tile and up to 64 of the item's queries: wave w owns the 32 queries of query block
// w & 1 (B operands: their dpad coordinates in 16 x 4 VGPRs, lane (h, j): query j, coordinates 8 s + 4 h + e) and the 32-row
// block w >> 1 of every tile -- so a list probed by few queries (pass 1) still keeps two waves busy, a full item four.
// (Measured against 128-query items with one query block per wave, and against roles chosen per item by its query
// count: both slower -- profiles/r03_b_listmajor_experiments.txt.)
// MFMA D[i][j]: i = row of the block, j = query: lane (h, j) holds the distances of rows 8 g + 4 h + e (acc[4 g + e]) to
// ITS query -- threshold, scan position and segment are per-lane registers, nothing crosses lanes.
// FULL: dpad == 128 (no bound checks in the k loop).
template <int METRIC, int KIND, int PASS, bool FULL>
__global__ void __launch_bounds__(LM_THREADS, KIND == 1 ? 3 : 2) ivf_lm_scan_kernel(IvfLmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const int j = lane & 31;
    const int np = p.nprobe;
    const int ns = FULL ? 16 : (p.dpad >> 3);
    float* rnl = (float*)(smem + (KIND == 0 ? LM2_LDS_RN : LM_LDS_RN));
    const unsigned lds_base = __builtin_amdgcn_readfirstlane(lm_lds_addr(smem));
    // IVFFlat: DMAs per wave and tile -- 8 x 1 KB of rows (+ the 64 row norms, by every wave, so that all count alike)
    constexpr int NDMA = 8 + (METRIC == METRIC_L2 ? 1 : 0);

    // pass 2: this wave's slice of parked candidates
    u64* pk_keys = (u64*)(smem + (KIND == 0 ? LM2_LDS_PARK : LM_LDS_PARK)) + wave * LM_PARK;
    uint32_t* pk_q = (uint32_t*)(smem + (KIND == 0 ? LM2_LDS_PARK : LM_LDS_PARK) + 4 * LM_PARK * 8) + wave * LM_PARK;
    int wcnt = 0; // (wave-uniform) parked candidates
    auto flush = [&]() __attribute__((always_inline)) {
        for (int e = lane; e < wcnt; e += 64) {
            const u64 key = pk_keys[e];
            const uint32_t qq = pk_q[e];
            uint32_t slot;
            uint32_t* cp = p.cnt + qq;
            const uint32_t one = 1u;
            // (asm, with its own wait: a returning atomic the compiler can see, waited for only inside branches, makes
            // it guard the top of the tile loop with s_waitcnt vmcnt(0) -- draining the prefetch DMAs on EVERY tile)
            asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(slot) : "v"(cp), "v"(one) : "memory");
            if ((int64_t)slot < p.stride) p.keys[(int64_t)qq * p.stride + slot] = key;
        }
        wcnt = 0;
    };

    // IVFFlat: byte offsets of this lane's 8 DMA pieces relative to the first row of a tile
    unsigned voff_row[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = wave * 16 + 2 * i + (lane >> 5);
        int c = (lane & 31) ^ (row & 15);
        if (!FULL && c * 4 >= p.dpad) c = 0; // (columns the k loop never reads)
        voff_row[i] = (unsigned)(row * (int)p.ldv * 4 + c * 16);
    }

    const uint32_t it0 = p.item_bounds[PASS - 1], it1 = p.item_bounds[PASS];
    // blocks b, b + 8, ... share an XCD (and its L2): give them consecutive items
    const int nblk = (int)gridDim.x;
    int vb = (int)blockIdx.x;
    if ((nblk & 7) == 0) vb = ((int)blockIdx.x & 7) * (nblk >> 3) + ((int)blockIdx.x >> 3);

    for (uint32_t it = it0 + (uint32_t)vb; it < it1; it += (uint32_t)nblk) {
        const IvfLmItem item = p.items[it];
        const int bk = __builtin_amdgcn_readfirstlane(item.bucket);
        const int qt = __builtin_amdgcn_readfirstlane(item.qt);
        const int rt = __builtin_amdgcn_readfirstlane(item.rt);
        const int list = bk >> 1;
        const int len = (int)p.list_len[list];
        const int64_t start = p.list_start[list];
        const uint32_t pb = p.bucket_start[bk];
        const int npair = min(p.qpi, (int)(p.bucket_start[bk + 1 + item.both] - pb) - qt * p.qpi);
        const int r0 = rt * p.rows_per_item;
        const int r1 = p.force_all ? len : min(len, r0 + p.rows_per_item);

        // IVFFlat: the 8 DMAs of this wave cover rows 16 w .. 16 w + 15 of a tile; lane l of DMA i lands at chunk position
        // l & 31 of row 16 w + 2 i + (l >> 5) and therefore fetches source chunk (l & 31) ^ (row & 15) (the swizzle of the
        // LDS image, applied through the source address).  Addressing: a wave-uniform 64-bit base per tile (SGPRs) plus
        // per-lane 32-bit offsets that never change (voff_*, set up once per kernel).  Tiles are fetched whole: the rows
        // behind the end of a list belong to the next list or to the arena's padding (index.cpp ensure_arena_) and their
        // distances are never looked at.
        auto issue_tile = [&](int t, int buf_) __attribute__((always_inline)) {
            if (p.dbg & 4) return;
            const int bf = __builtin_amdgcn_readfirstlane(buf_);
            const char* sb = lm_uniform_ptr((const char*)(p.arena_vecs + (start + t) * p.ldv));
#pragma unroll
            for (int i = 0; i < 8; ++i)
                lm_glds16_s(sb, voff_row[i], lds_base + (unsigned)(bf * LM_TILE_BYTES + (wave * 8 + i) * 1024));
            if (METRIC == METRIC_L2)
                lm_glds4_s(lm_uniform_ptr((const char*)(p.arena_rn + start + t)), (unsigned)lane * 4u,
                           lds_base + (unsigned)(LM2_LDS_RN + bf * LM_TR * 4));
        };
        // (first tile on its way before the query operands are gathered)
        if (KIND == 0) issue_tile(r0, 0);

        // ---- this lane's query
        const int wq = wave & 1, wr = wave >> 1; // query block, row block of this wave
        const int my = wq * 32 + j;
        const bool qv = my < npair;
        const uint32_t pi = p.pairs[pb + (uint32_t)(qt * p.qpi) + (uint32_t)(qv ? my : 0)];
        const int q = (int)(pi / (uint32_t)np);
        const int pr = (int)(pi - (uint32_t)q * (uint32_t)np);
        const bool wave_active = wq * 32 < npair; // wave-uniform
        const float* qrow = p.xq + (int64_t)q * p.ldq;
        f32x4 bq[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            if (FULL || s < ns) bq[s] = *(const f32x4*)(qrow + 8 * s + 4 * h);
            else bq[s] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        float xn = 0.f;
        if (KIND == 2) {
            // scalar quantizer: B = a o s, xn = |a|^2 (L2) or <q, b> + coarse (inner product); see lm_sq_query
            const float* cen = (METRIC == METRIC_L2 && p.sq_by_residual) ? p.centroids + (int64_t)list * p.ldc : p.sq_zero;
            const float coarse = (METRIC != METRIC_L2 && p.sq_by_residual) ? p.coarse_dis[pi] : 0.f;
            lm_sq_query<METRIC, FULL>(p, ns, h, qrow, cen, coarse, bq, xn);
        } else if (KIND == 1 && METRIC == METRIC_L2) {
            // residual against the list's centroid; |q - c|^2 = the two interleaved half chains of the lane pair
            const float* cen = p.centroids + (int64_t)list * p.ldc;
            float acc = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                if (FULL || s < ns) {
                    const f32x4 c4 = *(const f32x4*)(cen + 8 * s + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = bq[s][e] - c4[e];
                        bq[s][e] = v;
                        acc = __fmaf_rn(v, v, acc);
                    }
                }
            }
            xn = acc + __shfl_xor(acc, 32, 64);
        } else if (METRIC == METRIC_L2) {
            xn = p.xqn[q];
        } else if (KIND == 1) {
            xn = p.coarse_dis[pi];
        }
        uint32_t base_pos = p.prefix[(int64_t)q * (np + 1) + pr];
        // pass 1: the segment slot of row 0 of this list (the pass-1 rows of a query are dense in its segment)
        uint32_t base_slot = PASS == 1 ? p.prefix1[(int64_t)q * (np + 1) + pr] : 0u;
        u64* kq = p.keys + (int64_t)q * p.stride;
        float thr_f = 0.f;
        if (PASS == 2) {
            const uint32_t tk = p.thr[q];
            if (tk >= kInvalidOrdKey) thr_f = METRIC == METRIC_L2 ? INFINITY : -INFINITY;
            else thr_f = unordkey<METRIC>(tk);
        }

        // scalar quantizer: this thread's share of a tile (two chunks of its row as 4-component pieces, and the row norm)
        // is fetched into registers a tile ahead of its decode into LDS
        uint2 sq_pf[2][4];
        float sq_rn = 0.f;
        auto sq_fetch = [&](int tt) __attribute__((always_inline)) {
            if (p.dbg & 4) return;
            const int l = tid & 63, qd = tid >> 6;
            const int ct = p.sq_ct;
            const int chb = sq_chunk_bytes(ct);
            const uint8_t* blk = p.arena_codes + (size_t)((start + tt) >> 6) * 64 * p.sq_ld + (size_t)l * chb;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = qd + 4 * i;
                if (16 * c < p.dpad) {
                    const uint8_t* ch = blk + (size_t)c * 64 * chb;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        if (ct == SQ_U8) sq_pf[i][g] = lm_sq_load_piece<SQ_U8>(ch + lm_sq_piece_off<SQ_U8>(g));
                        else if (ct == SQ_U4) sq_pf[i][g] = lm_sq_load_piece<SQ_U4>(ch + lm_sq_piece_off<SQ_U4>(g));
                        else if (ct == SQ_U6) sq_pf[i][g] = lm_sq_load_piece<SQ_U6>(ch + lm_sq_piece_off<SQ_U6>(g));
                        else sq_pf[i][g] = lm_sq_load_piece<SQ_F16>(ch + lm_sq_piece_off<SQ_F16>(g));
                    }
                }
            }
            sq_rn = 0.f;
            if (METRIC == METRIC_L2 && tid < LM_TR && tt + tid < r1) sq_rn = p.arena_rn[start + tt + tid];
        };
        if (KIND == 2) sq_fetch(r0);
        int buf = 0;
        if (KIND == 0) {
            // every load of the item set-up has landed HERE, as far as the compiler's scoreboard goes (the values pass
            // through an opaque asm): otherwise its wait for them would sit at their first use inside the tile loop and,
            // counting only its own loads, would also drain the prefetch of the next tile
#pragma unroll
            for (int s = 0; s < 16; ++s) asm volatile("" : "+v"(bq[s]));
            asm volatile("" : "+v"(xn), "+v"(thr_f));
            asm volatile("" : "+v"(kq), "+v"(base_pos), "+v"(base_slot));
        }
        for (int t = r0; t < r1; t += LM_TR, buf ^= 1) {
            const char* tile = smem + (KIND == 0 ? buf * LM_TILE_BYTES : 0);
            const float* rnt = rnl + (KIND == 0 ? buf * LM_TR : 0);
            if (KIND == 0) {
                // tile t + 1 into the other buffer (its readers passed the barrier that ended the previous iteration);
                // then everything older than those DMAs -- tile t -- must have landed
                if (t + LM_TR < r1) {
                    issue_tile(t + LM_TR, buf ^ 1);
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                lm_barrier();
            } else {
            __syncthreads(); // everybody is done with the previous tile
            // ---- tile [t, t + 64) of the list -> LDS: row r at r * 512, 16-byte chunk c at (c ^ (r & 15)) * 16
            if (p.dbg & 4) {
                // (timing experiments: the tile is not loaded)
            } else if (KIND == 2) {
                // scalar quantizer: thread (row l, quarter) turns the 16-component chunks quarter, quarter + 4 of row l into
                // floats -- the CODE values; scale and offset live in the query operands.  The chunk bytes were fetched
                // into registers one tile ahead (sq_fetch below: the 64 threads of a quarter read one chunk of all rows of
                // the chunk-major 64-row block in one coalesced sweep, kernels.h sq_code_offset)
                const int l = tid & 63, qd = tid >> 6;
                const int ct = p.sq_ct;
                char* rowp = smem + l * LM_ROWB;
                const int sw = l & 15;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int c = qd + 4 * i;
                    if (16 * c < p.dpad) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            f32x4 v;
                            if (ct == SQ_U8) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = lm_sq_comp<SQ_U8>(sq_pf[i][g], e);
                            } else if (ct == SQ_U4) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = lm_sq_comp<SQ_U4>(sq_pf[i][g], e);
                            } else if (ct == SQ_U6) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = lm_sq_comp<SQ_U6>(sq_pf[i][g], e, lm_sq_piece_shift(g));
                            } else {
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = lm_sq_comp<SQ_F16>(sq_pf[i][g], e);
                            }
                            *(f32x4*)(rowp + (((4 * c + g) ^ sw) << 4)) = v;
                        }
                    }
                }
            } else {
                // IVFPQ: thread (row l, quarter) decodes sub-quantizers quarter, quarter + 4, ...: stored byte
                // (m - l) mod M of row l (rotated block layout, kernels.h pq_code_offset) -> dsub floats of the codebook
                const int l = tid & 63, qd = tid >> 6;
                const int M = p.M, dsub = p.dsub, ch = pq_chunk_bytes(M);
                const uint8_t* blk = p.arena_codes + (size_t)((start + t) >> 6) * 64 * M;
                char* rowp = smem + l * LM_ROWB;
                const int sw = l & 15;
                for (int m = qd; m < M; m += 4) {
                    int jb = (m - l) % M;
                    if (jb < 0) jb += M;
                    const unsigned code = blk[(size_t)(jb / ch) * 64 * ch + (size_t)l * ch + (jb % ch)];
                    const float* src = p.pq_centroids + ((size_t)m * 256 + code) * dsub;
                    for (int jd = 0; jd < dsub; ++jd) {
                        const int col = m * dsub + jd;
                        *(float*)(rowp + ((((col >> 2) ^ sw)) << 4) + ((col & 3) << 2)) = src[jd];
                    }
                }
                // zero padding d .. dpad
                for (int col = p.d + qd; col < p.dpad; col += 4)
                    *(float*)(rowp + ((((col >> 2) ^ sw)) << 4) + ((col & 3) << 2)) = 0.f;
            }
            if (KIND == 2) {
                if (tid < LM_TR) rnl[tid] = sq_rn;
            } else if (tid < LM_TR) {
                float v = 0.f;
                if (METRIC == METRIC_L2 && t + tid < r1) v = p.arena_rn[start + t + tid];
                rnl[tid] = v;
            }
            __syncthreads();
            // scalar quantizer: the next tile's bytes on their way while this one is multiplied
            if (KIND == 2 && t + LM_TR < r1) sq_fetch(t + LM_TR);
            }

            // (wave-uniform condition: a wave without queries, or whose block of the list's last tile is empty, only
            // takes part in the loads and barriers)
            if (wave_active && t + wr * 32 < r1) {
                const int blk2 = wr;
                const char* rowp = tile + (blk2 * 32 + j) * LM_ROWB;
                const int sw = j & 15; // ((32 + j) & 15 == j & 15)
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                if (p.dbg & 2) {
                    // (timing experiments: no matrix work)
                } else if (FULL) {
                    f32x4 aA = *(const f32x4*)(rowp + (((0 + h) ^ sw) << 4));
                    f32x4 aB;
#pragma unroll
                    for (int s = 0; s < 16; s += 2) {
                        aB = *(const f32x4*)(rowp + (((2 * (s + 1) + h) ^ sw) << 4));
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aA[e], bq[s][e], acc, 0, 0, 0);
                        if (s + 2 < 16) aA = *(const f32x4*)(rowp + (((2 * (s + 2) + h) ^ sw) << 4));
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aB[e], bq[s + 1][e], acc, 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int s = 0; s < 16; ++s) {
                        if (s < ns) {
                            const f32x4 a = *(const f32x4*)(rowp + (((2 * s + h) ^ sw) << 4));
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], bq[s][e], acc, 0, 0, 0);
                        }
                    }
                }
                // ---- epilogue: 16 distances of this lane's query
                auto dist_of = [&](float ip, float rn) -> float {
                    if (METRIC == METRIC_L2) {
                        const float dd = __fmaf_rn(-2.f, ip, xn + rn);
                        return dd < 0.f ? 0.f : dd;
                    }
                    return xn + ip;
                };
                const int row_b = t + blk2 * 32 + 4 * h; // row of the list of acc[4 g + e]: row_b + 8 g + e
                if (PASS == 1) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 b4 = *(const f32x4*)(rnt + blk2 * 32 + 8 * g + 4 * h);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int rowl = row_b + 8 * g + e;
                            const uint32_t pos = base_pos + (uint32_t)rowl;
                            if (qv && rowl < r1 && !(p.dbg & 1))
                                kq[base_slot + (uint32_t)rowl] = ((u64)ordkey<METRIC>(dist_of(acc[4 * g + e], b4[e])) << 32) | pos;
                        }
                    }
                } else {
                    // which of the 16 pass the bound; ONE atomic per lane and block takes their slots (a returning
                    // atomic per candidate would put a memory round trip behind each of them)
                    unsigned mask = 0;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 b4 = *(const f32x4*)(rnt + blk2 * 32 + 8 * g + 4 * h);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float dis = dist_of(acc[4 * g + e], b4[e]);
                            const bool pass = (METRIC == METRIC_L2 ? dis <= thr_f : dis >= thr_f) && row_b + 8 * g + e < r1;
                            mask |= pass ? 1u << (4 * g + e) : 0u;
                        }
                    }
                    if (!qv || (p.dbg & 1)) mask = 0;
                    if (__ballot(mask != 0u)) {
                        // (wave-uniform branch) park the candidates: this lane's go behind those of the lanes before it
                        const int c = __popc(mask);
                        const int inc = (int)wave_incl_scan((unsigned)c); // inclusive scan over the lanes (DPP, common.h)
                        const int total = __builtin_amdgcn_readlane(inc, 63);
                        if (total > LM_PARK) {
                            // more candidates in one 32-row block than a slice holds (a bound that admits everything):
                            // straight to the segment, one atomic per lane
                            if (mask) {
                                uint32_t slot;
                                uint32_t* cp = p.cnt + q;
                                const uint32_t nc = (uint32_t)c;
                                asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)"
                                             : "=&v"(slot)
                                             : "v"(cp), "v"(nc)
                                             : "memory");
#pragma unroll
                                for (int g = 0; g < 4; ++g) {
                                    const f32x4 b4 = *(const f32x4*)(rnt + blk2 * 32 + 8 * g + 4 * h);
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        if (mask & (1u << (4 * g + e))) {
                                            const uint32_t pos = base_pos + (uint32_t)(row_b + 8 * g + e);
                                            if ((int64_t)slot < p.stride)
                                                kq[slot] = ((u64)ordkey<METRIC>(dist_of(acc[4 * g + e], b4[e])) << 32) | pos;
                                            ++slot;
                                        }
                                    }
                                }
                            }
                        } else {
                            if (wcnt + total > LM_PARK) flush();
                            int at = wcnt + inc - c;
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const f32x4 b4 = *(const f32x4*)(rnt + blk2 * 32 + 8 * g + 4 * h);
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    if (mask & (1u << (4 * g + e))) {
                                        const uint32_t pos = base_pos + (uint32_t)(row_b + 8 * g + e);
                                        pk_keys[at] = ((u64)ordkey<METRIC>(dist_of(acc[4 * g + e], b4[e])) << 32) | pos;
                                        pk_q[at] = (uint32_t)q;
                                        ++at;
                                    }
                                }
                            }
                            wcnt += total;
                        }
                    }
                }
            }
            if (KIND == 0) lm_barrier(); // everybody is done with this buffer before the next iteration's DMAs target it
        }
    }
    if (PASS == 2 && wcnt > 0) flush();
}

// ------------------------------------------------------------------ IVFFlat pass 2, register-fed (round 3, third kernel)
// A row of a list is only ever needed by the few wavefronts whose queries probe it, so pass 2 -- where the work is --
// shares nothing through LDS: a work item is (list, ONE 32-query block, up to 1024 rows) and belongs to ONE wavefront;
// lane (h, j) loads the 16-byte pieces 8 s + 4 h of row j of the current 32-row block straight into the registers it feeds
// the MFMAs from, and refills a[s] for the NEXT block right behind the four MFMAs that read it -- sixteen loads in flight
// per lane, each with ~3800 cycles to land.  No barrier, no DMA bookkeeping, and no load imbalance between the waves of a
// workgroup: every wavefront DRAWS its next item from a counter (items differ 3 : 1 in length, a static deal leaves the
// pipe idle behind the longest hand).  Same operands, same chain: bit-identical to the LDS kernel.
// The block loop holds no vector-memory STORE (candidates go to the wave's LDS slice; when it is full the loop is left, the
// slice flushed, the loop re-entered): with loads and stores in one loop hipcc stops counting the loads in flight and
// waits vmcnt(0) before every use.
constexpr int LR_THREADS = 256;
// parked candidates per wave: a whole block (32 rows x 32 queries) always fits an empty slice; two 60 KB workgroups per
// CU.  (The scalar quantizer's instantiations need 16-32 instead of 64 registers of A operands and were measured with
// 1024-entry slices and three workgroups per CU -- 168 registers, a few of them spilled: 0.364 / 0.359 / 0.368 ms for
// 2 / 3 / 4 workgroups per CU at nb = 1M, profiles/r03_g_ivfsq_listmajor.txt.  Occupancy is not what the loop lacks.)
template <int CT>
struct LrCfg {
    static constexpr int PARK = 1280;
    static constexpr int LDS = 4 * PARK * (8 + 4);
    static constexpr int WG_PER_CU = 2;
};

// CT: -1 = IVFFlat (fp32 rows); a SqCodeType = IVF scalar quantizer: the register a[s] holds the 4 CODES of the lane's
// operand group (a dword of 8-bit codes, 16 bits of 4-bit codes, two dwords of fp16), converted to floats right in front
// of the MFMAs that read them; scale / offset / centroid are folded into the query operands (lm_sq_query).  Same item
// walk, same refill schedule: the loads are 4 to 16 times narrower.
// PASS 1 (round 3, late): the same walk over the items of pass 1 -- every distance goes to its dense slot of the query's
// segment.  The key stores are issued from inline asm: hipcc does not see them, so it keeps COUNTING the loads in flight
// (with a store of its own in the loop it waits vmcnt(0) before every use); its counted waits stay correct -- loads
// return in order, "at most N operations outstanding" still covers every load older than the N youngest -- and merely
// become conservative by the stores in between.
template <int METRIC, bool FULL, int CT, int PASS>
__global__ void __launch_bounds__(LR_THREADS, LrCfg<CT>::WG_PER_CU) ivf_lm_flat_reg_kernel(IvfLmParams p) {
    constexpr int LR_PARK = LrCfg<CT>::PARK;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const int j = lane & 31;
    const int np = p.nprobe;
    const int ns = FULL ? 16 : (p.dpad >> 3);
    u64* pk_keys = (u64*)smem + wave * LR_PARK;
    uint32_t* pk_q = (uint32_t*)(smem + 4 * LR_PARK * 8) + wave * LR_PARK;
    int wcnt = 0; // (wave-uniform) parked candidates
    auto flush = [&]() __attribute__((always_inline)) {
        for (int e = lane; e < wcnt; e += 64) {
            const u64 key = pk_keys[e];
            const uint32_t qq = pk_q[e];
            const uint32_t slot = atomicAdd(p.cnt + qq, 1u);
            if ((int64_t)slot < p.stride) p.keys[(int64_t)qq * p.stride + slot] = key;
        }
        wcnt = 0;
        // every store above has completed, AS FAR AS THE COMPILER KNOWS TOO (vmcnt(0) through the builtin): with a store
        // possibly in flight on some path into the block loop it would not count the loop's loads (mixed loads and
        // stores on one counter) and wait vmcnt(0) at the head of every block
        __builtin_amdgcn_s_waitcnt(0x0F70);
    };

    const uint32_t it0 = p.item_bounds[PASS - 1], it1 = p.item_bounds[PASS];
    // (Eight per-XCD item queues -- neighbouring items read the same rows, workgroups b, b + 8, ... share an L2 -- were
    // measured: L2 misses 19.4 -> 14.2 GB per launch at nb = 10M, same time; slower at nb = 1M.  profiles/r03_b_*)
    uint32_t* ctr = p.item_bounds + (PASS == 2 ? 4 : 5); // next item of this pass (zeroed by the plan)
    uint32_t static_it = (uint32_t)(blockIdx.x * 4 + wave); // (dbg 64, timing experiments: a static deal, no counter)
    __shared__ uint32_t wg_it_s; // (dbg 128, timing experiments: the four waves of a workgroup draw four CONSECUTIVE items)
    for (;;) {
        uint32_t it = 0;
        bool wg_done = false;
        if (p.dbg & 128) {
            // consecutive items are the query groups of one (list, row chunk): drawn together they read the same rows at
            // the same time (one L2 miss instead of one per group); items of a chunk take the same time whatever their
            // query count, so the waves meet again at the next draw.  Measured (profiles/r03_g_ivfsq_listmajor.txt): pass 2
            // at nb = 10M 2.97 / 2.88 -> 3.05 / 3.07 ms, nb = 1M unchanged -- the re-reads are not what the loop waits for.
            __syncthreads();
            if (tid == 0) wg_it_s = atomicAdd(ctr, 4u);
            __syncthreads();
            const uint32_t base = wg_it_s;
            wg_done = it0 + base >= it1;
            it = base + (uint32_t)wave;
        } else if (p.dbg & 64) {
            it = static_it;
            static_it += gridDim.x * 4;
        } else if (lane == 0) {
            it = atomicAdd(ctr, 1u);
        }
        it = it0 + (uint32_t)__builtin_amdgcn_readfirstlane((int)it);
        if (p.dbg & 128) {
            if (wg_done) break;
            if (it >= it1) continue;
        } else if (it >= it1) {
            break;
        }
        const IvfLmItem item = p.items[it];
        const int bk = __builtin_amdgcn_readfirstlane(item.bucket);
        const int qt = __builtin_amdgcn_readfirstlane(item.qt);
        const int rt = __builtin_amdgcn_readfirstlane(item.rt);
        const int list = bk >> 1;
        const int len = (int)p.list_len[list];
        // (dbg 8, timing experiments: every item reads the rows of one of four lists -- a working set the L2 holds)
        const int64_t start = p.list_start[(p.dbg & 8) ? (list & 3) : list];
        const uint32_t pb = p.bucket_start[bk];
        const int npair = min(32, (int)(p.bucket_start[bk + 1 + item.both] - pb) - qt * 32);
        const int r0 = rt * p.rows_per_item;
        const int r1 = p.force_all ? len : min(len, r0 + p.rows_per_item);

        // ---- this lane's query
        const bool qv = j < npair;
        const uint32_t pi = p.pairs[pb + (uint32_t)(qt * 32) + (uint32_t)(qv ? j : 0)];
        const int q = (int)(pi / (uint32_t)np);
        const int pr = (int)(pi - (uint32_t)q * (uint32_t)np);
        const float* qrow = p.xq + (int64_t)q * p.ldq;
        f32x4 bq[16];
        float xn = 0.f;
        if constexpr (CT >= 0) {
            const float* cen = (METRIC == METRIC_L2 && p.sq_by_residual) ? p.centroids + (int64_t)list * p.ldc : p.sq_zero;
            const float coarse = (METRIC != METRIC_L2 && p.sq_by_residual) ? p.coarse_dis[pi] : 0.f;
            lm_sq_query<METRIC, FULL>(p, ns, h, qrow, cen, coarse, bq, xn);
        } else {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                if (FULL || s < ns) bq[s] = *(const f32x4*)(qrow + 8 * s + 4 * h);
                else bq[s] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            xn = METRIC == METRIC_L2 ? p.xqn[q] : 0.f;
        }
        const uint32_t base_pos = p.prefix[(int64_t)q * (np + 1) + pr];
        float thr_f = 0.f;
        u64* kslot = nullptr; // pass 1: the segment slot of row 0 of this list (dense: one slot per pass-1 row)
        if constexpr (PASS == 2) {
            const uint32_t tk = p.thr[q];
            if (tk >= kInvalidOrdKey) thr_f = METRIC == METRIC_L2 ? INFINITY : -INFINITY;
            else thr_f = unordkey<METRIC>(tk);
        } else {
            kslot = p.keys + (int64_t)q * p.stride + p.prefix1[(int64_t)q * (np + 1) + pr];
        }
        auto dist_of = [&](float ip, float rnv) -> float {
            if (METRIC == METRIC_L2) {
                const float dd = __fmaf_rn(-2.f, ip, xn + rnv);
                return dd < 0.f ? 0.f : dd;
            }
            return xn + ip;
        };

        int t = r0;
        while (t < r1) {
            // ---- (re-)entry: the rows of block t (row j of the block; rows behind the end of the list belong to the next
            // list or the arena's padding, index.cpp ensure_arena_: loaded, never looked at).  Always 16 pieces per row
            // (rows shorter than 128 floats: the surplus reads the following rows and is ignored).
            // The loads are issued in the order the block loop re-issues them (a[0] .. a[15]; the empty asm statements keep
            // hipcc from reordering them): its wait for a[0] at the head of a block is then "all but the 15 youngest" on
            // every path into the loop -- with a[0] loaded last here it was vmcnt(0) for every block.
            // (scalar quantizer: row j of block t is row ((start + t) & 63) + j of the 64-row code block (start + t) >> 6 --
            // lists start on block boundaries, t is a multiple of 32 --; the piece of operand group s sits in chunk
            // s >> 1 at byte ((s & 1) * 2 + h) * PIECE of the row's chunk bytes)
            typedef typename std::conditional<CT >= 0, uint2, f32x4>::type areg_t;
            const float* arow = nullptr;
            const uint8_t* crow = nullptr;     // piece of the EVEN operand groups of this lane (g = h) in chunk 0 of its row
            const uint8_t* crow_odd = nullptr; // ... of the odd ones (g = 2 + h)
            int sh_even = 0, sh_odd = 0;       // 6-bit codes: bit position of those pieces in the loaded dword pair
            int64_t cstep0 = 0, cstep1 = 0; // byte steps to the next 32-row block from an even / odd half of a code block
            int nch1 = 7;                   // last chunk of a row (rows shorter than 8 chunks: the surplus pieces re-read it)
            if constexpr (CT >= 0) {
                constexpr int CTT = CT < 0 ? 0 : CT;
                constexpr int CHB = LmSq<CTT>::CHB;
                const int64_t R = start + t;
                const uint8_t* rowc = p.arena_codes + (R >> 6) * 64 * (int64_t)p.sq_ld + (int64_t)(((int)R & 63) + j) * CHB;
                crow = rowc + lm_sq_piece_off<CTT>(h);
                crow_odd = rowc + lm_sq_piece_off<CTT>(2 + h);
                sh_even = lm_sq_piece_shift(h);
                sh_odd = lm_sq_piece_shift(2 + h);
                cstep0 = 32 * CHB;
                cstep1 = 64 * (int64_t)p.sq_ld - 32 * CHB;
                nch1 = p.sq_ld / CHB - 1;
            } else {
                arow = p.arena_vecs + (start + t + j) * p.ldv + 4 * h;
            }
            auto load_a = [&](int s) __attribute__((always_inline)) -> areg_t {
                if constexpr (CT >= 0) {
                    constexpr int CHB = LmSq<CT < 0 ? 0 : CT>::CHB;
                    const int c = FULL ? (s >> 1) : min(s >> 1, nch1);
                    return lm_sq_load_piece<CT < 0 ? 0 : CT>(((s & 1) ? crow_odd : crow) + c * 64 * CHB);
                } else {
                    return *(const f32x4*)(arow + 8 * s);
                }
            };
            auto comp_a = [&](const areg_t& v, int e, int s) __attribute__((always_inline)) -> float {
                if constexpr (CT >= 0) return lm_sq_comp<CT < 0 ? 0 : CT>(v, e, (s & 1) ? sh_odd : sh_even);
                else return v[e];
            };
            areg_t a[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                a[s] = load_a(s);
                asm volatile("" ::: "memory");
            }
            const float* rnp = p.arena_rn + start + t + 4 * h; // rn of rows 8 g + 4 h + e of the block: rnp[8 g + e]
            bool full = false; // the slice cannot take the candidates of the block in hand: leave, flush, come back
            for (; t < r1; t += 32) {
                if constexpr (CT >= 0) {
                    const int64_t step = (((int)(start & 63) + t) & 32) ? cstep1 : cstep0;
                    crow += step;
                    crow_odd += step;
                }
                else arow += 32 * p.ldv;
                rnp += 32;
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                f32x4 rn[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) rn[g] = f32x4{0.f, 0.f, 0.f, 0.f};
                // (the refills are UNCONDITIONAL -- behind the last block they read the rows that follow the list: a branch
                // around them would stop the compiler from counting the loads in flight)
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    if (FULL || s < ns) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(comp_a(a[s], e, s), bq[s][e], acc, 0, 0, 0);
                    }
                    // refill for the next block right behind the MFMAs that consumed a[s]
                    a[s] = load_a(s);
                    // the row norms of THIS block, half a block ahead of the epilogue that needs them (not carried from
                    // block to block: the register allocator would shuffle them at the head of the block, waiting for
                    // the youngest loads there)
                    if (s == 7 && METRIC == METRIC_L2) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) rn[g] = *(const f32x4*)(rnp - 32 + 8 * g);
                    }
                }
                // the instruction order above IS the schedule: four MFMAs, one load (five behind group 7), sixteen times
                // (left to itself the scheduler gathers the loads behind the last MFMA, where they have only the epilogue to
                // land in)
#pragma unroll
                for (int s = 0; s < 7; ++s) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); // VMEM read
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                if constexpr (METRIC == METRIC_L2) __builtin_amdgcn_sched_group_barrier(0x020, 5, 0);
                else __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
#pragma unroll
                for (int s = 8; s < 16; ++s) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
                // (nothing of the epilogue may be scheduled above this line)
                __builtin_amdgcn_sched_barrier(0);
                const int row_b = t + 4 * h; // row of the list of acc[4 g + e]: row_b + 8 g + e
                if constexpr (PASS == 1) {
                    // ---- epilogue of pass 1: every distance to its slot (stores the compiler does not see, see above)
                    if (!(p.dbg & 1)) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int rowl = row_b + 8 * g + e;
                                if (qv && rowl < r1) {
                                    const u64 key = ((u64)ordkey<METRIC>(dist_of(acc[4 * g + e], rn[g][e])) << 32) |
                                                    (u64)(base_pos + (uint32_t)rowl);
                                    const u64* at = kslot + rowl;
                                    asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(at), "v"(key));
                                }
                            }
                        }
                    }
                    continue;
                }
                // ---- epilogue: which of this lane's 16 distances pass its query's bound
                unsigned mask = 0;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float dis = dist_of(acc[4 * g + e], rn[g][e]);
                        const bool pass = (METRIC == METRIC_L2 ? dis <= thr_f : dis >= thr_f) && row_b + 8 * g + e < r1;
                        mask |= pass ? 1u << (4 * g + e) : 0u;
                    }
                }
                if (!qv || (p.dbg & 1)) mask = 0;
                if (__ballot(mask != 0u)) {
                    // (wave-uniform branch) park the candidates: this lane's go behind those of the lanes before it
                    const int c = __popc(mask);
                    const int inc = (int)wave_incl_scan((unsigned)c); // inclusive scan over the lanes (DPP, common.h)
                    const int total = __builtin_amdgcn_readlane(inc, 63);
                    if (wcnt + total > LR_PARK) {
                        full = true; // (block t is redone after the flush: its candidates were not parked)
                        break;
                    }
                    int at = wcnt + inc - c;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (mask & (1u << (4 * g + e))) {
                                const uint32_t pos = base_pos + (uint32_t)(row_b + 8 * g + e);
                                pk_keys[at] = ((u64)ordkey<METRIC>(dist_of(acc[4 * g + e], rn[g][e])) << 32) | pos;
                                pk_q[at] = (uint32_t)q;
                                ++at;
                            }
                        }
                    }
                    wcnt += total;
                }
            }
            if (full) flush();
        }
    }
    if (wcnt > 0) flush();
}

template <int METRIC, int CT, int PASS>
static void lr_launch2(const IvfLmParams& p, int grid_blocks, hipStream_t stream) {
    constexpr int LR_LDS_TOTAL = LrCfg<CT>::LDS;
    const int lds = PASS == 2 ? LR_LDS_TOTAL : 0; // (pass 1 parks nothing)
    if (p.dpad == 128) {
        HIP_CHECK(hipFuncSetAttribute((const void*)ivf_lm_flat_reg_kernel<METRIC, true, CT, PASS>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, LR_LDS_TOTAL));
        hipLaunchKernelGGL((ivf_lm_flat_reg_kernel<METRIC, true, CT, PASS>), dim3((unsigned)grid_blocks), dim3(LR_THREADS), lds,
                           stream, p);
    } else {
        HIP_CHECK(hipFuncSetAttribute((const void*)ivf_lm_flat_reg_kernel<METRIC, false, CT, PASS>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, LR_LDS_TOTAL));
        hipLaunchKernelGGL((ivf_lm_flat_reg_kernel<METRIC, false, CT, PASS>), dim3((unsigned)grid_blocks), dim3(LR_THREADS), lds,
                           stream, p);
    }
}
template <int METRIC, int CT>
static void lr_launch(const IvfLmParams& p, int pass, int grid_blocks, hipStream_t stream) {
    if (pass == 1) lr_launch2<METRIC, CT, 1>(p, grid_blocks, stream);
    else lr_launch2<METRIC, CT, 2>(p, grid_blocks, stream);
}
template <int METRIC>
static void lr_launch_sq(const IvfLmParams& p, int pass, int grid_blocks, hipStream_t stream) {
    switch (p.sq_ct) {
        case SQ_U8: lr_launch<METRIC, SQ_U8>(p, pass, grid_blocks, stream); break;
        case SQ_U4: lr_launch<METRIC, SQ_U4>(p, pass, grid_blocks, stream); break;
        case SQ_U6: lr_launch<METRIC, SQ_U6>(p, pass, grid_blocks, stream); break;
        case SQ_F16: lr_launch<METRIC, SQ_F16>(p, pass, grid_blocks, stream); break;
        default: FA_THROW_MSG("list-major scan: scalar-quantizer code type not supported");
    }
}
// pass 1 by the register-fed kernel?  Measured at nb = 1M (profiles/r03_g_ivfsq_listmajor.txt): scalar quantizer 0.353 ms
// with the LDS-tile kernel (its threads decode the tile between two barriers) against 0.225 register-fed; IVFFlat 0.229
// with the LDS-DMA tiles against 0.276 register-fed (pass-1 items hold 10-17 queries: a DMA tile feeds two half-idle
// waves for free, a register-fed wave pays every row's load latency alone).  FAISS_AMD_LM_P1_REG = 0 / 1 overrides.
static bool lm_p1_reg(int kind) {
    static const char* e = experiment_env("FAISS_AMD_LM_P1_REG");
    if (e) return atoi(e) != 0;
    return kind == 2;
}

// ------------------------------------------------------------------ IVFPQ, codebook in LDS (round 3, third kernel)
// The generic kernel above decodes an IVFPQ tile through 64 codebook gathers per row from L2 -- more expensive than the
// matrix work the tile feeds (profiles/r03_b_listmajor_experiments.txt: 0.44 of 0.92 ms).  Here the whole codebook
// [M][256][dsub] (d KB of fp32: 128 KB at d = 128) lives in LDS for the life of a persistent workgroup, and a lane builds
// its MFMA A operand -- 4 consecutive coordinates of ITS row -- straight from it: code byte(s) of the row, then one to
// four LDS gathers.  No decoded tile, no codebook traffic outside the CU.  Same operands, same MFMA chain as the generic
// kernel: bit-identical results.
// A wavefront works alone (the second kernel of this round ran 8 waves in lock step over a 128-row tile, two barriers a
// tile, every operand decoded twice -- by the waves of the two query blocks: the LDS side, not the matrix pipe, was its
// bound, dbg = 3 in profiles/r03_b_*): it draws an item (list, up to 64 queries, row chunk) from a counter, keeps the
// B operands of both 32-query blocks in 128 registers, and walks the rows in blocks of 32: the block's code bytes go
// global -> registers (one block ahead) -> a private 32-row LDS slice, un-rotated on the way in (pq_code_offset), so that
// the code bytes of an operand are one aligned LDS read.  Code reads run two operand pairs ahead of the MFMAs, codebook
// gathers one pair ahead (explicit software pipeline: left alone hipcc hoists all 64 LDS reads of a block to its top and
// spills).  One decoded operand feeds the MFMAs of both query blocks; an item with <= 32 queries skips the second
// block's.  No barrier after the codebook load.
// DS: 1 / 2 = dsub itself, 4 = any multiple of 4 (the 4 coordinates of an operand then lie inside one sub-vector).
constexpr int LQ_THREADS = 512;
constexpr int LQ_BR = 32;    // rows per block
constexpr int LQ_PARK = 96;  // parked candidates per wave
struct LqLayout {
    int cb_bytes, rs, off_codes, off_park, total;
};
__host__ __device__ static inline LqLayout lq_layout(int d, int M) {
    LqLayout L;
    L.cb_bytes = d * 256 * 4;
    L.rs = ((M + 15) & ~15) + 16;                   // bytes per row of a code slice (16-byte pieces; + 16: rows on different banks)
    L.off_codes = (L.cb_bytes + 15) & ~15;
    L.off_park = L.off_codes + 8 * LQ_BR * L.rs;
    L.total = L.off_park + 8 * LQ_PARK * (8 + 4);
    return L;
}
bool ivf_lm_pq_lds_supported(int d, int dpad, int M) {
    if (dpad > 128 || M < 4 || (M & 3) || d % M) return false;
    const int dsub = d / M;
    if (!(dsub == 1 || dsub == 2 || (dsub & 3) == 0)) return false;
    return lq_layout(d, M).total <= 160 * 1024;
}

template <int METRIC, int PASS, int DS, bool FULL>
__global__ void __launch_bounds__(LQ_THREADS, 2) ivf_lm_pq_kernel(IvfLmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const int j = lane & 31;
    const int np = p.nprobe;
    const int M = p.M, dsub = p.dsub;
    const int ns = FULL ? 16 : (p.dpad >> 3);
    const LqLayout L = lq_layout(p.d, M);
    const float* cb = (const float*)smem;

    // ---- the codebook, once per workgroup
    {
        const f32x4* src = (const f32x4*)p.pq_centroids;
        f32x4* dst = (f32x4*)smem;
        for (int i = tid; i < p.d * 64; i += LQ_THREADS) dst[i] = src[i];
    }
    unsigned char* crow = (unsigned char*)(smem + L.off_codes) + (wave * LQ_BR + j) * L.rs; // this lane's row of the slice
    u64* pk_keys = (u64*)(smem + L.off_park) + wave * LQ_PARK;
    uint32_t* pk_q = (uint32_t*)(smem + L.off_park + 8 * LQ_PARK * 8) + wave * LQ_PARK;
    int wcnt = 0; // (wave-uniform) parked candidates of pass 2
    auto flush = [&]() __attribute__((always_inline)) {
        for (int e = lane; e < wcnt; e += 64) {
            const u64 key = pk_keys[e];
            const uint32_t qq = pk_q[e];
            uint32_t slot;
            uint32_t* cp = p.cnt + qq;
            const uint32_t one = 1u;
            asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(slot) : "v"(cp), "v"(one) : "memory");
            if ((int64_t)slot < p.stride) p.keys[(int64_t)qq * p.stride + slot] = key;
        }
        wcnt = 0;
    };
    __syncthreads();

    const int ch = pq_chunk_bytes(M);
    const int nch = M >> 4;            // 16-byte pieces of a stored row (ch == 16)
    const int cpl = (nch + 1) >> 1;    // ... per lane: lane (j, h) moves pieces h * cpl .. of row j (at most 4: M <= 112)
    const uint32_t it0 = p.item_bounds[PASS - 1], it1 = p.item_bounds[PASS];
    uint32_t* ctr = p.item_bounds + (PASS == 1 ? 5 : 4); // next item of this pass (zeroed by the plan)
    for (;;) {
        uint32_t it = 0;
        if (lane == 0) it = atomicAdd(ctr, 1u);
        it = it0 + (uint32_t)__builtin_amdgcn_readfirstlane((int)it);
        if (it >= it1) break;
        const IvfLmItem item = p.items[it];
        const int bk = __builtin_amdgcn_readfirstlane(item.bucket);
        const int qt = __builtin_amdgcn_readfirstlane(item.qt);
        const int rt = __builtin_amdgcn_readfirstlane(item.rt);
        const int list = bk >> 1;
        const int len = (int)p.list_len[list];
        const int64_t start = p.list_start[list];
        const uint32_t pb = p.bucket_start[bk];
        const int npair = min(kLmQueriesPerItem, (int)(p.bucket_start[bk + 1 + item.both] - pb) - qt * kLmQueriesPerItem);
        const bool two = npair > 32; // (wave-uniform) the second query block holds queries
        const int r0 = rt * p.rows_per_item;
        const int r1 = p.force_all ? len : min(len, r0 + p.rows_per_item);

        // ---- the code bytes of a block: global -> registers -> this wave's slice
        const bool fast = ch == 16 && nch <= 4; // lane (j, h) moves the 16-byte pieces h * cpl .. (at most two) of row j
        uint4 creg[2];
        auto fetch = [&](int t) __attribute__((always_inline)) {
            if (fast) {
                const int64_t row = start + t + j; // arena row (inside the list's capacity: a multiple of 64 rows)
                const unsigned char* src = p.arena_codes + (size_t)(row >> 6) * 64 * M + (size_t)(row & 63) * 16;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int c = h * cpl + i;
                    if (i < cpl && c < nch) creg[i] = *(const uint4*)(src + (size_t)c * 1024);
                }
            }
        };
        auto stage = [&](int t) __attribute__((always_inline)) {
            const int64_t row = start + t + j;
            if (fast) {
                const int lrot = (int)(row & 63) % M; // stored byte x of the row is sub-quantizer (x + row) mod M
                // (the registers are opaque up to here: hipcc otherwise takes the bytes apart right behind the loads of
                // fetch() -- a wait for the loads a block early, 32 live registers instead of 8)
#pragma unroll
                for (int i = 0; i < 2; ++i) asm volatile("" : "+v"(creg[i].x), "+v"(creg[i].y), "+v"(creg[i].z), "+v"(creg[i].w));
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int c = h * cpl + i;
                    if (i < cpl && c < nch) {
                        const unsigned w[4] = {creg[i].x, creg[i].y, creg[i].z, creg[i].w};
#pragma unroll
                        for (int b = 0; b < 16; ++b) {
                            int m = 16 * c + b + lrot;
                            m -= m >= M ? M : 0;
                            crow[m] = (unsigned char)(w[b >> 2] >> (8 * (b & 3)));
                        }
                    }
                }
            } else {
                for (int m = h; m < M; m += 2) crow[m] = p.arena_codes[pq_code_offset(M, row, m)];
            }
            // the slice is written and read by this wavefront only: LDS executes a wave's accesses in order, the compiler
            // must not move the reads of other lanes' bytes above the writes
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        };
        fetch(r0);

        // ---- this lane's queries: one per 32-query block
        bool qv[2];
        int q[2];
        uint32_t base_pos[2], base_slot[2];
        float xn[2], thr_f[2];
        f32x4 bq[2][16];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int my = b * 32 + j;
            qv[b] = my < npair;
            const uint32_t pi = p.pairs[pb + (uint32_t)(qt * kLmQueriesPerItem) + (uint32_t)(qv[b] ? my : 0)];
            q[b] = (int)(pi / (uint32_t)np);
            const int pr = (int)(pi - (uint32_t)q[b] * (uint32_t)np);
            const float* qrow = p.xq + (int64_t)q[b] * p.ldq;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                if ((FULL || s < ns) && (b == 0 || two)) bq[b][s] = *(const f32x4*)(qrow + 8 * s + 4 * h);
                else bq[b][s] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            xn[b] = 0.f;
            if (METRIC == METRIC_L2) {
                if (b == 0 || two) {
                    const float* cen = p.centroids + (int64_t)list * p.ldc;
                    float acc = 0.f;
#pragma unroll
                    for (int s = 0; s < 16; ++s) {
                        if (FULL || s < ns) {
                            const f32x4 c4 = *(const f32x4*)(cen + 8 * s + 4 * h);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float v = bq[b][s][e] - c4[e];
                                bq[b][s][e] = v;
                                acc = __fmaf_rn(v, v, acc);
                            }
                        }
                    }
                    xn[b] = acc + __shfl_xor(acc, 32, 64);
                }
            } else {
                xn[b] = p.coarse_dis[pi];
            }
            base_pos[b] = p.prefix[(int64_t)q[b] * (np + 1) + pr];
            base_slot[b] = PASS == 1 ? p.prefix1[(int64_t)q[b] * (np + 1) + pr] : 0u;
            thr_f[b] = 0.f;
            if (PASS == 2) {
                const uint32_t tk = p.thr[q[b]];
                if (tk >= kInvalidOrdKey) thr_f[b] = METRIC == METRIC_L2 ? INFINITY : -INFINITY;
                else thr_f[b] = unordkey<METRIC>(tk);
            }
        }
        auto dist_of = [&](float ip, float rn, float xnb) -> float {
            if (METRIC == METRIC_L2) {
                const float dd = __fmaf_rn(-2.f, ip, xnb + rn);
                return dd < 0.f ? 0.f : dd;
            }
            return xnb + ip;
        };

        for (int t = r0; t < r1; t += LQ_BR) {
            if (!(p.dbg & 16)) stage(t); // (dbg: timing experiments, results wrong)
            const bool more = t + LQ_BR < r1;
            // code byte(s) of operand s of this lane's row: one aligned read of the un-rotated slice
            // (FULL means d == 128 for every shape this kernel serves.  Otherwise the last operand may start at a
            // coordinate >= d (dpad > d): it is built from whatever LDS holds there and replaced by zeros with a select,
            // no divergent branch in the operand stream)
            auto codes_of = [&](int s_) __attribute__((always_inline)) -> unsigned {
                if (!(FULL || s_ < ns)) return 0u;
                const int kb = 8 * s_ + 4 * h; // first coordinate of the operand
                if (DS == 2) return *(const unsigned short*)(crow + (kb >> 1));
                if (DS == 1) return *(const unsigned*)(crow + kb);
                return crow[kb / dsub];
            };
            auto operand_of = [&](int s_, unsigned cw) __attribute__((always_inline)) -> f32x4 {
                f32x4 a = {0.f, 0.f, 0.f, 0.f};
                if (!(FULL || s_ < ns)) return a;
                const int kb = 8 * s_ + 4 * h;
                if (DS == 2) {
                    const int m0 = kb >> 1;
                    const float2 lo = *(const float2*)(cb + ((m0 << 8) + (cw & 255u)) * 2);
                    const float2 hi = *(const float2*)(cb + (((m0 + 1) << 8) + (cw >> 8)) * 2);
                    a = f32x4{lo.x, lo.y, hi.x, hi.y};
                } else if (DS == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[e] = cb[((kb + e) << 8) + ((cw >> (8 * e)) & 255u)];
                } else {
                    const int m = kb / dsub, off = kb - m * dsub;
                    a = *(const f32x4*)(cb + ((m << 8) + (cw & 255u)) * dsub + off);
                }
                if (!FULL && kb >= p.d) a = f32x4{0.f, 0.f, 0.f, 0.f};
                return a;
            };
            f32x16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc0[r] = 0.f;
                acc1[r] = 0.f;
            }
            f32x4 rn4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) rn4[g] = f32x4{0.f, 0.f, 0.f, 0.f};
            auto operands_and_mfmas = [&](auto two_c) __attribute__((always_inline)) {
                constexpr bool TWO = decltype(two_c)::value;
                // pairs of operands: codes of pair g + 2 | gathers of pair g + 1 | MFMAs of pair g
                unsigned cw[3][2];
                f32x4 av[2][2];
                cw[0][0] = codes_of(0), cw[0][1] = codes_of(1);
                cw[1][0] = codes_of(2), cw[1][1] = codes_of(3);
                av[0][0] = operand_of(0, cw[0][0]), av[0][1] = operand_of(1, cw[0][1]);
    #pragma unroll
                for (int g = 0; g < 8; ++g) {
                    if (g + 2 < 8) cw[(g + 2) % 3][0] = codes_of(2 * g + 4), cw[(g + 2) % 3][1] = codes_of(2 * g + 5);
                    if (g + 1 < 8)
                        av[(g + 1) & 1][0] = operand_of(2 * g + 2, cw[(g + 1) % 3][0]), av[(g + 1) & 1][1] = operand_of(2 * g + 3, cw[(g + 1) % 3][1]);
                    if (g == 1 && more) fetch(t + LQ_BR);
                    if (g == 4 && METRIC == METRIC_L2) {
                        // |r^|^2 of the block's rows (lane: rows 4 h + 8 g + e), consumed by the epilogue
    #pragma unroll
                        for (int gg = 0; gg < 4; ++gg) rn4[gg] = *(const f32x4*)(p.arena_rn + start + t + 8 * gg + 4 * h);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (FULL || 2 * g < ns) {
    #pragma unroll
                        for (int u = 0; u < 2; ++u)
    #pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g & 1][u][e], bq[0][2 * g + u][e], acc0, 0, 0, 0);
                                if (TWO) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g & 1][u][e], bq[1][2 * g + u][e], acc1, 0, 0, 0);
                            }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            if (p.dbg & 2) {
            } else if (two) operands_and_mfmas(std::true_type{});
            else operands_and_mfmas(std::false_type{});
            if (p.dbg & 32) continue;
            // ---- epilogue: 16 distances of each of this lane's queries (as in the generic kernel)
            const int row_b = t + 4 * h; // row of the list of acc[4 g + e]: row_b + 8 g + e
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                if (b == 1 && !two) break;
                const f32x16& acc = b == 0 ? acc0 : acc1;
                if (PASS == 1) {
                    u64* kq = p.keys + (int64_t)q[b] * p.stride;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int rowl = row_b + 8 * g + e;
                            const uint32_t pos = base_pos[b] + (uint32_t)rowl;
                            if (qv[b] && rowl < r1 && !(p.dbg & 1))
                                kq[base_slot[b] + (uint32_t)rowl] = ((u64)ordkey<METRIC>(dist_of(acc[4 * g + e], rn4[g][e], xn[b])) << 32) | pos;
                        }
                    }
                } else {
                    unsigned mask = 0;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float dis = dist_of(acc[4 * g + e], rn4[g][e], xn[b]);
                            const bool pass = (METRIC == METRIC_L2 ? dis <= thr_f[b] : dis >= thr_f[b]) && row_b + 8 * g + e < r1;
                            mask |= pass ? 1u << (4 * g + e) : 0u;
                        }
                    }
                    if (!qv[b] || (p.dbg & 1)) mask = 0;
                    if (__ballot(mask != 0u)) {
                        const int c = __popc(mask);
                        const int inc = (int)wave_incl_scan((unsigned)c); // inclusive scan over the lanes (DPP, common.h)
                        const int total = __builtin_amdgcn_readlane(inc, 63);
                        if (total > LQ_PARK) {
                            if (mask) {
                                u64* kq = p.keys + (int64_t)q[b] * p.stride;
                                uint32_t slot;
                                uint32_t* cp = p.cnt + q[b];
                                const uint32_t nc = (uint32_t)c;
                                asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)"
                                             : "=&v"(slot)
                                             : "v"(cp), "v"(nc)
                                             : "memory");
#pragma unroll
                                for (int g = 0; g < 4; ++g) {
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        if (mask & (1u << (4 * g + e))) {
                                            const uint32_t pos = base_pos[b] + (uint32_t)(row_b + 8 * g + e);
                                            if ((int64_t)slot < p.stride)
                                                kq[slot] = ((u64)ordkey<METRIC>(dist_of(acc[4 * g + e], rn4[g][e], xn[b])) << 32) | pos;
                                            ++slot;
                                        }
                                    }
                                }
                            }
                        } else {
                            if (wcnt + total > LQ_PARK) flush();
                            int at = wcnt + inc - c;
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    if (mask & (1u << (4 * g + e))) {
                                        const uint32_t pos = base_pos[b] + (uint32_t)(row_b + 8 * g + e);
                                        pk_keys[at] = ((u64)ordkey<METRIC>(dist_of(acc[4 * g + e], rn4[g][e], xn[b])) << 32) | pos;
                                        pk_q[at] = (uint32_t)q[b];
                                        ++at;
                                    }
                                }
                            }
                            wcnt += total;
                        }
                    }
                }
            }
            // (the next block's stage() writes the slice: every read above was issued before it, LDS keeps a wave's order)
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (PASS == 2 && wcnt > 0) flush();
}

template <int METRIC, int PASS>
static void lq_launch(const IvfLmParams& p, int grid_blocks, hipStream_t stream) {
    const int lds = lq_layout(p.d, p.M).total;
    const int ds = p.dsub == 1 ? 1 : p.dsub == 2 ? 2 : 4;
    const bool full = p.dpad == 128;
#define FA_LQ(DS_, F_)                                                                                                  \
    do {                                                                                                                \
        HIP_CHECK(hipFuncSetAttribute((const void*)ivf_lm_pq_kernel<METRIC, PASS, DS_, F_>,                             \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds));                                \
        hipLaunchKernelGGL((ivf_lm_pq_kernel<METRIC, PASS, DS_, F_>), dim3((unsigned)grid_blocks), dim3(LQ_THREADS), lds, \
                           stream, p);                                                                                  \
    } while (0)
    if (ds == 1) {
        if (full) FA_LQ(1, true);
        else FA_LQ(1, false);
    } else if (ds == 2) {
        if (full) FA_LQ(2, true);
        else FA_LQ(2, false);
    } else {
        if (full) FA_LQ(4, true);
        else FA_LQ(4, false);
    }
#undef FA_LQ
}

template <int METRIC, int KIND, int PASS>
static void lm_launch3(const IvfLmParams& p, int grid_blocks, hipStream_t stream) {
    const int lds = KIND == 0 ? LM2_LDS_TOTAL : LM_LDS_TOTAL_P;
    if (p.dpad == 128) {
        HIP_CHECK(hipFuncSetAttribute((const void*)ivf_lm_scan_kernel<METRIC, KIND, PASS, true>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        hipLaunchKernelGGL((ivf_lm_scan_kernel<METRIC, KIND, PASS, true>), dim3((unsigned)grid_blocks), dim3(LM_THREADS), lds,
                           stream, p);
    } else {
        HIP_CHECK(hipFuncSetAttribute((const void*)ivf_lm_scan_kernel<METRIC, KIND, PASS, false>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        hipLaunchKernelGGL((ivf_lm_scan_kernel<METRIC, KIND, PASS, false>), dim3((unsigned)grid_blocks), dim3(LM_THREADS), lds,
                           stream, p);
    }
}
template <int METRIC, int KIND>
static void lm_launch2(const IvfLmParams& p, int pass, int grid_blocks, hipStream_t stream) {
    if (pass == 1) lm_launch3<METRIC, KIND, 1>(p, grid_blocks, stream);
    else lm_launch3<METRIC, KIND, 2>(p, grid_blocks, stream);
}
int ivf_lm_blocks_per_cu(int kind) {
    return kind == 1 ? 3 : 2;
}
static bool lm_flat_lds_env() {
    static const char* e = experiment_env("FAISS_AMD_LM_FLAT_LDS"); // timing experiments: 1 = the LDS-tile kernel for IVFFlat
    return e && atoi(e) == 1;
}
int ivf_lm_queries_per_item(int kind) {
    // IVFFlat and the scalar quantizer: pass 2 by the register-fed kernel (one wavefront and 32 queries per item);
    // pass 1 walks the same 32-query items with the LDS-tile kernel
    return (kind == 0 || kind == 2) && !lm_flat_lds_env() ? 32 : kLmQueriesPerItem;
}
static bool lm_use_pq_lds(const IvfLmParams& p) {
    static const char* e = experiment_env("FAISS_AMD_LM_PQ_GENERIC"); // timing experiments: 1 = the generic (L2-gather) kernel
    return p.kind == 1 && ivf_lm_pq_lds_supported(p.d, p.dpad, p.M) && !(e && atoi(e) == 1);
}
int ivf_lm_grid_blocks(const IvfLmParams& p, int num_cus) {
    if (lm_use_pq_lds(p)) return num_cus; // one 8-wave workgroup per CU (the codebook fills its LDS)
    if (p.kind == 2) { // register-fed kernels in both passes (FAISS_AMD_LM_SQ_WG: occupancy experiments)
        static const char* e = experiment_env("FAISS_AMD_LM_SQ_WG");
        const int per = e ? std::max(1, atoi(e)) : LrCfg<0>::WG_PER_CU;
        return per * num_cus / 8 * 8;
    }
    // IVFFlat: two 4-wave workgroups per CU for both kernels (pass 1: LDS tiles, 2 x 75 KB; pass 2: registers)
    return ivf_lm_blocks_per_cu(p.kind) * num_cus / 8 * 8;
}
void launch_ivf_lm_scan(const IvfLmParams& p, int pass, int grid_blocks, hipStream_t stream) {
    if (p.nq == 0) return;
    FA_THROW_IF_NOT(ivf_lm_supported(p.kind, p.dpad, p.M, p.d) && (pass == 1 || pass == 2) && grid_blocks > 0);
    FA_THROW_IF_NOT(p.ldq % 4 == 0 && (p.kind != 0 || p.ldv % 4 == 0) && (p.kind == 0 || p.ldc % 4 == 0));
    if (p.kind == 0 && p.qpi == 32 && (pass == 2 || lm_p1_reg(0))) {
        if (p.metric == METRIC_L2) lr_launch<METRIC_L2, -1>(p, pass, grid_blocks, stream);
        else lr_launch<METRIC_INNER_PRODUCT, -1>(p, pass, grid_blocks, stream);
        HIP_CHECK(hipGetLastError());
        return;
    }
    if (p.kind == 2) {
        FA_THROW_IF_NOT(p.sq_s && p.sq_b && p.sq_zero && p.arena_codes && p.sq_ld > 0 && (p.metric != METRIC_L2 || p.arena_rn));
        if (p.qpi == 32 && (pass == 2 || lm_p1_reg(2))) {
            if (p.metric == METRIC_L2) lr_launch_sq<METRIC_L2>(p, pass, grid_blocks, stream);
            else lr_launch_sq<METRIC_INNER_PRODUCT>(p, pass, grid_blocks, stream);
        } else {
            if (p.metric == METRIC_L2) lm_launch2<METRIC_L2, 2>(p, pass, grid_blocks, stream);
            else lm_launch2<METRIC_INNER_PRODUCT, 2>(p, pass, grid_blocks, stream);
        }
        HIP_CHECK(hipGetLastError());
        return;
    }
    FA_THROW_IF_NOT(p.qpi == kLmQueriesPerItem || p.kind == 0);
    if (lm_use_pq_lds(p)) {
        if (p.metric == METRIC_L2) {
            if (pass == 1) lq_launch<METRIC_L2, 1>(p, grid_blocks, stream);
            else lq_launch<METRIC_L2, 2>(p, grid_blocks, stream);
        } else {
            if (pass == 1) lq_launch<METRIC_INNER_PRODUCT, 1>(p, grid_blocks, stream);
            else lq_launch<METRIC_INNER_PRODUCT, 2>(p, grid_blocks, stream);
        }
        HIP_CHECK(hipGetLastError());
        return;
    }
    if (p.metric == METRIC_L2) {
        if (p.kind == 0) lm_launch2<METRIC_L2, 0>(p, pass, grid_blocks, stream);
        else lm_launch2<METRIC_L2, 1>(p, pass, grid_blocks, stream);
    } else {
        if (p.kind == 0) lm_launch2<METRIC_INNER_PRODUCT, 0>(p, pass, grid_blocks, stream);
        else lm_launch2<METRIC_INNER_PRODUCT, 1>(p, pass, grid_blocks, stream);
    }
    HIP_CHECK(hipGetLastError());
}

} // namespace faiss_amd
