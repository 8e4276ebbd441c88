// faiss_amd/csrc/flat_filter.hip -- brute-force search at the fp16 MFMA rate with fp32-exact
// results: a candidate FILTER on v_mfma_f32_32x32x16_f16 followed by an exact fp32 RE-RANK.
//
// gfx950's fp32-input MFMA runs at the fp32 vector rate (157 TFLOP/s); its f16 MFMA is 16x
// faster (2.5 PFLOP/s dense).  The reference's contract for this path is fp32 in / fp32 out
// (faiss/gpu/impl/Distance.cu:120-406 runDistance<float>; faiss/utils/distances.cpp:424-511 on the
// CPU), so the fast units may only be used where they cannot change the answer:
//
//   1. filter (flat_filter_kernel): approximate scores  t~(q,y) = <fp16(q), fp16(y)> - |y|^2/2
//      (L2; plain inner product for IP; larger is better).  A rigorous bound
//      e_q >= |t~(q,y) - s(q,y)| for every y (s = the exact score the fp32 path would produce) is
//      computed per query from the fp16 rounding model (flat_filter_err_bound).  Every row whose
//      approximate score is within 2*e_q of the running k-th best approximate score is kept in the
//      (query, split) reservoir: if the k best by t~ all have exact score >= t~_k - e_q, then
//      a row with t~ < t~_k - 2 e_q has exact score < t~_k - e_q and cannot be among the k best.
//      The candidate set is therefore a SUPERSET of the exact top-k, whatever the data.
//   2. re-rank (flat_rerank_kernel): exact fp32 distances of the surviving candidates (k plus
//      the rows inside the error band, usually a handful) with the very same fmaf chain as the
//      fp32 MFMA kernel (flat_kernels.hip / oracle orc_ip_chain), then k-selection under
//      (distance, id).  Distances and labels are bit-identical to the fp32 path.
//   3. overflow: a query whose band does not fit its reservoir (adversarial data: thousands of
//      rows within fp16 rounding of the k-th neighbour, or fp16 range overflow) is flagged and
//      re-run through the exact fp32 MFMA kernel by the host code.  Never a silent approximation.
#include <type_traits>
#include "kernels.h"
#include "wg_select.h"

namespace faiss_amd {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// score keys: larger score = better = smaller key (same mapping as the inner-product ordkey)
__device__ __forceinline__ uint32_t score_key(float t) {
    return ordkey<METRIC_INNER_PRODUCT>(t);
}
__device__ __forceinline__ float key_score(uint32_t k) {
    return unordkey<METRIC_INNER_PRODUCT>(k);
}

// threshold strictly below (t_k - 2e): keeps exact ties of the k-th score inside the band even
// when e underflows to 0
__device__ __forceinline__ float band_threshold(float tk, float e) {
    return tk - 2.f * e - 9.6e-7f * fabsf(tk) - 1e-37f;
}

// ---------------------------------------------------------------------------------
// fp32 -> fp16 copies (database at add time, queries per search)
// ---------------------------------------------------------------------------------
// dst[i][0..dh) = fp16(src[i][0..d)) zero padded; absmax_bits = max over |x| (uint bits of a
// non-negative float order like the float); flags[i] = 1 when row i leaves the fp16 range or
// holds a NaN (flags may be null)
__global__ void convert_f16_kernel(const float* __restrict__ src, int64_t ld_src, int64_t n, int d,
                                   _Float16* __restrict__ dst, int dh, unsigned* __restrict__ absmax_bits,
                                   uint32_t* __restrict__ flags) {
    const int64_t i = blockIdx.x;
    const float* r = src + i * ld_src;
    _Float16* o = dst + i * dh;
    float mx = 0.f;
    bool bad = false;
    for (int c = threadIdx.x; c < dh; c += blockDim.x) {
        const float v = c < d ? r[c] : 0.f;
        const float a = fabsf(v);
        if (!(a <= 65000.f)) bad = true; // NaN, inf, or beyond the fp16 normal range
        mx = fmaxf(mx, a);
        o[c] = (_Float16)v;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    const bool anybad = __ballot(bad) != 0ull;
    if (threadIdx.x == 0) {
        if (absmax_bits) atomicMax(absmax_bits, anybad ? 0x7f800000u : __float_as_uint(mx));
        if (flags) flags[i] = anybad ? 1u : 0u;
    }
}

void launch_convert_f16(const float* src, int64_t ld_src, int64_t n, int d, void* dst, int dh,
                        unsigned* absmax_bits, uint32_t* flags, hipStream_t stream) {
    if (n == 0) return;
    hipLaunchKernelGGL(convert_f16_kernel, dim3((unsigned)n), dim3(64), 0, stream, src, ld_src, n, d,
                       (_Float16*)dst, dh, absmax_bits, flags);
    HIP_CHECK(hipGetLastError());
}

// Per-search query preparation in ONE launch (the searches of small batches are launch-latency bound): fp16 copy +
// range flag of convert_f16_kernel, |q|^2 as the sequential fmaf chain of l2_norms_kernel (lane 0), and the reset
// of the overflow counter the re-rank kernel appends to.  One wavefront per query.
__global__ void __launch_bounds__(64) prep_queries_kernel(const float* __restrict__ xq_pad, int64_t ld, int64_t n, int d,
                                                          int dpad, _Float16* __restrict__ qh, int dh,
                                                          uint32_t* __restrict__ flags, float* __restrict__ qnorm,
                                                          unsigned* __restrict__ counter, ClearList clear) {
    const int64_t i = blockIdx.x;
    for (int k = 0; k < clear.cnt; ++k)
        for (uint32_t w = (uint32_t)i * 64u + threadIdx.x; w < clear.n[k]; w += gridDim.x * 64u) clear.p[k][w] = 0u;
    const float* r = xq_pad + i * ld;
    _Float16* o = qh + i * dh;
    bool bad = false;
    for (int c = threadIdx.x; c < dh; c += 64) {
        const float v = c < d ? r[c] : 0.f;
        if (!(fabsf(v) <= 65000.f)) bad = true; // NaN, inf, or beyond the fp16 normal range
        o[c] = (_Float16)v;
    }
    const bool anybad = __ballot(bad) != 0ull;
    // |q|^2: ONE sequential chain over k = 0 .. dpad-1 (the order of l2_norms_kernel / the oracle).  The coordinates are
    // loaded by all lanes, 64 at a time, and handed to the chain through v_readlane (a scalar operand of the fma)
    // instead of 128 dependent single-lane loads (40 us per 10k queries, round 2 profile); the zeros behind dpad are
    // fmaf(0, 0, acc) = acc, exact no-ops.
    float acc = 0.f;
    for (int base = 0; base < dpad; base += 64) {
        const int c = base + (int)threadIdx.x;
        const float v = c < dpad ? r[c] : 0.f;
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            const float vk = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), k));
            acc = __fmaf_rn(vk, vk, acc);
        }
    }
    if (threadIdx.x == 0) {
        flags[i] = anybad ? 1u : 0u;
        qnorm[i] = acc;
        if (i == 0) *counter = 0u;
    }
}
void launch_prep_queries(const float* xq_pad, int64_t ld, int64_t n, int d, int dpad, void* qh, int dh, uint32_t* flags,
                         float* qnorm, unsigned* counter, hipStream_t stream, const ClearList* clear) {
    if (n == 0) return;
    ClearList c{};
    if (clear) c = *clear;
    hipLaunchKernelGGL(prep_queries_kernel, dim3((unsigned)n), dim3(64), 0, stream, xq_pad, ld, n, d, dpad,
                       (_Float16*)qh, dh, flags, qnorm, counter, c);
    HIP_CHECK(hipGetLastError());
}
__global__ void __launch_bounds__(256) clear_words_kernel(ClearList clear) {
    for (int k = 0; k < clear.cnt; ++k)
        for (uint32_t w = blockIdx.x * 256u + threadIdx.x; w < clear.n[k]; w += gridDim.x * 256u) clear.p[k][w] = 0u;
}
void launch_clear_words(const ClearList& c, hipStream_t stream) {
    uint32_t mx = 0;
    for (int k = 0; k < c.cnt; ++k) mx = std::max(mx, c.n[k]);
    if (mx == 0) return;
    hipLaunchKernelGGL(clear_words_kernel, dim3(std::min<unsigned>((unsigned)div_up(mx, 256), 1024u)), dim3(256), 0, stream, c);
    HIP_CHECK(hipGetLastError());
}

// out[i] = -|y_i|^2 / 2 (L2) or 0 (IP) for i < n, -inf for the npad entries that follow: the value the
// filter kernel's accumulators START from (score = <q,y> - |y|^2/2; rows past the end can never qualify)
__global__ void half_norms_kernel(const float* __restrict__ xn, int64_t n, int npad, int metric,
                                  float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = metric == METRIC_L2 ? -0.5f * xn[i] : 0.f;
    else if (i < n + npad) out[i] = -INFINITY;
}
void launch_half_norms(const float* xn, int64_t n, int npad, int metric, float* out, hipStream_t stream) {
    if (n + npad == 0) return;
    hipLaunchKernelGGL(half_norms_kernel, dim3((unsigned)div_up(n + npad, 256)), dim3(256), 0, stream, xn, n, npad,
                       metric, out);
    HIP_CHECK(hipGetLastError());
}

// max of n non-negative floats into *out_bits (as uint bits)
__global__ void max_f32_kernel(const float* __restrict__ x, int64_t n, unsigned* __restrict__ out_bits) {
    float mx = 0.f;
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        if (!(v <= FLT_MAX)) bad = true;
        mx = fmaxf(mx, v);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    const bool anybad = __ballot(bad) != 0ull;
    if ((threadIdx.x & 63) == 0) atomicMax(out_bits, anybad ? 0x7f800000u : __float_as_uint(mx));
}
void launch_max_f32(const float* x, int64_t n, unsigned* out_bits, hipStream_t stream) {
    if (n == 0) return;
    unsigned grid = (unsigned)std::min<int64_t>(div_up(n, 256), 1024);
    hipLaunchKernelGGL(max_f32_kernel, dim3(grid), dim3(256), 0, stream, x, n, out_bits);
    HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------
// filter kernel
//
// Workgroup = 256 threads = 4 waves; wave w owns 64 queries (two 32-query MFMA column blocks whose
// fp16 coordinates stay in 64 VGPRs as B operands); the 4 waves share one stream of 64-row
// database tiles (64 x 128 halfs = 16 KB) brought in by global_load_lds_dwordx4 (no staging
// registers) through a 3-slot LDS ring.  The 16-byte chunks of a row are XOR-swizzled with the
// row number through the SOURCE address of the LDS-DMA (the LDS image is lane-linear), so the
// ds_read_b128 of 16 different rows at one k offset hits 16 different 16-byte bank groups.
// Per tile and wave: 2 x 2 blocks x 8 k-steps = 32 MFMAs (32 cycles each) against ~130 VALU
// instructions of epilogue.
//
// Split s owns the tiles s, s + nsplit, s + 2 nsplit, ... (striped, so that a run of similar
// consecutive rows is spread over every split), and the kernel runs in two modes that share the
// MFMA loop:
//   MODE_MAX      every lane keeps 8 running maxima per query (one per position class of the MFMA
//                 output fragment) over a 1/tstride sample of its split's tiles: 16 * nsplit
//                 "chunk maxima" per query.  The k-th largest of them is a lower bound of the k-th
//                 best score of the whole database (each maximum is a distinct row), and because
//                 rows are dealt to the chunks round-robin (period 64 * nsplit rows) only about
//                 k + k^2/(2S) rows of the sample beat it.  No selection, no memory traffic.
//   MODE_COLLECT  with the per-query threshold fixed (flat_tighten_kernel: k-th largest maximum
//                 minus the error band), rows above it are appended to the (query, split)
//                 segment: a few hundred per query in total.
//   MODE_DUMP     test hook: every approximate score to memory.
// ---------------------------------------------------------------------------------
constexpr int FQ_TR = 64;            // rows per tile
constexpr int FQ_KS = 128;           // halfs per k-slab
constexpr int FQ_TILE_BYTES = FQ_TR * FQ_KS * 2; // 16384
constexpr int MODE_MAX = 0, MODE_COLLECT = 1, MODE_DUMP = 2;

// Two geometries share the code (QB = 32-query MFMA column blocks per wave):
//   QB = 2: 4 waves x 64 queries, both 32-row blocks of a tile in flight (2 x 2 accumulators).
//           256 queries per workgroup, 2 workgroups per CU.  For small query batches.
//   QB = 4: 8 waves x 128 queries, the two 32-row blocks of a tile one after the other
//           (1 x 4 accumulators).  1024 queries per workgroup, 1 workgroup per CU: one 16 KB tile
//           feeds 8 x 64 MFMAs, i.e. 4 B/clk/CU of L2->LDS traffic at full MFMA rate, where the
//           QB = 2 geometry needs 32 B/clk/CU and is bound by the CU's ~10 B/clk fill path.
template <int QB>
struct FqGeom {
    static constexpr int WAVES = QB == 4 ? 8 : 4;
    static constexpr int THREADS = WAVES * 64;
    static constexpr int QPW = 32 * QB;          // queries per wave
    static constexpr int QPB = WAVES * QPW;      // queries per workgroup
    static constexpr int RBP = QB == 4 ? 1 : 2;  // 32-row blocks processed together
    static constexpr int NCL = QB == 4 ? 4 : 8;  // running maxima (position classes) per lane and query
    static constexpr int CPS = 2 * NCL;          // chunk maxima per (query, split)
    static constexpr int DMA_ROWS = 16 / WAVES;  // 1 KB row-chunk DMAs per wave and tile
    // steps (tiles) between two workgroup barriers.  Measured on the 8-wave geometry: 2 tiles per
    // barrier (6-slot ring) = 1 tile per barrier within noise, and so is a static s_setprio for one
    // half of the waves -- the barrier is not what holds the matrix pipe at ~30 % in the collect pass
    static constexpr int TPB = 1;
    static constexpr int RING = 3 * TPB;         // LDS ring slots: TPB computing, 2 * TPB in flight
    static constexpr int LDS_BIAS = RING * FQ_TILE_BYTES;
    static constexpr int LDS_GEN = LDS_BIAS + RING * FQ_TR * 4; // step at which all waves sift their slices
    static constexpr int LDS_CNT = LDS_GEN + 16;                // [queries per workgroup] append counters
    static constexpr int LDS_THR = LDS_CNT + QPB * 4;           // [queries per workgroup] collect thresholds
    // collect pass: a lane whose 16 scores of one (query, 32-row block) hold a candidate parks the
    // whole 16-score fragment in its wave's LDS slice; the slices are sifted and written to the
    // (query, split) segments outside the tile loop (a global store inside the loop would share
    // the vmcnt counter with the LDS-DMA prefetch, and per-score work inside it costs more VALU
    // time than the MFMAs take)
    static constexpr int WBLK = QB == 4 ? 128 : 96;              // parked fragments per wave
    static constexpr int LDS_BLKV = (LDS_THR + QPB * 4 + 15) & ~15; // [WAVES][WBLK][16] scores
    static constexpr int LDS_BLKM = LDS_BLKV + WAVES * WBLK * 64;   // [WAVES][WBLK] {first row, local query}
    static constexpr int LDS_TOTAL = LDS_BLKM + WAVES * WBLK * 8;
};

// LDS-DMA issued from inline asm: hipcc makes every ds_read that follows a
// __builtin_amdgcn_global_load_lds wait for vmcnt(0) (it cannot tell the two LDS regions apart),
// which would drain the prefetch before the tile in hand is even read.  Hidden in asm, the DMA
// is invisible to the compiler's counters; completion is enforced by our own counted
// s_waitcnt vmcnt(N) + barrier before the tile is consumed (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
            "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(gsrc), "s"(lds_dst)
            : "memory");
}
__device__ __forceinline__ void glds4(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
            "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(gsrc), "s"(lds_dst)
            : "memory");
}
// a wave-uniform pointer, forced into an SGPR pair (the register allocator may have computed it on the VALU)
__device__ __forceinline__ const char* uniform_ptr(const char* ptr) {
    const unsigned long long v = (unsigned long long)ptr;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const char*)(((unsigned long long)hi << 32) | lo);
}
// same, scalar 64-bit base + 32-bit per-lane byte offset (no address VGPR pair, no per-tile address math).
// The leading s_nop 4 covers the SALU-write -> VMEM-read hazard on the base SGPRs, which hipcc cannot
// see inside an asm statement (cdna_hip_programming.md 5.7, item 2).
__device__ __forceinline__ void glds16_s(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
            "s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(voff), "s"(sbase), "s"(lds_dst)
            : "memory");
}
__device__ __forceinline__ void glds4_s(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
            "s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(voff), "s"(sbase), "s"(lds_dst)
            : "memory");
}
// max(a, b, c) in one VALU instruction.  fmaxf() compiles to a canonicalising v_max_f32 x, x per operand
// on top of the maximum itself; the scores here are MFMA outputs (never signalling NaNs), and a quiet NaN
// operand is dropped by the hardware maximum like fmaxf drops it.
// The compiler's hazard recogniser does not look into asm statements, and the hardware does not interlock
// a VALU read of a register an MFMA is still writing: max3() consumers of accumulators sit behind
// mfma_results_ready(), and all of them are volatile so that they stay behind it.
__device__ __forceinline__ float max3(float a, float b, float c) {
    float r;
    asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// 20 wait states after the youngest MFMA that wrote one of the given accumulators (the ISA asks for 19
// between a 16-pass XDL write and a VALU read, 11 for 8 passes)
__device__ __forceinline__ void mfma_results_ready(const f32x16& a, const f32x16& b, const f32x16& c, const f32x16& d) {
    asm volatile("s_nop 15\n\ts_nop 3" ::"v"(a), "v"(b), "v"(c), "v"(d));
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}

// SINGLE: dh == 128, one k-slab per tile: the query operands never leave their registers and
// the loop holds no compiler-visible global load, whose counted s_waitcnt would otherwise also
// wait for the (younger, hidden) LDS-DMAs of the prefetch.
template <int METRIC, int MODE, bool SINGLE, int QB>
__global__ void __launch_bounds__(FqGeom<QB>::THREADS, 2) flat_filter_kernel(FlatFilterParams p) {
    using G = FqGeom<QB>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const int j = lane & 31;

    int split, grp;
    {
        const int b = blockIdx.x;
        if ((p.nsplit & 7) == 0) {
            // blocks b, b+8, b+16, ... run on one XCD: give them the same split (same rows, shared L2)
            const int xcd = b & 7;
            const int t = b >> 3;
            grp = t % p.ngroups;
            split = (t / p.ngroups) * 8 + xcd;
        } else {
            split = b % p.nsplit;
            grp = b / p.nsplit;
        }
    }
    // tiles of this split: split, split + nsplit, ...; the maxima pass visits every tstride-th of them
    const int total_tiles = (p.nb + FQ_TR - 1) / FQ_TR;
    const int nt_split = split < total_tiles ? (total_tiles - split + p.nsplit - 1) / p.nsplit : 0;
    const int tstep = MODE == MODE_MAX ? p.tstride : 1;
    const int ntiles = (nt_split + tstep - 1) / tstep;
    const int nslab = SINGLE ? 1 : p.dh / FQ_KS;
    const int nsteps = ntiles * nslab;
    auto tile_row0_of = [&](int tl) { return (split + tl * tstep * p.nsplit) * FQ_TR; };

    // ---- this lane's QB queries
    const int qbase = grp * G::QPB + wave * G::QPW; // wave-uniform
    const _Float16* qrow[QB];
    bool qvalid[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int q = qbase + qb * 32 + j;
        qvalid[qb] = q < p.nq;
        const int qc = qvalid[qb] ? q : p.nq - 1;
        qrow[qb] = p.xqh + (int64_t)qc * p.ldqh;
    }
    unsigned* lcnt = (unsigned*)(smem + G::LDS_CNT);
    // collect thresholds of this lane's queries: +inf for queries the filter cannot serve (flagged by
    // the tighten kernel) and for idle lanes
    float thr[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int q = qbase + qb * 32 + j;
        thr[qb] = (MODE == MODE_COLLECT && q < p.nq && !(p.dbg & 4)) ? p.thr[q] : INFINITY;
    }
    // every wave parks fragments in its own slice: the fill count is a wave-uniform register and
    // slots come from ballot prefix counts -- no LDS atomic round trip
    constexpr int WBLK = G::WBLK;
    float* lthr = (float*)(smem + G::LDS_THR);
    float* lblkv = (float*)(smem + G::LDS_BLKV) + wave * WBLK * 16;
    int2* lblkm = (int2*)(smem + G::LDS_BLKM) + wave * WBLK;
    int wcnt = 0;
    if (MODE == MODE_COLLECT) {
        for (int i = tid; i < G::QPB; i += G::THREADS) lcnt[i] = 0;
        if (tid == 0) *(unsigned*)(smem + G::LDS_GEN) = 0;
        if (h == 0) {
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) lthr[wave * G::QPW + qb * 32 + j] = thr[qb];
        }
    }

    float mx[QB][G::NCL];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int c = 0; c < G::NCL; ++c) mx[qb][c] = -INFINITY;

    if (nsteps == 0) {
        if (MODE == MODE_MAX) {
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                const int q = qbase + qb * 32 + j;
                if (qvalid[qb]) {
                    float* o = p.maxes + ((int64_t)q * p.nsplit + split) * G::CPS + h * G::NCL;
#pragma unroll
                    for (int c = 0; c < G::NCL; ++c) o[c] = -INFINITY;
                }
            }
        } else if (MODE == MODE_COLLECT) {
            for (int i = tid; i < G::QPB; i += G::THREADS) {
                const int q = grp * G::QPB + i;
                if (q < p.nq) p.res_cnt[(int64_t)q * p.nsplit + split] = 0;
            }
        }
        return;
    }

    // ---- LDS-DMA staging of step u (tile u / nslab, slab u % nslab) into a ring slot: 16 x 1 KB of
    // row chunks over the waves + (every wave, redundantly, so that all waves count the same number
    // of DMAs) the 64 per-row biases.  Addressing: a wave-uniform 64-bit base per tile (SGPRs) plus a
    // per-lane 32-bit offset that never changes (row-in-tile, swizzled chunk).  The fp16 rows and the
    // bias array are padded by one tile past the last row (the bias padding is +inf, which sends the
    // score of a row beyond the end to -inf), so no clamping and no masking anywhere.
    const unsigned lds_base = __builtin_amdgcn_readfirstlane(lds_addr(smem));
    constexpr int DMA_PER_STAGE = G::DMA_ROWS + 1;
    unsigned voff[G::DMA_ROWS];
#pragma unroll
    for (int i = 0; i < G::DMA_ROWS; ++i) {
        const int g = (wave * G::DMA_ROWS + i) * 64 + lane; // 16-byte chunk of the LDS image
        const int row = g >> 4, cpos = g & 15;
        const int c = cpos ^ (row & 15);                    // chunk of the source row that lands there
        voff[i] = (unsigned)(row * (int)p.ldbh * 2 + c * 16);
    }
    const unsigned voff_b = (unsigned)lane * 4u;
    auto stage = [&](int u_, int slot_) __attribute__((always_inline)) {
        // (wave-uniform by construction; spelled out because the asm operands below must be SGPRs)
        const int u = __builtin_amdgcn_readfirstlane(u_), slot = __builtin_amdgcn_readfirstlane(slot_);
        const int tl = u / nslab, sl = u - tl * nslab;
        const int row0 = __builtin_amdgcn_readfirstlane(tile_row0_of(tl));
        const char* sb = uniform_ptr((const char*)p.xbh + ((int64_t)row0 * p.ldbh + sl * FQ_KS) * 2);
#pragma unroll
        for (int i = 0; i < G::DMA_ROWS; ++i)
            glds16_s(sb, voff[i], lds_base + slot * FQ_TILE_BYTES + (wave * G::DMA_ROWS + i) * 1024);
        glds4_s(uniform_ptr((const char*)(p.xbhn + row0)), voff_b, lds_base + G::LDS_BIAS + slot * FQ_TR * 4);
    };

    half8 bq[QB][8];
    f32x16 acc[G::RBP][QB];

    auto load_b = [&](int sl) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int s = 0; s < 8; ++s) bq[qb][s] = *(const half8*)(qrow[qb] + sl * FQ_KS + s * 16 + h * 8);
    };

    // epilogue of the 32-row block(s) held in acc (they started from -|y|^2/2, so they ARE the scores):
    // rbase = first block index (0 or 1).  Lane (h, j) holds, for its query of column block qb, the
    // rows rb*32 + 8g + 4h + e at acc[..][qb][4g + e].
    auto epilogue = [&](int tl, int rbase) __attribute__((always_inline)) {
        const int tile_row0 = tile_row0_of(tl);
        if (MODE != MODE_DUMP) {
            if (QB == 4) mfma_results_ready(acc[0][0], acc[0][1], acc[0][QB - 2], acc[0][QB - 1]);
            else mfma_results_ready(acc[0][0], acc[0][1], acc[G::RBP - 1][0], acc[G::RBP - 1][1]);
        }
#pragma unroll
        for (int rp = 0; rp < G::RBP; ++rp) {
            const int rb = rbase + rp;
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                const f32x16& a = acc[rp][qb];
                if (MODE == MODE_MAX) {
                    // class = 4 consecutive rows of the tile (QB = 2: per 32-row block, 8 classes;
                    // QB = 4: the two blocks share 4 classes); fmaxf drops NaN scores
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int cl = QB == 4 ? g : rp * 4 + g;
                        mx[qb][cl] = max3(max3(mx[qb][cl], a[4 * g], a[4 * g + 1]), a[4 * g + 2], a[4 * g + 3]);
                        // opaque to the optimiser: otherwise it merges this block's maximum with the next
                        // block's into one max3 chain, which keeps two accumulator sets alive (spills)
                        if (QB == 4) asm volatile("" : "+v"(mx[qb][cl]));
                    }
                } else if (MODE == MODE_DUMP) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int grow = tile_row0 + rb * 32 + 8 * (i >> 2) + 4 * h + (i & 3);
                        const int q = qbase + qb * 32 + j;
                        if (qvalid[qb] && grow < p.nb) p.dump[(int64_t)q * p.nb + grow] = a[i];
                    }
                } else {
                    // 8 max3 + one compare per 16 scores; a wave-wide hit (a candidate among 32 rows x 32
                    // queries x 2) parks the lane's fragment as it is -- which of the 16 is sorted out later
                    const float th = thr[qb];
                    const float m = max3(max3(max3(a[0], a[1], a[2]), max3(a[3], a[4], a[5]), a[15]),
                                         max3(a[6], a[7], a[8]), max3(max3(a[9], a[10], a[11]), max3(a[12], a[13], a[14]), a[12]));
                    const u64 pm = __ballot(m > th);
                    if (__builtin_expect(pm != 0ull, 0)) {
                        // everything below hangs off operands made opaque here, so that none of the cold
                        // path's address arithmetic is speculated into the tile loop
                        int trow = tile_row0 + rb * 32 + 4 * h;
                        unsigned qlo = wave * G::QPW + qb * 32 + j;
                        asm volatile("" : "+v"(trow), "+v"(qlo));
                        if (m > th && !(p.dbg & 1)) {
                            const int pos = wcnt + __popcll(pm & ((1ull << lane) - 1ull));
                            if (pos < WBLK) {
                                f32x4* dst = (f32x4*)(lblkv + pos * 16);
#pragma unroll
                                for (int g = 0; g < 4; ++g)
                                    dst[g] = f32x4{a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]};
                                lblkm[pos] = int2{trow, (int)qlo};
                            } else {
                                // slice full before the loop could be left for a sift (tiles with far more
                                // candidates than expected): straight to the segment in HBM
                                const u64* rk = p.res_keys;
                                asm volatile("" : "+s"(rk));
                                u64* seg = const_cast<u64*>(rk) +
                                           ((int64_t)(grp * G::QPB + qlo) * p.nsplit + split) * p.cap;
#pragma unroll
                                for (int i = 0; i < 16; ++i) {
                                    if (a[i] > th) {
                                        const int grow = trow + 8 * (i >> 2) + (i & 3);
                                        const unsigned slot_ = atomicAdd(&lcnt[qlo], 1u);
                                        if (slot_ < (unsigned)p.cap)
                                            seg[slot_] = ((u64)score_key(a[i]) << 32) | (unsigned)grow;
                                    }
                                }
                            }
                        }
                        if (!(p.dbg & 1)) wcnt += __popcll(pm);
                    }
                }
            }
        }
    };

    // one step: the MFMAs of tile/slab u out of ring slot `slot`, and the epilogue behind its last slab
    // a wave none of whose queries exists (last query group of a batch) only stages its share of the tiles and
    // meets the barriers: the partner wave on its SIMD gets the matrix pipe to itself
    const bool wave_idle = qbase >= p.nq; // wave-uniform
    auto compute = [&](int u, int slot) __attribute__((always_inline)) {
        if (wave_idle) return;
        const int tl = u / nslab, sl = u - tl * nslab;
        if (!SINGLE && u > 0) load_b(sl);
        const char* tile = smem + slot * FQ_TILE_BYTES;
        const float* bias = (const float*)(smem + G::LDS_BIAS) + slot * FQ_TR;
        const int sw = j & 15;
        if (G::RBP == 2) {
            if (sl == 0) {
                // accumulators start from -|y|^2/2 of their rows (the bias travels with every slab's DMA)
#pragma unroll
                for (int rp = 0; rp < G::RBP; ++rp) {
                    f32x16 c0;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 b4 = *(const f32x4*)(bias + rp * 32 + 8 * g + 4 * h);
#pragma unroll
                        for (int e = 0; e < 4; ++e) c0[4 * g + e] = b4[e];
                    }
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) acc[rp][qb] = c0;
                }
            }
            const char* rowp0 = tile + j * 256;
            const char* rowp1 = tile + (32 + j) * 256;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int off = ((2 * s + h) ^ sw) << 4;
                const half8 a0 = *(const half8*)(rowp0 + off);
                const half8 a1 = *(const half8*)(rowp1 + off);
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    acc[0][qb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bq[qb][s], acc[0][qb], 0, 0, 0);
                    acc[G::RBP - 1][qb] =
                            __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, bq[qb][s], acc[G::RBP - 1][qb], 0, 0, 0);
                }
            }
            if (sl == nslab - 1) epilogue(tl, 0);
        } else {
            // one 32-row block at a time; with several k-slabs (d > 128) the two blocks of a tile
            // need both accumulator sets, so that geometry is SINGLE only (launch_flat_filter)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                // the first k-step takes -|y|^2/2 of the block's rows as its C operand
                f32x16 c0;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 b4 = *(const f32x4*)(bias + rb * 32 + 8 * g + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) c0[4 * g + e] = b4[e];
                }
                const char* rowp = tile + (rb * 32 + j) * 256;
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const int off = ((2 * s + h) ^ sw) << 4;
                    const half8 a0 = *(const half8*)(rowp + off);
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb)
                        acc[0][qb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bq[qb][s], s == 0 ? c0 : acc[0][qb],
                                                                            0, 0, 0);
                }
                // keep the epilogue of this block and the MFMAs of the next one apart: interleaved, the
                // scheduler holds two accumulator sets (128 VGPRs) and spills the query operands; the
                // other wave of the SIMD fills the matrix pipe meanwhile
                __builtin_amdgcn_sched_barrier(0);
                epilogue(tl, rb);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    load_b(0);
    // every compiler-visible load lands BEFORE the first hidden DMA is issued (vmcnt(0), the other
    // counters untouched): from here on the compiler's own scoreboard holds no pending load and it
    // inserts no counted vmcnt wait into the loop, which would also wait for the younger DMAs
    __builtin_amdgcn_s_waitcnt(0x0F70);
    // ring: group g (TPB steps) lives in slots (g % 3) * TPB ..; groups g+1 and g+2 are in flight while g computes
    constexpr int TPB = G::TPB;
#pragma unroll
    for (int t = 0; t < 2 * TPB; ++t)
        if (t < nsteps) stage(t, t);
    if (nsteps >= 2 * TPB) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_STAGE * TPB) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // Sift: this wave's parked fragments -> candidates in their (query, split) segments in HBM.  One lane
    // per fragment: repeat the kernel's own comparison on its 16 scores, take the segment slots with ONE
    // LDS atomic, write the hits.  Wave-private (own slice, own queries' counters).  It sits OUTSIDE the
    // tile loop (the loop is left and re-entered around it) so that none of its addresses is kept in
    // registers across the tiles, and all waves of the workgroup do it at the same step: a wave sifting on
    // its own would hold the other seven at the next barrier.
    auto flush = [&]() __attribute__((always_inline)) {
        const int nmine = (p.dbg & 2) ? 0 : min(wcnt, WBLK);
        for (int b = lane; b < nmine; b += 64) {
            const int2 mt = lblkm[b];
            const float th = lthr[mt.y];
            const f32x4* src = (const f32x4*)(lblkv + b * 16);
            unsigned mask = 0;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = src[g];
#pragma unroll
                for (int e = 0; e < 4; ++e) mask |= v[e] > th ? 1u << (4 * g + e) : 0u;
            }
            unsigned slot_ = atomicAdd(&lcnt[mt.y], (unsigned)__popc(mask));
            u64* seg = p.res_keys + ((int64_t)(grp * G::QPB + mt.y) * p.nsplit + split) * p.cap;
            while (mask) {
                const int i = __ffs(mask) - 1;
                mask &= mask - 1;
                const float v = lblkv[b * 16 + i];
                if (slot_ < (unsigned)p.cap)
                    seg[slot_] = ((u64)score_key(v) << 32) | (unsigned)(mt.x + 8 * (i >> 2) + (i & 3));
                ++slot_;
            }
        }
        wcnt = 0;
    };
    constexpr int WFLUSH = WBLK / 2;
    lds_volatile_u32* lgen = lds_volatile(smem + G::LDS_GEN); // (typed to LDS: a generic volatile store is a FLAT store + vmcnt(0), common.h)

    int gslot = 0; // (u / TPB) % 3
    int u = 0;
    while (u < nsteps) {
        bool sift;
        do {
            const int gslot2 = gslot >= 1 ? gslot - 1 : 2; // ring position of the group two ahead
#pragma unroll
            for (int t = 0; t < TPB; ++t)
                if (u + 2 * TPB + t < nsteps) stage(u + 2 * TPB + t, gslot2 * TPB + t);
#pragma unroll
            for (int t = 0; t < TPB; ++t)
                if (u + t < nsteps) compute(u + t, gslot * TPB + t);
            u += TPB;
            // a slice half full: every wave sifts after this step's barrier
            if (MODE == MODE_COLLECT && wcnt > WFLUSH) *lgen = (unsigned)u;
            // the next group must have landed (this wave's share) before anybody passes the barrier; a
            // complete group issued above stays in flight across it
            if (u + 2 * TPB <= nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_STAGE * TPB) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            gslot = gslot == 2 ? 0 : gslot + 1;
            sift = MODE == MODE_COLLECT && __builtin_amdgcn_readfirstlane(*lgen) == (unsigned)u;
        } while (u < nsteps && !sift);
        // (the sift's stores share the vector-memory counter with the DMAs the loop counts; loads complete
        // in order among themselves, so extra younger stores only make the counted waits conservative)
        if (MODE == MODE_COLLECT) flush();
    }

    if (MODE == MODE_MAX) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            const int q = qbase + qb * 32 + j;
            if (qvalid[qb]) {
                float* o = p.maxes + ((int64_t)q * p.nsplit + split) * G::CPS + h * G::NCL;
#pragma unroll
                for (int c = 0; c < G::NCL; c += 4)
                    *(f32x4*)(o + c) = f32x4{mx[qb][c], mx[qb][c + 1], mx[qb][c + 2], mx[qb][c + 3]};
            }
        }
    } else if (MODE == MODE_COLLECT) {
        // every wave flushed its slice when it left the loop
        __syncthreads();
        for (int i = tid; i < G::QPB; i += G::THREADS) {
            const int q = grp * G::QPB + i;
            if (q < p.nq) {
                const unsigned c = lcnt[i];
                if (c > (unsigned)p.cap) p.flags[q] = 1; // segment overflow: exact fallback for this query
                p.res_cnt[(int64_t)q * p.nsplit + split] = c > (unsigned)p.cap ? (unsigned)p.cap : c;
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// Small databases in ONE launch (round 6): the coarse quantizer of an IVF search -- 10 000 queries against nlist <= 4096
// centroids, k = nprobe -- through maxima pass, threshold, collect pass, exact re-rank and ordering inside one workgroup
// per 32 queries, instead of the five launches (maxima / tighten / collect / re-rank + the overflow count's round trip) the
// general path spends ~ 130 us on for 10 GFLOP (profiles/r5_ivfpq_1m_kernel_stats.csv).  Same scheme, same
// guarantees as above: approximate scores on v_mfma_f32_32x32x16_f16 over the fp16 rows, the per-query error band of
// flat_filter_err_bound, a SUPERSET of the answer re-ranked with the fmaf chain of flat_scan_kernel -- results bit-identical
// to the general path.  (Round 2's one-launch kernel lost because a 4-wave workgroup walked two serial sweeps of all rows
// with one block in flight; here the four waves split the rows and keep three 32-row blocks of loads ahead of the MFMAs, two
// workgroups per CU.)
//   * a wave owns the 32-row blocks w, w + 4, ...; B operands = the workgroup's 32 queries (8 k-steps in registers);
//   * scores = MFMA chain from zero + (-|y|^2 / 2) (one more rounding than starting the chain there: inside the band,
//     which budgets d roundings of that magnitude);
//   * pass 1: maximum per (query, 16-row half block) -> LDS; threshold = band_threshold(k-th largest maximum, e_q) (the k
//     rows that realise the k largest maxima are distinct, so the k-th best score is at least that);
//   * pass 2: rows at or above the threshold -> a list per LANE (no atomics: 16 returning LDS atomics per block serialised pass 2
//     in the first version), merged into the query's candidate list (u16 row numbers, FS_CAP per query) afterwards;
//   * re-rank: one thread per (query, candidate), the chain of flat_rerank_kernel; ordering by counting.
// Queries flagged by prep_queries (fp16 range, NaN) or with more than FS_CAP candidates go to the exact scan (ovf_list).
// ---------------------------------------------------------------------------------
constexpr int FS_THREADS = 256, FS_WAVES = FS_THREADS / 64, FS_Q = 32, FS_CAP = 128, FS_MAXCH = 256, FS_SUB = 16;
struct FsShared {
    float cmax[FS_Q][FS_MAXCH]; // pass 1: chunk maxima; later: the candidates' exact keys (u64 [FS_Q][FS_CAP])
    // pass 2: what a LANE found (query j, half h of wave w: list 2 w + h), no atomics -- a record per 16-row half block that holds a
    // candidate: block << 16 | one bit per score position (round 6, second version: a compare + add-with-carry per position instead
    // of a branch, a row number and a store per position: pass 2 61 000 -> 47 000 cycles per workgroup)
    uint32_t sub[FS_Q][2 * FS_WAVES][FS_SUB]; // (dead behind the merge: the row staging of the re-rank, FS_STAGE bytes per wave)
    float qs[FS_Q][128];                      // fp32 queries (re-rank)
    uint16_t cand[FS_Q][FS_CAP];              // the query's candidates, compacted
    float thr[FS_Q], xn[FS_Q];
    unsigned cnt8[FS_Q][2 * FS_WAVES];
    unsigned cnt[FS_Q], pre[FS_Q + 1];
    unsigned bad[FS_Q];
};
static_assert(sizeof(float) * FS_MAXCH == sizeof(u64) * FS_CAP, "the exact keys reuse the maxima");
constexpr int FS_STAGE = 64 * 64; // a wave's staging area of the re-rank: 16 coordinates of 64 rows
static_assert(FS_WAVES * FS_STAGE <= (int)(sizeof(uint32_t) * FS_Q * 2 * FS_WAVES * FS_SUB), "the staging areas overlay the records of pass 2");

// TOPSEL (k <= 32; round 6, second version): pass 1 keeps every lane's EIGHT best chunk maxima in registers (a sorted insertion per
// block: 16 VALU instructions under the loads) and the threshold is the k-th best of the 64 a query's eight lanes hold -- a lower
// bound of the k-th best of all its maxima (every listed value is one of them), within 0.3 % of its rank on random data (a lane
// holds more than eight of the k <= 32 best of 256 once in a few hundred queries).  The bisection then counts 8 keys per lane
// instead of 32: `select` 30 000 -> 6 000 cycles of 118 000 per workgroup (FS_TIMING).
template <int METRIC, bool TOPSEL>
__global__ void __launch_bounds__(FS_THREADS, 2) flat_small_fused_kernel(FlatSmallParams p) {
    __shared__ FsShared sh;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31;
    const int q0 = blockIdx.x * FS_Q;
    const int nblk = (p.nb + 31) >> 5, nch = 2 * nblk;
    const int qj = min(q0 + j, p.nq - 1);
    // ---- queries: B operands, fp32 copies, thresholds' ingredients
    half8 bq[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) bq[s] = *(const half8*)(p.xqh + (int64_t)qj * p.ldqh + 16 * s + 8 * h);
    for (int i = tid; i < FS_Q * 32; i += FS_THREADS) { // (four floats per thread and step)
        const int qq = i >> 5, c = (i & 31) * 4;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (q0 + qq < p.nq && c < p.dpad) v = *(const f32x4*)(p.xq + (int64_t)(q0 + qq) * p.ldq + c);
        *(f32x4*)&sh.qs[qq][c] = v;
    }
    if (tid < FS_Q) {
        const int q = q0 + tid;
        sh.bad[tid] = (q >= p.nq) ? 2u : (p.flags[q] ? 1u : 0u);
        sh.xn[tid] = q < p.nq ? p.xqn[q] : 0.f;
    }
    // one pass over this wave's blocks; MODE_MAX writes the chunk maxima, MODE_COLLECT the lane's candidate rows
    auto sweep = [&](auto mode_c) __attribute__((always_inline)) {
        constexpr int MODE = decltype(mode_c)::value;
        const float th = MODE == MODE_COLLECT ? sh.thr[j] : 0.f;
        uint32_t* mysub = sh.sub[j][2 * wave + h];
        unsigned nsub = 0, ncand = 0; // records, candidates of this lane
        // (operand-major copy of the rows, flat_operand_major_kernel: k-step s of block b = one contiguous KB, lane l its 16 bytes.
        // From the row-major copy every load instruction touched 32 cache lines -- the L1's request rate, not bytes, paced the
        // first version: 0.21 ms for the launch)
        // (UNCONDITIONAL: behind the last block the loads re-read it -- a branch around a prefetch makes hipcc wait vmcnt(0) at the
        // join, i.e. for the prefetch it has just issued: 1700 cycles per block in the first version)
        // (the block's row terms travel with it: loaded at the head of `block` they were a 500-cycle L2 round trip per block that
        // eight MFMAs could not cover)
        auto loadblk = [&](int b, half8 (&a)[8], f32x4 (&bias)[4]) __attribute__((always_inline)) {
            const int bc = min(b, nblk - 1);
            const half8* r = (const half8*)p.xbo + (int64_t)bc * 512 + lane;
#pragma unroll
            for (int s = 0; s < 8; ++s) a[s] = r[64 * s];
#pragma unroll
            for (int g = 0; g < 4; ++g) bias[g] = *(const f32x4*)(p.xbhn + 32 * bc + 8 * g + 4 * h);
        };
        // three blocks of loads ahead of the MFMAs in three named buffers: the loop is unrolled by three so that no buffer is ever
        // COPIED (a rotating copy waits for the newest load before it moves it: one block of look-ahead instead of three --
        // 1700 cycles per block in the first version, an L2 round trip each)
        half8 a0[8], a1[8], a2[8];
        f32x4 bias0[4], bias1[4], bias2[4];
        float top[8]; // TOPSEL, pass 1: this lane's best chunk maxima, descending
#pragma unroll
        for (int i = 0; i < 8; ++i) top[i] = -INFINITY;
        auto block = [&](int b, half8 (&a)[8], f32x4 (&bias)[4]) __attribute__((always_inline)) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s], bq[s], acc, 0, 0, 0);
            float sc[16], m = -INFINITY;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // (rows behind nb: their bias is the -inf padding of xbhn, 64 rows of it -- no test)
                    const float v = acc[4 * g + e] + bias[g][e];
                    sc[4 * g + e] = v;
                    m = fmaxf(m, v); // (drops a NaN score like the general path's v_max3)
                }
            loadblk(b + 3 * FS_WAVES, a, bias);
            if (MODE == MODE_MAX) {
                if constexpr (TOPSEL) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float hi = fmaxf(top[i], m);
                        m = fminf(top[i], m);
                        top[i] = hi;
                    }
                } else {
                    sh.cmax[j][2 * b + h] = m;
                }
            } else if (m >= th) {
                unsigned mask = 0;
#pragma unroll
                for (int i = 15; i >= 0; --i) mask = mask + mask + (sc[i] >= th ? 1u : 0u); // (bit i = score position i)
                if (nsub < (unsigned)FS_SUB) mysub[nsub] = ((unsigned)b << 16) | mask;
                ++nsub;
                ncand += (unsigned)__popc(mask);
            }
        };
        int b = wave;
        loadblk(b, a0, bias0);
        loadblk(b + FS_WAVES, a1, bias1);
        loadblk(b + 2 * FS_WAVES, a2, bias2);
        for (; b < nblk; b += 3 * FS_WAVES) {
            block(b, a0, bias0);
            if (b + FS_WAVES < nblk) block(b + FS_WAVES, a1, bias1);
            if (b + 2 * FS_WAVES < nblk) block(b + 2 * FS_WAVES, a2, bias2);
        }
        // (a lane that ran out of records reports more candidates than a query may have: the query goes to the exact scan)
        if (MODE == MODE_COLLECT) sh.cnt8[j][2 * wave + h] = nsub > (unsigned)FS_SUB ? (unsigned)FS_CAP + 1u : ncand;
        if (MODE == MODE_MAX && TOPSEL) {
#pragma unroll
            for (int i = 0; i < 8; ++i) sh.cmax[j][8 * (2 * wave + h) + i] = top[i];
        }
    };
#ifdef FS_TIMING
    unsigned long long tm[8];
    int tmi = 0;
#define FS_MARK() tm[tmi++] = __builtin_readcyclecounter()
#else
#define FS_MARK()
#endif
    FS_MARK();
    sweep(std::integral_constant<int, MODE_MAX>{});
    __syncthreads();
    FS_MARK();
    // ---- thresholds: wave w serves queries 8 w .. 8 w + 7, EIGHT LANES per query (32 maxima each); k-th smallest score key of the
    // nch maxima by bisection on the bits, all eight queries of the wave at once (the first version bisected one query at a
    // time through ballots: 256 dependent VALU -> SALU round trips per wave, 23 us)
    {
        const int qq = (FS_Q / FS_WAVES) * wave + (lane >> 3), sl = lane & 7;
        constexpr int NU = TOPSEL ? 8 : FS_MAXCH / 8; // keys per lane
        const int nkey = TOPSEL ? 64 : nch;
        uint32_t key[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) key[u] = 8 * u + sl < nkey ? score_key(sh.cmax[qq][8 * u + sl]) : 0xffffffffu;
        uint32_t pre = 0;
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t cand = pre | ((1u << bit) - 1u); // (keys <= cand: this bit clear under the prefix found so far)
            // keys <= cand, counted through the borrow of cand - key (a v_cmp -> v_cndmask pair per key costs two wait states on the
            // condition register each: 27 000 cycles for the 32 x 32 compares of a lane in the first version)
            int c = NU;
#pragma unroll
            for (int u = 0; u < NU; ++u) c += (int)(((unsigned long long)cand - (unsigned long long)key[u]) >> 32);
            // sum over the eight lanes of the query: quad swaps, then the mirrored lane of the other quad
            c += __builtin_amdgcn_update_dpp(0, c, 0xB1, 0xf, 0xf, false);  // quad_perm [1, 0, 3, 2]
            c += __builtin_amdgcn_update_dpp(0, c, 0x4E, 0xf, 0xf, false);  // quad_perm [2, 3, 0, 1]
            c += __builtin_amdgcn_update_dpp(0, c, 0x141, 0xf, 0xf, false); // row_half_mirror
            if (c < p.k) pre |= 1u << bit;
        }
        if (sl == 0) {
            const float e = flat_filter_err_bound(METRIC, p.d, sh.xn[qq], p.yn_max, false);
            const float tk = key_score(pre);
            sh.thr[qq] = (sh.bad[qq] || !(e < FLT_MAX)) ? INFINITY : (tk > -INFINITY ? band_threshold(tk, e) : -INFINITY);
            if (!(e < FLT_MAX) && sh.bad[qq] == 0) sh.bad[qq] = 1u;
        }
    }
    __syncthreads();
    FS_MARK();
    sweep(std::integral_constant<int, MODE_COLLECT>{});
    __syncthreads();
    FS_MARK();
    // ---- the eight lane lists of a query -> one list (thread = (query, list)); a list or a query that overflowed: exact scan
    {
        const int qq = tid >> 3, sl = tid & 7;
        unsigned off = 0, tot = 0;
        bool over = false;
#pragma unroll
        for (int u = 0; u < 2 * FS_WAVES; ++u) {
            const unsigned c = sh.cnt8[qq][u];
            if (u < sl) off += c;
            tot += c;
        }
        over = over || tot > (unsigned)FS_CAP;
        if (!over && sh.bad[qq] == 0) {
            const unsigned c = sh.cnt8[qq][sl];
            for (unsigned i = 0, r = 0; i < c; ++r) {
                const uint32_t rec = sh.sub[qq][sl][r];
                for (uint32_t m = rec & 0xffffu; m; m &= m - 1u) {
                    const int pos = __builtin_ctz(m);
                    sh.cand[qq][off + i++] = (uint16_t)(32 * (rec >> 16) + 8 * (pos >> 2) + 4 * (sl & 1) + (pos & 3));
                }
            }
        }
        if (sl == 0) {
            if (over && sh.bad[qq] == 0) sh.bad[qq] = 1u;
            sh.cnt[qq] = sh.bad[qq] ? 0u : tot;
        }
    }
    __syncthreads();
    if (tid == 0) {
        unsigned acc = 0;
        for (int qq = 0; qq < FS_Q; ++qq) {
            sh.pre[qq] = acc;
            acc += sh.cnt[qq];
        }
        sh.pre[FS_Q] = acc;
    }
    __syncthreads();
    if (p.bad_out) {
        if (tid < FS_Q && q0 + tid < p.nq) p.bad_out[q0 + tid] = sh.bad[tid] == 1u ? 1u : 0u;
    } else if (tid < FS_Q && sh.bad[tid] == 1u) {
        p.ovf_list[atomicAdd(p.ovf_cnt, 1u)] = (uint32_t)(q0 + tid);
    }
    FS_MARK();
    // ---- exact distances of the candidates: (query, candidate) pair g -> thread g % 256 in round g / 256
    const unsigned total = sh.pre[FS_Q];
    u64* ekey = (u64*)&sh.cmax[0][0]; // [FS_Q][FS_CAP] (the maxima are dead since the thresholds were taken)
    const int niter = (int)((total + FS_THREADS - 1) / FS_THREADS);
    auto finish = [&](int qq, unsigned row, float acc) __attribute__((always_inline)) -> u64 {
        float dis;
        if (METRIC == METRIC_L2) {
            dis = __fmaf_rn(-2.f, acc, sh.xn[qq] + p.xbn[row]);
            dis = dis < 0.f ? 0.f : dis;
        } else {
            dis = acc;
        }
        return ((u64)ordkey<METRIC>(dis) << 32) | row;
    };
    // the chain of flat_scan_kernel / flat_rerank_kernel: 8-float steps, e and 4 + e interleaved
    auto step = [&](float& acc, const float* qs, const f32x4& y0, const f32x4& y1, int s) __attribute__((always_inline)) {
        const f32x4 x0 = *(const f32x4*)(qs + s), x1 = *(const f32x4*)(qs + s + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc = __fmaf_rn(y0[e], x0[e], acc);
            acc = __fmaf_rn(y1[e], x1[e], acc);
        }
    };
    {
        int qq = 0; // (the pairs of a thread come in query order: the search for a pair's query resumes where the last one ended)
        if (p.dpad == 128) {
            // Rows through LDS (round 6, second version).  A thread owns a pair and needs the candidate's 512-byte row in chain order;
            // read by the thread itself, every load instruction of a wave touched 64 cache lines (the L1's tag rate: 50 000 cycles
            // per workgroup for 1200 rows).  Now FOUR LANES fetch 64 bytes of a row together -- 16 lines per instruction --, the pieces
            // pass through the wave's 4 KB staging area (the records of pass 2, dead since the merge), and every lane reads the 16
            // coordinates of its own row back: eight stages per pair.  Staged piece c of row r sits at chunk c ^ ((r >> 2) & 3) of the
            // row's 64 bytes: sixteen lanes' 16-byte reads (rows l .. l + 15, one chunk index) fall into sixteen different bank groups.
            char* stg = (char*)&sh.sub[0][0][0] + wave * FS_STAGE;
            for (int m_ = 0; m_ < niter; ++m_) {
                const unsigned g = (unsigned)m_ * FS_THREADS + (unsigned)tid;
                const bool valid = g < total;
                if (valid)
                    while (sh.pre[qq + 1] <= g) ++qq;
                const int pos = valid ? (int)(g - sh.pre[qq]) : 0;
                const unsigned row = valid ? sh.cand[qq][pos] : 0u;
                // the rows this lane helps to fetch: slot 16 i + (lane >> 2), i = 0 .. 3; its chunk of their 64 bytes: lane & 3
                const float* src[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) src[i] = p.xb + (int64_t)__shfl((int)row, 16 * i + (lane >> 2)) * p.ldb + 4 * (lane & 3);
                // (a ring of four stages: sixteen loads in flight, the slot of a stage refilled as soon as it has been staged)
                f32x4 v[4][4];
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[t][i] = *(const f32x4*)(src[i] + 16 * t);
                const float* qs = sh.qs[qq];
                float acc = 0.f;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_wave_barrier(); // (the reads of the stage before were issued: LDS keeps a wave's order)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = 16 * i + (lane >> 2);
                        *(f32x4*)(stg + r * 64 + 16 * ((lane & 3) ^ ((r >> 2) & 3))) = v[t & 3][i];
                        if (t + 4 < 8) v[t & 3][i] = *(const f32x4*)(src[i] + 16 * (t + 4));
                    }
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                    f32x4 y[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) y[u] = *(const f32x4*)(stg + lane * 64 + 16 * (u ^ ((lane >> 2) & 3)));
                    step(acc, qs, y[0], y[1], 16 * t);
                    step(acc, qs, y[2], y[3], 16 * t + 8);
                }
                if (valid) ekey[qq * FS_CAP + pos] = finish(qq, row, acc);
            }
        } else {
            for (int m_ = 0; m_ < niter; ++m_) {
                const unsigned g = (unsigned)m_ * FS_THREADS + (unsigned)tid;
                if (g >= total) break;
                while (sh.pre[qq + 1] <= g) ++qq;
                const int pos = (int)(g - sh.pre[qq]);
                const unsigned row = sh.cand[qq][pos];
                const float* yr = p.xb + (int64_t)row * p.ldb;
                const float* qs = sh.qs[qq];
                float acc = 0.f;
                int s = 0;
                for (; s + 32 <= p.dpad; s += 32) {
                    f32x4 y[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) y[u] = *(const f32x4*)(yr + s + 4 * u);
#pragma unroll
                    for (int u = 0; u < 4; ++u) step(acc, qs, y[2 * u], y[2 * u + 1], s + 8 * u);
                }
                for (; s < p.dpad; s += 8) step(acc, qs, *(const f32x4*)(yr + s), *(const f32x4*)(yr + s + 4), s);
                ekey[qq * FS_CAP + pos] = finish(qq, row, acc);
            }
        }
    }
    FS_MARK();
    __syncthreads();
    // ---- exact top-k under (distance, id): every key straight to its rank
    const float pad = neutral_distance(METRIC);
    {
        int qq = 0;
        for (int m_ = 0; m_ < niter; ++m_) {
            const unsigned g = (unsigned)m_ * FS_THREADS + (unsigned)tid;
            if (g >= total) break;
            while (sh.pre[qq + 1] <= g) ++qq;
            const int n = (int)sh.cnt[qq];
            const u64 x = ekey[qq * FS_CAP + (int)(g - sh.pre[qq])];
            const ulonglong2* kk = (const ulonglong2*)(ekey + qq * FS_CAP);
            int r = 0;
            for (int i = 0; i < n; i += 2) {
                const ulonglong2 k2 = kk[i >> 1];
                r += (k2.x < x ? 1 : 0) + ((i + 1 < n && k2.y < x) ? 1 : 0);
            }
            if (r < p.k) {
                const uint32_t wk = (uint32_t)(x >> 32);
                const bool ok = wk < kInvalidOrdKey;
                p.out_dis[(int64_t)(q0 + qq) * p.k + r] = ok ? unordkey<METRIC>(wk) : pad;
                p.out_ids[(int64_t)(q0 + qq) * p.k + r] = ok ? (int64_t)(uint32_t)x : -1;
            }
        }
    }
    // (fewer candidates than k: only when the database holds fewer rows)
    for (int i = tid; i < FS_Q * p.k; i += FS_THREADS) {
        const int qq = i / p.k, r = i - qq * p.k;
        // (deferred overflow: a handed-back query names no row at all until it is redone)
        if (q0 + qq < p.nq && ((sh.bad[qq] == 0 && r >= (int)sh.cnt[qq]) || (p.bad_out && sh.bad[qq] == 1u))) {
            p.out_dis[(int64_t)(q0 + qq) * p.k + r] = pad;
            p.out_ids[(int64_t)(q0 + qq) * p.k + r] = -1;
        }
    }
    FS_MARK();
#ifdef FS_TIMING
    if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == 200))
        printf("fs timing block %d: pass1 %llu select %llu pass2 %llu merge %llu rerank %llu rank+out %llu (cycles) total cand %u\n", (int)blockIdx.x,
               tm[1] - tm[0], tm[2] - tm[1], tm[3] - tm[2], tm[4] - tm[3], tm[5] - tm[4], tm[6] - tm[5], total);
#endif
}

// operand-major fp16 copy of a small database for flat_small_fused_kernel: out[(b * 8 + s) * 64 + l] (16 bytes) = coordinates
// 16 s + 8 (l >> 5) .. + 7 of row 32 b + (l & 31) -- what lane l feeds the MFMA of k-step s (rows behind nb: zeros)
__global__ void flat_operand_major_kernel(const _Float16* __restrict__ xbh, int64_t ldbh, int nb, half8* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x; // (b * 8 + s) * 64 + l
    const int nblk = (nb + 31) >> 5;
    if (i >= nblk * 512) return;
    const int l = i & 63, s = (i >> 6) & 7, b = i >> 9;
    const int row = 32 * b + (l & 31);
    half8 v = half8{0, 0, 0, 0, 0, 0, 0, 0};
    if (row < nb) v = *(const half8*)(xbh + (int64_t)row * ldbh + 16 * s + 8 * (l >> 5));
    out[i] = v;
}
void launch_flat_operand_major(const void* xbh, int64_t ldbh, int nb, void* out, hipStream_t stream) {
    if (nb == 0) return;
    const int n = ((nb + 31) >> 5) * 512;
    hipLaunchKernelGGL(flat_operand_major_kernel, dim3((unsigned)div_up(n, 256)), dim3(256), 0, stream, (const _Float16*)xbh, ldbh, nb,
                       (half8*)out);
    HIP_CHECK(hipGetLastError());
}
bool flat_small_fused_supported(int metric, int nb, int d, int dh, int k) {
    const int nch = 2 * ((nb + 31) / 32);
    return (metric == METRIC_L2 || metric == METRIC_INNER_PRODUCT) && dh == 128 && d <= 128 && nb <= 32 * (FS_MAXCH / 2) &&
           k <= 64 && nch >= 4 * k && nb >= k;
}
void launch_flat_small_fused(const FlatSmallParams& p, hipStream_t stream) {
    if (p.nq == 0) return;
    FA_THROW_IF_NOT(flat_small_fused_supported(p.metric, p.nb, p.d, (int)p.ldbh, p.k) && p.dpad <= 128 && p.dpad % 8 == 0 && p.xb && p.xbo);
    const dim3 grid((unsigned)div_up(p.nq, FS_Q)), block(FS_THREADS);
    if (p.k <= 32) {
        if (p.metric == METRIC_L2) hipLaunchKernelGGL((flat_small_fused_kernel<METRIC_L2, true>), grid, block, 0, stream, p);
        else hipLaunchKernelGGL((flat_small_fused_kernel<METRIC_INNER_PRODUCT, true>), grid, block, 0, stream, p);
    } else {
        if (p.metric == METRIC_L2) hipLaunchKernelGGL((flat_small_fused_kernel<METRIC_L2, false>), grid, block, 0, stream, p);
        else hipLaunchKernelGGL((flat_small_fused_kernel<METRIC_INNER_PRODUCT, false>), grid, block, 0, stream, p);
    }
    HIP_CHECK(hipGetLastError());
}

int flat_filter_queries_per_block(int geom) {
    return geom == 2 ? FqGeom<4>::QPB : FqGeom<2>::QPB;
}
int flat_filter_chunks_per_split(int geom) {
    return geom == 2 ? FqGeom<4>::CPS : FqGeom<2>::CPS;
}

template <int METRIC, int MODE>
static void launch_filter_mode(const FlatFilterParams& p, hipStream_t stream) {
    dim3 grid((unsigned)(p.nsplit * p.ngroups));
    if (p.geom == 2) {
        using G = FqGeom<4>;
        hipLaunchKernelGGL((flat_filter_kernel<METRIC, MODE, true, 4>), grid, dim3(G::THREADS), G::LDS_TOTAL, stream, p);
    } else {
        using G = FqGeom<2>;
        if (p.dh == FQ_KS)
            hipLaunchKernelGGL((flat_filter_kernel<METRIC, MODE, true, 2>), grid, dim3(G::THREADS), G::LDS_TOTAL, stream, p);
        else
            hipLaunchKernelGGL((flat_filter_kernel<METRIC, MODE, false, 2>), grid, dim3(G::THREADS), G::LDS_TOTAL, stream, p);
    }
}

void launch_flat_filter(const FlatFilterParams& p_, int mode, hipStream_t stream) {
    if (p_.nq == 0 || p_.nb == 0) return;
    FlatFilterParams p = p_;
    if (const char* e = experiment_env("FAISS_AMD_FILTER_DBG")) p.dbg = atoi(e);
    FA_THROW_IF_NOT(p.dh % FQ_KS == 0 && p.ldqh % 8 == 0 && p.ldbh % 8 == 0);
    FA_THROW_IF_NOT(p.tstride >= 1 && p.nsplit >= 1);
    FA_THROW_IF_NOT_MSG(p.geom == 0 || (p.geom == 2 && p.dh == FQ_KS), "the 8-wave geometry needs dh == 128");
    FA_THROW_IF_NOT(p.cps == flat_filter_chunks_per_split(p.geom));
#define FA_LAUNCH_M(M)                                                            \
    do {                                                                          \
        if (mode == MODE_MAX) launch_filter_mode<M, MODE_MAX>(p, stream);         \
        else if (mode == MODE_COLLECT) launch_filter_mode<M, MODE_COLLECT>(p, stream); \
        else launch_filter_mode<M, MODE_DUMP>(p, stream);                         \
    } while (0)
    if (p.metric == METRIC_L2) FA_LAUNCH_M(METRIC_L2);
    else FA_LAUNCH_M(METRIC_INNER_PRODUCT);
#undef FA_LAUNCH_M
    HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------
// tighten kernel: one workgroup per query turns the S = cps * nsplit chunk maxima into the
// collect threshold  thr = (k-th largest maximum) - 2 e_q  (band_threshold)
// ---------------------------------------------------------------------------------
constexpr int TG_THREADS = 256;

template <int METRIC>
__global__ void __launch_bounds__(TG_THREADS) flat_tighten_kernel(FlatFilterParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int S = p.nsplit * p.cps;
    u64* keys = (u64*)smem;                     // [S]
    unsigned* hist = (unsigned*)(keys + S);     // [256]
    WgSelCtl* ctl = (WgSelCtl*)(hist + 256);
    const int q = blockIdx.x;
    const int tid = threadIdx.x;
    const float e = flat_filter_err_bound(METRIC, p.d, p.xqn[q], p.yn_max, p.exact_inputs != 0);
    if (p.flags[q] || !(e < FLT_MAX)) {
        // fp16 range overflow / NaN in the query: not served by the filter
        if (tid == 0) {
            p.flags[q] = 1;
            p.thr[q] = INFINITY;
        }
        return;
    }
    const float* mxs = p.maxes + (int64_t)q * S;
    for (int i = tid; i < S; i += TG_THREADS) keys[i] = ((u64)score_key(mxs[i]) << 32) | (unsigned)i;
    __syncthreads();
    float tk;
    if (S > p.k) {
        const u64 kth = wg_select_kth<TG_THREADS>(keys, S, p.k, hist, ctl);
        tk = key_score((uint32_t)(kth >> 32));
    } else {
        tk = -INFINITY; // fewer chunks than k: no pruning possible
    }
    if (tid == 0) p.thr[q] = tk > -INFINITY ? band_threshold(tk, e) : -INFINITY;
}

void launch_flat_tighten(const FlatFilterParams& p, hipStream_t stream) {
    if (p.nq == 0) return;
    const size_t lds = (size_t)p.nsplit * p.cps * 8 + 1024 + 64;
    FA_THROW_IF_NOT(lds <= 64 * 1024);
    if (p.metric == METRIC_L2)
        hipLaunchKernelGGL((flat_tighten_kernel<METRIC_L2>), dim3((unsigned)p.nq), dim3(TG_THREADS), lds, stream, p);
    else
        hipLaunchKernelGGL((flat_tighten_kernel<METRIC_INNER_PRODUCT>), dim3((unsigned)p.nq), dim3(TG_THREADS), lds,
                           stream, p);
    HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------
// re-rank kernel: one workgroup per query
// ---------------------------------------------------------------------------------
constexpr int RR_THREADS = 256;
constexpr int RR_GATHER = 4096; // upper limit of FlatRerankParams::gcap (approximate candidates gathered into LDS)
constexpr int RR_CAND = 2048;   // rows inside the error band that are re-ranked exactly
constexpr int RR_RANK_MAX = 512; // final stage: up to this many keys go straight to their rank (rank_of) instead of select + sort

struct RrShared {
    unsigned hist[256];
    WgSelCtl ctl;
    unsigned total;
};

// THREADS: 256, or 64 for small k (the coarse quantizer of an IVF search: k = nprobe, ~ 60 candidates per query -- a 256-thread
// workgroup keeps three of its four waves idle through every dependent load while it occupies a CU's wave slots: a wavefront per
// query doubles the queries in flight)
template <int METRIC, int THREADS>
__global__ void __launch_bounds__(THREADS) flat_rerank_kernel(FlatRerankParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    RrShared* sh = (RrShared*)smem;
    u64* cand = (u64*)(smem + ((sizeof(RrShared) + 15) & ~(size_t)15)); // [gcap]
    float* qs = (float*)(cand + p.gcap);                                  // [dpad]
    int64_t* w_id = (int64_t*)(qs + p.dpad + (p.dpad & 1));               // [kp] (8-byte aligned)
    unsigned* w_key = (unsigned*)(w_id + p.kp);                           // [kp]
    const int q = blockIdx.x;
    const int tid = threadIdx.x;

    if (p.flags[q]) {
        // the filter could not guarantee a superset for this query: hand it to the exact path
        if (tid == 0) p.ovf_list[atomicAdd(p.ovf_cnt, 1u)] = (uint32_t)q;
        return;
    }
    if (tid == 0) sh->total = 0;
    for (int c = tid; c < p.dpad; c += THREADS) qs[c] = p.xq[(int64_t)q * p.ldq + c];
    __syncthreads();
    // ---- gather the (query, split) segments into LDS: one thread per segment, so the nsplit
    // dependent (count -> keys) global loads run side by side instead of one after the other
    for (int s = tid; s < p.nsplit; s += THREADS) {
        const unsigned cnt = p.res_cnt[(int64_t)q * p.nsplit + s];
        if (cnt) {
            const unsigned base = atomicAdd(&sh->total, cnt);
            const u64* seg = p.res_keys + ((int64_t)q * p.nsplit + s) * p.cap;
            // four keys per round trip (a segment holds ~3 on average; one load per iteration would serialise
            // their latencies).  The slots behind cnt belong to the segment (cap >= 32, a multiple of 4).
            for (unsigned i = 0; i < cnt; i += 4) {
                u64 kv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) kv[u] = seg[i + u];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (i + u < cnt && base + i + u < (unsigned)p.gcap) cand[base + i + u] = kv[u];
            }
        }
    }
    __syncthreads();
    int n = (int)sh->total;
    if (n > p.gcap) {
        if (tid == 0) p.ovf_list[atomicAdd(p.ovf_cnt, 1u)] = (uint32_t)q;
        return;
    }

    // rank of key x among cand[0, n) = number of smaller keys (keys are unique); two keys per LDS read, every lane the
    // same address (broadcast).  For the few hundred keys a query has here this beats the radix select (up to eight
    // histogram passes with three barriers each, most of them over a digit all keys share) and, at the end, replaces
    // select + sorting network: the rank IS the output position.
    auto rank_of = [&](u64 x, int n_) {
        int r = 0;
        const ulonglong2* c2 = (const ulonglong2*)cand;
        for (int i = 0; i < n_; i += 2) {
            const ulonglong2 kk = c2[i >> 1];
            r += (kk.x < x ? 1 : 0) + ((i + 1 < n_ && kk.y < x) ? 1 : 0);
        }
        return r;
    };

    // ---- k-th best approximate score over all splits -> error band -> rows to re-rank
    if (n > p.k) {
        u64 kth;
        if (n <= THREADS) { // (one key per thread; beyond that the radix select is the faster one)
            if (tid < n) {
                const u64 x = cand[tid];
                if (rank_of(x, n) == p.k - 1) sh->ctl.kth = x;
            }
            __syncthreads();
            kth = sh->ctl.kth;
            // (wg_compact below starts with a barrier before anything is rewritten)
        } else {
            kth = wg_select_kth<THREADS>(cand, n, p.k, sh->hist, &sh->ctl);
        }
        const float e = flat_filter_err_bound(METRIC, p.d, p.xqn[q], p.yn_max, p.exact_inputs != 0);
        const float thr = band_threshold(key_score((uint32_t)(kth >> 32)), e);
        const u64 key_thr = ((u64)score_key(thr) << 32) | 0xffffffffull;
        wg_compact<THREADS>(cand, n, key_thr, &sh->ctl);
        n = (int)sh->ctl.cnt;
    }
    if (n > RR_CAND) {
        if (tid == 0) p.ovf_list[atomicAdd(p.ovf_cnt, 1u)] = (uint32_t)q;
        return;
    }

    // ---- exact fp32 distances, same chain as flat_scan_kernel / oracle orc_ip_chain
    const float xn = METRIC == METRIC_L2 ? p.xqn[q] : 0.f;
    for (int c = tid; c < n; c += THREADS) {
        const unsigned row = (unsigned)cand[c];
        const float* yr = p.xb ? p.xb + (int64_t)row * p.ldb : nullptr;
        const _Float16* yh = p.xb16 + (int64_t)row * p.ldb16;
        float acc = 0.f;
        // one 8-float step of the chain (the order of flat_scan_kernel's MFMA k-steps: e, 4 + e interleaved)
        auto step = [&](const f32x4& y0, const f32x4& y1, int s) {
            const f32x4 q0 = *(const f32x4*)(qs + s), q1 = *(const f32x4*)(qs + s + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc = __fmaf_rn(y0[e], q0[e], acc);
                acc = __fmaf_rn(y1[e], q1[e], acc);
            }
        };
        // The row is a random 512-byte (fp16 storage: 256-byte) read: all loads of a 32-float stretch are issued
        // before the chain consumes the first of them -- one memory round trip per stretch instead of one per step
        // (the chain itself stays the sequential one).
        int s = 0;
        if (yr) {
            for (; s + 32 <= p.dpad; s += 32) {
                f32x4 y[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) y[u] = *(const f32x4*)(yr + s + 4 * u);
#pragma unroll
                for (int u = 0; u < 4; ++u) step(y[2 * u], y[2 * u + 1], s + 8 * u);
            }
            for (; s < p.dpad; s += 8) step(*(const f32x4*)(yr + s), *(const f32x4*)(yr + s + 4), s);
        } else {
            // fp16 storage: the stored values, widened (exact), through the same chain
            auto widen = [](const half8& hv, f32x4& y0, f32x4& y1) {
                y0 = f32x4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
                y1 = f32x4{(float)hv[4], (float)hv[5], (float)hv[6], (float)hv[7]};
            };
            for (; s + 32 <= p.dpad; s += 32) {
                half8 hv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) hv[u] = *(const half8*)(yh + s + 8 * u);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    f32x4 y0, y1;
                    widen(hv[u], y0, y1);
                    step(y0, y1, s + 8 * u);
                }
            }
            for (; s < p.dpad; s += 8) {
                f32x4 y0, y1;
                widen(*(const half8*)(yh + s), y0, y1);
                step(y0, y1, s);
            }
        }
        float dis;
        if (METRIC == METRIC_L2) {
            dis = __fmaf_rn(-2.f, acc, xn + p.xbn[row]);
            dis = dis < 0.f ? 0.f : dis;
        } else {
            dis = acc;
        }
        cand[c] = ((u64)ordkey<METRIC>(dis) << 32) | row;
    }
    __syncthreads();

    // ---- exact top-k under (distance, id)
    if (n <= RR_RANK_MAX) {
        // the usual case (k plus the few rows inside the error band): every key straight to its rank
        const float pad = neutral_distance(METRIC);
        for (int c = tid; c < n; c += THREADS) {
            const u64 key = cand[c];
            const int r = rank_of(key, n);
            if (r < p.k) {
                const uint32_t wk = (uint32_t)(key >> 32);
                const bool ok = wk < kInvalidOrdKey;
                p.out_dis[(int64_t)q * p.k + r] = ok ? unordkey<METRIC>(wk) : pad;
                p.out_ids[(int64_t)q * p.k + r] = ok ? (int64_t)(uint32_t)key + p.id_base : -1;
            }
        }
        for (int i = n + tid; i < p.k; i += THREADS) {
            p.out_dis[(int64_t)q * p.k + i] = pad;
            p.out_ids[(int64_t)q * p.k + i] = -1;
        }
        return;
    }
    if (n > p.k) {
        const u64 kth = wg_select_kth<THREADS>(cand, n, p.k, sh->hist, &sh->ctl);
        wg_compact<THREADS>(cand, n, kth, &sh->ctl);
        n = (int)sh->ctl.cnt;
    }
    for (int i = tid; i < p.kp; i += THREADS) {
        unsigned wk = 0xffffffffu;
        int64_t wi = INT64_MAX;
        if (i < n) {
            const u64 key = cand[i];
            wk = (uint32_t)(key >> 32);
            wi = (int64_t)(uint32_t)key + p.id_base;
        }
        w_key[i] = wk;
        w_id[i] = wi;
    }
    __syncthreads();
    wg_bitonic_sort<THREADS>(w_key, w_id, p.kp);
    const float pad = neutral_distance(METRIC);
    for (int i = tid; i < p.k; i += THREADS) {
        float dis = pad;
        int64_t id = -1;
        if (i < n && w_key[i] < kInvalidOrdKey) {
            dis = unordkey<METRIC>(w_key[i]);
            id = w_id[i];
        }
        p.out_dis[(int64_t)q * p.k + i] = dis;
        p.out_ids[(int64_t)q * p.k + i] = id;
    }
}

void launch_flat_rerank(const FlatRerankParams& p, hipStream_t stream) {
    if (p.nq == 0) return;
    FA_THROW_IF_NOT(p.k >= 1 && p.k <= kMaxSelectionK && p.dpad % 8 == 0);
    FA_THROW_IF_NOT(p.gcap >= p.kp && p.gcap <= RR_GATHER && p.gcap % 2 == 0);
    const size_t lds = ((sizeof(RrShared) + 15) & ~(size_t)15) + (size_t)p.gcap * 8 +
                       (size_t)(p.dpad + (p.dpad & 1)) * 4 + (size_t)p.kp * 12;
    FA_THROW_IF_NOT_MSG(lds <= 160 * 1024, "re-rank workspace exceeds the LDS");
    static const char* e64 = experiment_env("FAISS_AMD_FLAT_RERANK_WAVE"); // A/B: 0 = 256-thread workgroups for every k
    const bool wave = p.k <= 64 && p.nsplit <= 64 && !(e64 && atoi(e64) == 0);
#define FA_RR(METRIC_, T_)                                                                                                     \
    do {                                                                                                                       \
        HIP_CHECK(hipFuncSetAttribute((const void*)flat_rerank_kernel<METRIC_, T_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)lds));                                                                              \
        hipLaunchKernelGGL((flat_rerank_kernel<METRIC_, T_>), dim3((unsigned)p.nq), dim3(T_), lds, stream, p);                 \
    } while (0)
    if (p.metric == METRIC_L2) {
        if (wave) FA_RR(METRIC_L2, 64);
        else FA_RR(METRIC_L2, RR_THREADS);
    } else {
        if (wave) FA_RR(METRIC_INNER_PRODUCT, 64);
        else FA_RR(METRIC_INNER_PRODUCT, RR_THREADS);
    }
#undef FA_RR
    HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------
// overflow fallback plumbing: gather flagged query rows, scatter their exact results back
// ---------------------------------------------------------------------------------
__global__ void gather_rows_kernel(const float* __restrict__ src, int64_t ld, int width,
                                   const uint32_t* __restrict__ list, float* __restrict__ dst) {
    const int64_t i = blockIdx.x;
    const float* r = src + (int64_t)list[i] * ld;
    for (int c = threadIdx.x; c < width; c += blockDim.x) dst[i * width + c] = r[c];
}
void launch_gather_rows(const float* src, int64_t ld, int width, const uint32_t* list, int n, float* dst,
                        hipStream_t stream) {
    if (n == 0) return;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)n), dim3(64), 0, stream, src, ld, width, list, dst);
    HIP_CHECK(hipGetLastError());
}
__global__ void scatter_results_kernel(const float* __restrict__ sd, const int64_t* __restrict__ si, int k,
                                       const uint32_t* __restrict__ list, float* __restrict__ dd,
                                       int64_t* __restrict__ di) {
    const int64_t i = blockIdx.x;
    const int64_t q = list[i];
    for (int c = threadIdx.x; c < k; c += blockDim.x) {
        dd[q * k + c] = sd[i * k + c];
        di[q * k + c] = si[i * k + c];
    }
}
void launch_scatter_results(const float* sd, const int64_t* si, int k, const uint32_t* list, int n, float* dd,
                            int64_t* di, hipStream_t stream) {
    if (n == 0) return;
    hipLaunchKernelGGL(scatter_results_kernel, dim3((unsigned)n), dim3(64), 0, stream, sd, si, k, list, dd, di);
    HIP_CHECK(hipGetLastError());
}

} // namespace faiss_amd
