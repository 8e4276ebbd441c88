// faiss_amd/csrc/ivf_lm_filter.hip -- list-major inverted-list search behind an f16 MFMA filter (round 4, gfx950).
//
// The list-major scan of round 3 (ivf_listmajor.hip) computes EVERY (query, row) distance of a batch on the f32 matrix
// pipe (v_mfma_f32_32x32x2_f32, 157 TFLOP/s) and reads a list once per 32-query group.  The f16 pipe is 16 x faster
// (v_mfma_f32_32x32x16_f16) and its B operands are half as wide, so here -- the scheme of the flat index
// (flat_filter.hip) carried into the lists -- the matrix pipe only ESTIMATES:
//   sweep 1 (MODE_MIN)      every probed row x every query that probes its list: estimate d~ on the f16 pipe; per lane the
//                           best estimate of every granule (16 G rows) goes to gmin[q][slot].  Nothing else is written.
//   bound                   T_q = k-th best granule estimate = an upper bound of the k-th best ESTIMATE of the query
//                           (granule minima are distinct rows); thr_q = T_q + 2 E_q with E_q >= |d~ - d| for every row
//                           (kernels.h ivf_filter_err_bound).  The k best rows by estimate have exact distance <= T_q + E_q, so
//                           a row of the exact top-k has estimate <= T_q + 2 E_q: the rows the second sweep collects are a
//                           SUPERSET of the exact answer, whatever the data.
//   sweep 2 (MODE_COLLECT)  the same sweep; rows with d~ <= thr_q are parked (wave-private LDS slice) and flushed to the
//                           query's candidate segment (~ k .. 2 k rows per query).
//   rerank                  the EXACT distance of every candidate, with the arithmetic of the QUERY-MAJOR scan
//                           (ivf_fused.hip: IVFFlat eight partial chains of (q - y)^2 + butterfly; IVFPQ the table sum on
//                           the query's power-of-two grid + the per-row term): a large batch now returns, bit for bit,
//                           what the same queries return one at a time (oracle: orc_ivf_search_ex, arith 0).
//   select                  select_k_kernel / wave_select_kernel over the segment, as for every other scan.
// A work item is (list, up to 96 of the queries probing it, a chunk of its rows) and belongs to ONE wavefront: the
// queries' fp16 coordinates are its B operands (32 VGPRs per 32-query block, three blocks), the rows' fp16 shadow
// (IVFFlat: arena_h, 2 bytes per coordinate) or the fp16 codebook entries their codes select (IVFPQ, codebook in LDS) are
// the A operands; a list is read once per sweep for up to 96 queries instead of once per 32.  The sweeps are bound by
// the row stream (IVFFlat: nb * d * 2 bytes per sweep) / the LDS gathers (IVFPQ), not by the matrix pipe any more.
// Shapes: IVFFlat d <= 512 (beyond 128: 16 / 24 / 32 k-steps per row, two or one query block per item); IVFPQ d <= 128,
// d % 16 == 0, dsub 1 / 2 / 4 / 8 k.  Work items are drawn from one counter per XCD (LmfDraw): the query groups of a list
// run behind the same L2.  DESIGN.md 3.10 holds the measurements every choice here rests on.
// Reference behaviour kept: faiss/gpu/impl/IVFInterleaved.cuh:33-224, PQScanMultiPassNoPrecomputed-inl.cuh:173-270
// (exhaustive scan of the probed lists, k best under (distance, scan position)).
#include "kernels.h"
#include "lmf_select.h"
#include <type_traits>

namespace faiss_amd {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned long long u64;

constexpr int LF_THREADS = 256;
constexpr int LF_PARK = 1024;                     // parked candidates per wave (a (32-row, 32-query) block always fits)
constexpr int LF_LDS = 4 * LF_PARK * (8 + 4) + 4 * (4 * (64 * 16 + 32) + 64 * 16); // 48 KB of parked candidates + 20.5 KB of staged records (LS_WAVE): two workgroups per CU
constexpr int MODE_MIN = 1, MODE_COLLECT = 2, MODE_DUMP = 3;

template <int METRIC>
__device__ __forceinline__ float lmf_worst() {
    return METRIC == METRIC_L2 ? INFINITY : -INFINITY;
}
template <int METRIC>
__device__ __forceinline__ float lmf_better(float a, float b) { // NaN-ignoring (v_min_f32 / v_max_f32 return the number)
    return METRIC == METRIC_L2 ? fminf(a, b) : fmaxf(a, b);
}
// stores the compiler does not see (see ivf_listmajor.hip, pass 1 of the register-fed kernel: with a store of its own in
// the block loop hipcc stops counting the loads in flight and waits vmcnt(0) before every use)
__device__ __forceinline__ void lmf_store_u32(uint32_t* at, uint32_t v) {
    asm volatile("global_store_dword %0, %1, off" ::"v"(at), "v"(v));
}
__device__ __forceinline__ void lmf_store_u64(u64* at, u64 v) {
    asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(at), "v"(v));
}
__device__ __forceinline__ void lmf_store_u16(uint16_t* at, uint32_t v) {
    asm volatile("global_store_short %0, %1, off" ::"v"(at), "v"(v));
}

bool ivf_lmf_supported(int kind, int d, int dpad, int M) {
    if ((dpad & 7) || d < 1) return false;
    if (kind == 0) return dpad <= 512; // (d > 128: one or two query blocks per item, see ivf_lmf_queries_per_item)
    // scalar quantizer (round 5): the IVFFlat sweeps over an fp16 copy of the CENTRED codes (exact in fp16), the scale folded
    // into per-(query, probe) B operands; M = the SqCodeType
    if (kind == 2) return d <= 512 && M >= SQ_U8 && M <= SQ_F16;
    if (dpad > 128) return false;
    if (kind == 1) {
        if (M < 4 || (M & 3) || d % M) return false;
        const int dsub = d / M;
        // an MFMA operand = 8 consecutive coordinates: whole sub-vectors (dsub 1, 2, 4, 8) or a piece of one (8 | dsub)
        if (!(dsub == 1 || dsub == 2 || dsub == 4 || (dsub & 7) == 0)) return false;
        return d == dpad && (d & 15) == 0; // (rows of a multiple of 16 coordinates: no tail handling in the decode)
    }
    return false;
}
// IVFPQ shapes served through the decoded-residual copy (lmf_pq_decode_kernel) when the codebook kernel does not take them
bool ivf_lmf_pq_decoded_supported(int d, int dpad, int M) {
    return (dpad & 7) == 0 && d >= 1 && dpad <= 512 && M >= 1 && d % M == 0;
}
// halfs per row of the fp16 shadow / the fp16 queries: d <= 128 whole 16-coordinate k-steps, beyond that whole groups of
// 8 k-steps (zeros behind d): the sweeps' block loops then have compile-time bounds (8 / 16 / 24 / 32 k-steps)
int ivf_lmf_row_halfs(int d) {
    return d <= 128 ? (d + 15) / 16 * 16 : (d + 127) / 128 * 128;
}
// B operands of a 32-query block: d / 4 VGPRs.  Three blocks at d <= 128 (96 VGPRs), two at d <= 256, one beyond.
static int lmf_query_blocks(int kind, int d) {
    return kind == 1 || d <= 128 ? kLmfQueryBlocks : d <= 256 ? 2 : 1;
}
int ivf_lmf_queries_per_item(int kind, int d) {
    return 32 * lmf_query_blocks(kind, d);
}

// ------------------------------------------------------------------ fp16 shadow of the IVFFlat rows (operand-major blocks)
__global__ void __launch_bounds__(256) lmf_shadow_kernel(const float* __restrict__ arena, int64_t ldv,
                                                         const float* __restrict__ arena_rn, int d, const uint32_t* list_len,
                                                         const int64_t* list_start, _Float16* __restrict__ arena_h, int dh,
                                                         unsigned* __restrict__ yn_max_bits,
                                                         const uint32_t* __restrict__ first_row) {
    const int list = blockIdx.x;
    const uint32_t len = list_len[list];
    const int64_t start = list_start[list]; // (a multiple of 32)
    const int nks = dh >> 4;
    float mx = 0.f;
    bool bad = false;
    // incremental maintenance (add()): only the 32-row blocks from row first_row[list] on are (re)written -- the blocks the
    // call appended to, or the whole list when it moved; 0xffffffff = the list did not change
    const uint32_t fr = first_row ? first_row[list] : 0u;
    if (fr == 0xffffffffu) return; // (workgroup-uniform)
    const int64_t b0 = fr >> 5;
    // piece i of the list's shadow: (block, k-step, lane) with the lane fastest -- consecutive threads write consecutive
    // 16-byte pieces; rows behind the end of the list are written as zeros
    const int64_t nblk = (len + 31) / 32 - b0;
    const int64_t total = nblk * nks * 64;
    for (int64_t i = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.y * blockDim.x) {
        const int ln = (int)(i & 63);
        const int64_t bs = i >> 6;
        const int s = (int)(bs % nks);
        const int64_t b = b0 + bs / nks;
        const int h = ln >> 5, j = ln & 31;
        const int64_t r = b * 32 + j;
        const int c = 16 * s + 8 * h;
        half8 o = half8{0, 0, 0, 0, 0, 0, 0, 0};
        if (r < (int64_t)len) {
            const float* src = arena + (start + r) * ldv + c;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = c + e < d ? src[e] : 0.f;
                if (!(fabsf(v) <= 65000.f)) bad = true; // NaN, inf, or beyond the fp16 normal range
                o[e] = (_Float16)v;
            }
            if (c == 0 && arena_rn) {
                const float n = arena_rn[start + r];
                if (!(n <= 3.0e38f)) bad = true;
                mx = fmaxf(mx, n);
            }
        }
        *(half8*)(arena_h + (((start >> 5) + b) * nks + s) * 512 + ln * 8) = o;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    const bool anybad = __ballot(bad) != 0ull;
    if ((threadIdx.x & 63) == 0) atomicMax(yn_max_bits, anybad ? 0x7f800000u : __float_as_uint(mx));
}
void launch_ivf_lmf_shadow(const float* arena, int64_t ldv, const float* arena_rn, int d, int nlist, const uint32_t* list_len,
                           const int64_t* list_start, void* arena_h, int dh, unsigned* yn_max_bits, const uint32_t* first_row,
                           hipStream_t stream) {
    if (nlist == 0) return;
    hipLaunchKernelGGL(lmf_shadow_kernel, dim3((unsigned)nlist, first_row ? 1 : 4), dim3(256), 0, stream, arena, ldv, arena_rn, d,
                       list_len, list_start, (_Float16*)arena_h, dh, yn_max_bits, first_row);
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------ scalar quantizer: fp16 copy of the centred codes
// component j of arena row `row` as the CENTRED code (code - mid; fp16 codes: the half itself): exactly representable in fp16
__device__ __forceinline__ float lmf_sq_centred(const uint8_t* arena, int64_t row, int j, int ct, int ld) {
    const int chb = sq_chunk_bytes(ct);
    const uint8_t* ch = arena + (row >> 6) * 64 * (int64_t)ld + (int64_t)(j >> 4) * 64 * chb + (row & 63) * chb;
    const int e = j & 15;
    if (ct == SQ_U8) return (float)ch[e] - 127.5f;
    if (ct == SQ_U4) return (float)((ch[e >> 1] >> (4 * (e & 1))) & 15u) - 7.5f;
    if (ct == SQ_U6) {
        const int off = 6 * e, by = off >> 3, sh = off & 7;
        const unsigned v = (unsigned)ch[by] | ((unsigned)ch[by + 1 < 12 ? by + 1 : by] << 8);
        return (float)((v >> sh) & 63u) - 31.5f;
    }
    return (float)*(const _Float16*)(ch + 2 * e);
}
// the operand-major blocks of lmf_shadow_kernel, filled with the centred codes.  stat_bits[0]: max |s o code'|^2 over the
// rows written (L2: from arena_rn; 0x7f800000 when a stored fp16 code is not finite), stat_bits[1]: max |code'|^2 (fp16 codes)
__global__ void __launch_bounds__(256) lmf_sq_shadow_kernel(const uint8_t* __restrict__ arena, int ct, int ld,
                                                            const float* __restrict__ arena_rn, int d, const uint32_t* list_len,
                                                            const int64_t* list_start, _Float16* __restrict__ arena_h, int dh,
                                                            unsigned* __restrict__ stat_bits, const uint32_t* __restrict__ first_row) {
    const int list = blockIdx.x;
    const uint32_t len = list_len[list];
    const int64_t start = list_start[list]; // (a multiple of 64)
    const int nks = dh >> 4;
    const uint32_t fr = first_row ? first_row[list] : 0u;
    if (fr == 0xffffffffu) return;
    const int64_t b0 = fr >> 5;
    const int64_t nblk = (len + 31) / 32 - b0;
    const int64_t total = nblk * nks * 64;
    float mx = 0.f, mc = 0.f;
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.y * blockDim.x) {
        const int ln = (int)(i & 63);
        const int64_t bs = i >> 6;
        const int s = (int)(bs % nks);
        const int64_t b = b0 + bs / nks;
        const int h = ln >> 5, j = ln & 31;
        const int64_t r = b * 32 + j;
        const int c = 16 * s + 8 * h;
        half8 o = half8{0, 0, 0, 0, 0, 0, 0, 0};
        if (r < (int64_t)len) {
            float sq = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = c + e < d ? lmf_sq_centred(arena, start + r, c + e, ct, ld) : 0.f;
                if (!(fabsf(v) <= 65000.f)) bad = true;
                o[e] = (_Float16)v;
                sq = __fmaf_rn(v, v, sq);
            }
            if (ct == SQ_F16) mc = fmaxf(mc, sq * (float)nks * 2.f); // (a piece's share times the pieces: an upper bound, refined below)
            if (c == 0 && arena_rn) {
                const float n = arena_rn[start + r];
                if (!(n <= 3.0e38f)) bad = true;
                mx = fmaxf(mx, n);
            }
        }
        *(half8*)(arena_h + (((start >> 5) + b) * nks + s) * 512 + ln * 8) = o;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        mc = fmaxf(mc, __shfl_xor(mc, off, 64));
    }
    const bool anybad = __ballot(bad) != 0ull;
    if ((threadIdx.x & 63) == 0) {
        atomicMax(stat_bits, anybad ? 0x7f800000u : __float_as_uint(mx));
        atomicMax(stat_bits + 1, __float_as_uint(mc));
    }
}
void launch_ivf_lmf_sq_shadow(const uint8_t* arena, int ct, int ld, const float* arena_rn, int d, int nlist, const uint32_t* list_len,
                              const int64_t* list_start, void* arena_h, int dh, unsigned* stat_bits, const uint32_t* first_row,
                              hipStream_t stream) {
    if (nlist == 0) return;
    hipLaunchKernelGGL(lmf_sq_shadow_kernel, dim3((unsigned)nlist, first_row ? 1 : 4), dim3(256), 0, stream, arena, ct, ld, arena_rn,
                       d, list_len, list_start, (_Float16*)arena_h, dh, stat_bits, first_row);
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------ IVFPQ beyond the codebook kernel: DECODED residuals
// IVFPQ shapes the LDS-codebook sweeps do not serve (d > 128: the fp16 codebook of d = 256 is 128 KB; d not a multiple of 16; dsub
// 3, 5, 6, ...) run the IVFFlat sweeps over an fp16 copy of the DECODED residuals r^ -- the operand-major blocks of
// lmf_shadow_kernel, 2 d bytes per row instead of M -- with the B operands and query terms per (query, probe) pair
// (lmf_sq_prepare_kernel with scale 1 and offset 0: fp16 (q - centroid), -|q - centroid|^2 / 2).  The estimates are the ones the
// codebook kernel computes (its A operands are the same fp16 codebook entries); bound, tightening and the exact rerank are IVFPQ's.
__global__ void __launch_bounds__(256) lmf_pq_decode_kernel(const uint8_t* __restrict__ arena_codes, const float* __restrict__ pq,
                                                            int d, int M, int dsub, const uint32_t* list_len,
                                                            const int64_t* list_start, _Float16* __restrict__ arena_h, int dh,
                                                            const uint32_t* __restrict__ first_row) {
    const int list = blockIdx.x;
    const uint32_t len = list_len[list];
    const int64_t start = list_start[list]; // (a multiple of 64)
    const int nks = dh >> 4;
    const uint32_t fr = first_row ? first_row[list] : 0u;
    if (fr == 0xffffffffu) return;
    const int64_t b0 = fr >> 5;
    const int64_t nblk = (len + 31) / 32 - b0;
    const int64_t total = nblk * nks * 64;
    for (int64_t i = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.y * blockDim.x) {
        const int ln = (int)(i & 63);
        const int64_t bs = i >> 6;
        const int s = (int)(bs % nks);
        const int64_t b = b0 + bs / nks;
        const int h = ln >> 5, j = ln & 31;
        const int64_t r = b * 32 + j;
        const int c = 16 * s + 8 * h;
        half8 o = half8{0, 0, 0, 0, 0, 0, 0, 0};
        if (r < (int64_t)len) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int jx = c + e;
                if (jx < d) {
                    const int m = jx / dsub, off = jx - m * dsub;
                    const unsigned code = arena_codes[pq_code_offset(M, start + r, m)];
                    o[e] = (_Float16)pq[((size_t)m * 256 + code) * dsub + off];
                }
            }
        }
        *(half8*)(arena_h + (((start >> 5) + b) * nks + s) * 512 + ln * 8) = o;
    }
}
void launch_ivf_lmf_pq_decode(const uint8_t* arena_codes, const float* pq, int d, int M, int nlist, const uint32_t* list_len,
                              const int64_t* list_start, void* arena_h, int dh, const uint32_t* first_row, hipStream_t stream) {
    if (nlist == 0) return;
    hipLaunchKernelGGL(lmf_pq_decode_kernel, dim3((unsigned)nlist, first_row ? 1 : 4), dim3(256), 0, stream, arena_codes, pq, d, M,
                       d / M, list_len, list_start, (_Float16*)arena_h, dh, first_row);
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------ IVFPQ: operand-major copy of the codes
void ivf_lmf_code_shadow_shape(int d, int M, int* bpl, int* piece) {
    const int dsub = d / M;
    const int ncode = dsub >= 8 ? 1 : 8 / dsub;
    *bpl = (d >> 4) * ncode;
    *piece = (*bpl % 16 == 0) ? 16 : 4;
}
__global__ void __launch_bounds__(256) lmf_code_shadow_kernel(const uint8_t* __restrict__ arena_codes, int d, int M,
                                                              const uint32_t* list_len, const int64_t* list_start,
                                                              uint8_t* __restrict__ arena_cs, int bpl, int piece,
                                                              const uint32_t* __restrict__ first_row) {
    const int list = blockIdx.x;
    const uint32_t len = list_len[list];
    const int64_t start = list_start[list]; // (a multiple of 64)
    const int dsub = d / M;
    const int ncode = dsub >= 8 ? 1 : 8 / dsub;
    const int npiece = (bpl + piece - 1) / piece;
    const uint32_t fr = first_row ? first_row[list] : 0u; // (incremental maintenance: see lmf_shadow_kernel)
    if (fr == 0xffffffffu) return;
    const int64_t b0 = fr >> 5;
    const int64_t nblk = (len + 31) / 32 - b0;
    const int64_t total = nblk * npiece * 64; // pieces of the list's shadow, lane fastest
    for (int64_t i = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.y * blockDim.x) {
        const int ln = (int)(i & 63);
        const int64_t bp = i >> 6;
        const int pc = (int)(bp % npiece);
        const int64_t b = b0 + bp / npiece;
        const int h = ln >> 5, j = ln & 31;
        const int64_t r = b * 32 + j;
        uint8_t* dst = arena_cs + ((start >> 5) + b) * (int64_t)(64 * npiece * piece) + ((int64_t)pc * 64 + ln) * piece;
        for (int u = 0; u < piece; ++u) {
            const int bb = pc * piece + u; // byte of the lane's share
            uint8_t v = 0;
            if (bb < bpl && r < (int64_t)len) {
                const int s = bb / ncode;
                const int m = (16 * s + 8 * h) / dsub + bb % ncode;
                v = arena_codes[pq_code_offset(M, start + r, m)];
            }
            dst[u] = v;
        }
    }
}
void launch_ivf_lmf_code_shadow(const uint8_t* arena_codes, int d, int M, int nlist, const uint32_t* list_len,
                                const int64_t* list_start, uint8_t* arena_cs, const uint32_t* first_row, hipStream_t stream) {
    if (nlist == 0) return;
    int bpl, piece;
    ivf_lmf_code_shadow_shape(d, M, &bpl, &piece);
    hipLaunchKernelGGL(lmf_code_shadow_kernel, dim3((unsigned)nlist, first_row ? 1 : 8), dim3(256), 0, stream, arena_codes, d, M,
                       list_len, list_start, arena_cs, bpl, piece, first_row);
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------ IVFPQ, PQ64 over d = 128: the codes with a COPY CHOICE
// The sweeps of this shape are bound by the LDS: every lane gathers 32 codebook entries of 4 bytes per 32-row block, and the
// 32 lanes of an access group ask one sub-quantizer's 1 KB table for 32 entries by 32 unrelated code bytes -- 32 balls into
// 32 banks, the fullest bank holds 3.2 on average, so a gather costs three LDS cycles where a conflict-free one costs one
// (PMC round 4: LDS instructions active 94 % of the busy cycles, 60 % of the LDS cycles conflict replays).  No layout of ONE
// table changes that.  TWO copies of it with different code -> bank maps do: copy 0 keeps entry c at slot c (bank c mod 32),
// copy 1 at slot rotr8(c, 3) (bank (c >> 3) mod 32), and for every (32-row block, sub-quantizer) the rows are dealt between
// the copies so that the fullest bank holds as little as possible (greedy: every row takes the less loaded of its two banks;
// 2.1 on average for unrelated codes) -- decided ONCE, when this copy of the codes is written, and stored with them: the sweeps
// read a 9-bit field (copy << 8 | slot) per gather instead of a code byte.  Both copies of all 64 tables are 128 KB of the
// 160 KB of LDS: [m][copy][256] entries of two halfs.
// Layout of a 32-row block (2560 bytes): per lane (h, j) 40 bytes = 8 k-steps x 5 bytes, stored as three pieces
// [64 lanes][16 B], [64 lanes][16 B], [64 lanes][8 B]; the 40 bits of k-step s (little endian) hold the fields of the gathers
// u = 0 .. 3 (sub-quantizer 8 s + 4 h + u of row j) at bits 9 u .. 9 u + 8.
constexpr int kLmfChoiceBlockBytes = 64 * 40;
__host__ __device__ static inline unsigned lmf_rotr8(unsigned c, int n) {
    return ((c >> n) | (c << (8 - n))) & 255u;
}
bool ivf_lmf_choice_shape(int d, int M) {
    return d == 128 && M == 64;
}
// one 64-thread workgroup per 32-row block: thread m deals the block's rows between the two copies of table m (phase 1),
// then thread (h, j) packs its lane's 32 fields (phase 2)
__global__ void __launch_bounds__(64) lmf_code_choice_kernel(const uint8_t* __restrict__ arena_codes, const uint32_t* list_len,
                                                             const int64_t* list_start, uint8_t* __restrict__ arena_cs,
                                                             const uint32_t* __restrict__ first_row) {
    constexpr int M = 64;
    __shared__ uint32_t copy_of[M]; // bit j: row j of the block reads sub-quantizer m from copy 1
    const int list = blockIdx.x;
    const uint32_t len = list_len[list];
    const int64_t start = list_start[list];
    const uint32_t fr = first_row ? first_row[list] : 0u;
    if (fr == 0xffffffffu) return;
    const int nblk = (int)((len + 31) / 32);
    const int tid = threadIdx.x;
    for (int b = (int)(fr >> 5) + (int)blockIdx.y; b < nblk; b += (int)gridDim.y) {
        const int64_t row0 = start + (int64_t)b * 32;
        const int nrow = min(32, (int)len - b * 32);
        {
            // phase 1: bank loads as 32 nibbles; rows in order, each to the less loaded of its two banks (ties: copy 0)
            const int m = tid;
            unsigned long long lo = 0ull, hi = 0ull; // nibble b of (hi:lo) = entries bank b holds so far (<= 15 by the cap below)
            uint32_t bits = 0u;
            for (int j = 0; j < nrow; ++j) {
                const unsigned c = arena_codes[pq_code_offset(M, row0 + j, m)];
                const unsigned b0 = c & 31u, b1 = (c >> 3) & 31u;
                const unsigned n0 = (unsigned)(((b0 < 16 ? lo : hi) >> (4 * (b0 & 15u))) & 15ull);
                const unsigned n1 = (unsigned)(((b1 < 16 ? lo : hi) >> (4 * (b1 & 15u))) & 15ull);
                const bool one = n1 < n0;
                const unsigned bk = one ? b1 : b0;
                if ((one ? n1 : n0) < 15u) {
                    if (bk < 16) lo += 1ull << (4 * bk);
                    else hi += 1ull << (4 * (bk & 15u));
                }
                bits |= one ? 1u << j : 0u;
            }
            copy_of[m] = bits;
        }
        __syncthreads();
        {
            const int h = tid >> 5, j = tid & 31;
            uint32_t out[10];
#pragma unroll
            for (int i = 0; i < 10; ++i) out[i] = 0u;
            if (j < nrow) {
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    unsigned long long v = 0ull;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int m = 8 * s + 4 * h + u;
                        const unsigned c = arena_codes[pq_code_offset(M, row0 + j, m)];
                        const unsigned one = (copy_of[m] >> j) & 1u;
                        const unsigned field = (one << 8) | (one ? lmf_rotr8(c, 3) : c);
                        v |= (unsigned long long)field << (9 * u);
                    }
                    // bytes 5 s .. 5 s + 4 of the lane's 40-byte string
                    const int o = 5 * s, i = o >> 2, r = o & 3;
                    out[i] |= (uint32_t)(v << (8 * r));
                    out[i + 1] |= (uint32_t)((r ? v >> (32 - 8 * r) : v >> 32));
                }
            }
            uint8_t* blk = arena_cs + ((start >> 5) + b) * (int64_t)kLmfChoiceBlockBytes;
            *(uint4*)(blk + tid * 16) = uint4{out[0], out[1], out[2], out[3]};
            *(uint4*)(blk + 1024 + tid * 16) = uint4{out[4], out[5], out[6], out[7]};
            *(uint2*)(blk + 2048 + tid * 8) = uint2{out[8], out[9]};
        }
        __syncthreads();
    }
}
void launch_ivf_lmf_code_choice(const uint8_t* arena_codes, int nlist, const uint32_t* list_len, const int64_t* list_start,
                                uint8_t* arena_cs, const uint32_t* first_row, hipStream_t stream) {
    if (nlist == 0) return;
    hipLaunchKernelGGL(lmf_code_choice_kernel, dim3((unsigned)nlist, first_row ? 4 : 32), dim3(64), 0, stream, arena_codes, list_len,
                       list_start, arena_cs, first_row);
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------ scores, shared by the two sweep kernels
// The epilogue of a 32-row x 32-query block works on SCORES, larger = better, whose order is the order of the estimates:
//   L2   score = <q', y'> - |y'|^2 / 2 - |q'|^2 / 2,   estimate = -2 score          IP   score = <q', y'> [+ coarse term] = estimate
// one fma per accumulator (the row term; -inf for rows that take no part: behind the end of the list, excluded by the
// IDSelector), then a maximum (8 x v_max3 per query block); the query's own term xh is added to that maximum only (sweep 1)
// or folded into the threshold (sweep 2, where the 16 scores are looked at one by one only when a lane has a hit).  The
// first version computed fmaf(-2, acc, |q'|^2 + |y'|^2), a bound check and a comparison per accumulator: 446 VALU
// instructions per 32-row block against 24 MFMAs (PMC, profiles/r04_e_pmc_ivfpq_10m.txt: SQ_INSTS_VALU) -- the sweeps were
// bound by the vector ALU.
struct LmfLane { // per (lane, query block)
    bool qv;
    uint32_t base_pos, qpr; // qpr = query << 11 | probe
    float xh;               // L2: -|q'|^2 / 2; IP: the coarse term (IVFPQ) or 0
    float tq;               // sweep 2: row scores (without xh) >= tq are collected (L2: -thr / 2 - xh; IP: thr - xh; +inf = nothing)
    float gm;               // sweep 1: best row score (without xh) of the granule so far
    uint32_t* gq;           // sweep 1: gmin + q * gstride + granule-slot base of this (query, probe) + h
    u64* kq;                // MODE_DUMP: keys + q * stride + base_pos
};
// threshold of sweep 2 on a row score (without the query's own term xh): scores >= it are collected.  thr = the query's
// threshold on the estimate.  "Nothing" (the bound kernel's sentinel for queries it hands to the redo path) becomes NaN:
// no score passes it, not even +inf.  "Everything" is clamped to -FLT_MAX: rows that take no part carry a score of -inf.
template <int METRIC>
__device__ __forceinline__ float lmf_collect_threshold(float thr, float xh) {
    const bool nothing = METRIC == METRIC_L2 ? thr == -INFINITY : thr == INFINITY;
    const float t = (METRIC == METRIC_L2 ? -0.5f * thr : thr) - xh;
    return nothing ? __builtin_nanf("") : fmaxf(t, -FLT_MAX);
}
template <int METRIC>
__device__ __forceinline__ float lmf_to_est(float sc) {
    return METRIC == METRIC_L2 ? -2.f * sc : sc;
}
// (volatile: the consumers of accumulators must stay behind lmf_scores' wait, see there)
__device__ __forceinline__ float lmf_max3(float a, float b, float c) {
    float r;
    asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float lmf_lane_max(const f32x16& a) { // 8 instructions
    const float m0 = lmf_max3(a[0], a[1], a[2]), m1 = lmf_max3(a[3], a[4], a[5]), m2 = lmf_max3(a[6], a[7], a[8]);
    const float m3 = lmf_max3(a[9], a[10], a[11]), m4 = lmf_max3(a[12], a[13], a[14]);
    return lmf_max3(lmf_max3(m0, m1, m2), lmf_max3(m3, m4, a[15]), a[15]);
}
// the accumulators of a block -> scores without the query's own term, in place: L2 acc - |y'|^2 / 2, IP acc; -inf for rows
// that take no part.  rn: |y'|^2 of the lane's rows 8 g + 4 h + e; tail: the block reaches past row r1 of the list; mw:
// IDSelector bits of the lane's rows (bit 8 g + e), SEL only.
template <int METRIC, bool SEL>
__device__ __forceinline__ void lmf_scores(f32x16& a, const f32x4 (&rn)[4], bool tail, int row_b, int r1, uint32_t mw) {
    if (METRIC == METRIC_L2) {
        // ONE v_fma_f32 per accumulator.  Rounds 4 / 5 asked for v_pk_fma_f32 here (two accumulators per instruction); round 6's
        // ablation (tools/pq_sweep_ablation.py, profiles/r6_pq_sweep_ablation.txt) put the always-on part of the epilogue at 0.6 of
        // the 3.2 ms of sweep 2 at nb = 100M -- ~ 520 cycles per 32-row block for 24 packed fma + 24 v_max3 --, and the guide's
        // timing table prices a packed f32 instruction beside MFMAs at + 22 cycles against two plain ones (the file is compiled
        // with -fno-slp-vectorize for the same reason: hipcc packs adjacent scalar f32 operations by itself).
#ifdef FAISS_AMD_LMF_PK_SCORES
        const f32x2 mh = f32x2{-0.5f, -0.5f};
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
                const f32x2 r2 = f32x2{rn[g][e], rn[g][e + 1]};
                f32x2 a2 = f32x2{a[4 * g + e], a[4 * g + e + 1]};
                a2 = __builtin_elementwise_fma(mh, r2, a2);
                a[4 * g + e] = a2[0];
                a[4 * g + e + 1] = a2[1];
            }
#else
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) a[4 * g + e] = __builtin_fmaf(-0.5f, rn[g][e], a[4 * g + e]);
#endif
    }
    else {
        // Inner product: nothing is computed on the accumulators, so the first instruction that reads them is the v_max3
        // of lmf_lane_max -- an asm statement the compiler's hazard recogniser does not look into, and the hardware does not
        // interlock a VALU read of a register an MFMA is still writing (19 wait states behind a 16-pass XDL write).  Found in
        // round 5: with ONE query block per item the v_max3 sat right behind the last MFMA and read stale maxima -- rows of
        // the answer were not collected, a few queries per thousand, depending on the instruction schedule.  (L2: the
        // compiler-visible v_pk_fma above reads the accumulators first and gets its wait states from the compiler.)
        asm volatile("s_nop 15\n\ts_nop 3" ::"v"(a));
    }
    // L2: rows behind the end of the chunk arrive with |y'|^2 = +inf (rn_fetch of the sweeps), so they are -inf already --
    // the 16 row tests hipcc hoisted out of the `tail` branch cost every block 32 VALU instructions.
    if ((tail && METRIC != METRIC_L2) || SEL) { // (wave-uniform)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool ok = !(METRIC != METRIC_L2 && tail && row_b + 8 * g + e >= r1) && (!SEL || ((mw >> (8 * g + e)) & 1u));
                a[4 * g + e] = ok ? a[4 * g + e] : -INFINITY;
            }
    }
}

// Candidate parking of sweep 2 (round 5).  Round 4 tested the 16 scores of every lane against the query's threshold in
// registers (80 VALU instructions per (row block, query block) that holds a hit), ran a wave scan and then 16 predicated
// store sequences -- and at nb <= 10M nearly EVERY pair holds a hit (250 candidates per query over 78 000 rows x 32 queries
// per block: 3 per pair at nb = 10M, 33 at nb = 1M), so sweep 2 cost twice sweep 1 at nb = 1M and a looser bound (sampled
// sweep 1) made it VALU-bound.  Now a lane whose maximum reaches its threshold dumps its 16 scores + its query's terms as
// one RECORD into a wave-private LDS staging area (slot = its rank among the hit lanes: one ballot + mbcnt, four 16-byte
// stores), and when the area is full a DENSE pass looks at the records with all 64 lanes -- 16 lanes per record, one
// score each, four records per step -- and appends the rows that pass to the parked candidates.
template <int NST> // records per wave: 64 (a pair may add 64 at once) or 32 (pairs are staged in two halves)
struct LmfStage {
    static constexpr int PLANE = NST * 16 + 32;        // bytes of one score plane [record][4 floats] (+ 8 banks: conflict-free reads)
    static constexpr int BYTES = 4 * PLANE + NST * 16; // + tq, pos, qpr, xh per record
    char* base;  // this wave's slice
    int cnt;     // (wave-uniform) records waiting
    __device__ __forceinline__ float* plane(int pl) const { return (float*)(base + pl * PLANE); }
    __device__ __forceinline__ float* tq() const { return (float*)(base + 4 * PLANE); }
    __device__ __forceinline__ uint32_t* pos() const { return (uint32_t*)(base + 4 * PLANE + NST * 4); }
    __device__ __forceinline__ uint32_t* qpr() const { return (uint32_t*)(base + 4 * PLANE + NST * 8); }
    __device__ __forceinline__ float* xh() const { return (float*)(base + 4 * PLANE + NST * 12); }
};
constexpr int LS_WAVE = LmfStage<64>::BYTES;
__device__ __forceinline__ int lmf_rank_in(unsigned long long bal) { // number of set bits below this lane
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
}
// lanes with `hit` write their record (room for all of them was made by the caller)
template <int NST>
__device__ __forceinline__ void lmf_stage_push(LmfStage<NST>& st, bool hit, unsigned long long bal, const f32x16& a, float tq,
                                               uint32_t pos0, uint32_t qpr, float xh) {
    const int idx = st.cnt + lmf_rank_in(bal);
    if (hit) {
#pragma unroll
        for (int g = 0; g < 4; ++g) *(f32x4*)(st.plane(g) + 4 * idx) = f32x4{a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]};
        st.tq()[idx] = tq;
        st.pos()[idx] = pos0;
        st.qpr()[idx] = qpr;
        st.xh()[idx] = xh;
    }
    st.cnt += __popcll(bal);
}
// parked candidates -> the queries' candidate lists in memory (slot = the query's counter).  Round 6: the returning atomics of up to
// four candidates per lane are ISSUED TOGETHER and waited for once -- one memory round trip per 256 candidates instead of four, each
// followed by a second one for the stores (the sweep's wavefronts do nothing else meanwhile: tools/pq_sweep_ablation.py, the epilogue
// was 0.29 of 0.55 ms of sweep 2 at nb = 10M).  Nothing waits for the stores: they are invisible to the compiler's count of the
// loads in flight, which only makes its waits conservative (an older load is complete whenever the counter allows it).
// (NOCONTEND: ablation only -- every atomic goes to a word of its own inside the key area, the slots are garbage)
template <bool NOCONTEND = false>
__device__ __forceinline__ void lmf_flush_parked(const IvfLmParams& p, int lane, const u64* pk_keys, const uint32_t* pk_q, int& wcnt) {
    for (int e0 = 0; e0 < wcnt; e0 += 256) {
        // (ONE asm statement from the first atomic to the wait: the compiler may copy an output register of an asm statement right
        // behind it -- it did, in front of a wait that stood in a statement of its own -- and would copy what the atomic has not
        // returned yet.  The lanes without a candidate are masked off inside the statement; exec is restored before it ends.)
        uint32_t slot[4];
        uint32_t* cp[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            cp[i] = p.cnt + (pk_q[e0 + 64 * i + lane] >> 11); // (slots behind wcnt: stale, masked off)
            if (NOCONTEND) cp[i] = (uint32_t*)p.keys + ((size_t)(blockIdx.x * 8 + (threadIdx.x >> 6)) * 256 + 64 * i + lane) * 16;
        }
        const int w0 = wcnt - e0, el = lane;
        const uint32_t one = 1u;
        unsigned long long sv;
        asm volatile(
                "s_mov_b64 %[sv], exec\n\t"
                "v_cmp_gt_i32 vcc, %[w0], %[el]\n\t"
                "s_and_b64 exec, %[sv], vcc\n\t"
                "global_atomic_add %[s0], %[a0], %[one], off sc0\n\t"
                "v_cmp_gt_i32 vcc, %[w1], %[el]\n\t"
                "s_and_b64 exec, %[sv], vcc\n\t"
                "global_atomic_add %[s1], %[a1], %[one], off sc0\n\t"
                "v_cmp_gt_i32 vcc, %[w2], %[el]\n\t"
                "s_and_b64 exec, %[sv], vcc\n\t"
                "global_atomic_add %[s2], %[a2], %[one], off sc0\n\t"
                "v_cmp_gt_i32 vcc, %[w3], %[el]\n\t"
                "s_and_b64 exec, %[sv], vcc\n\t"
                "global_atomic_add %[s3], %[a3], %[one], off sc0\n\t"
                "s_mov_b64 exec, %[sv]\n\t"
                "s_waitcnt vmcnt(0)"
                : [s0] "=&v"(slot[0]), [s1] "=&v"(slot[1]), [s2] "=&v"(slot[2]), [s3] "=&v"(slot[3]), [sv] "=&s"(sv)
                : [a0] "v"(cp[0]), [a1] "v"(cp[1]), [a2] "v"(cp[2]), [a3] "v"(cp[3]), [one] "v"(one), [el] "v"(el), [w0] "s"(w0),
                  [w1] "s"(w0 - 64), [w2] "s"(w0 - 128), [w3] "s"(w0 - 192)
                : "vcc", "memory");
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = e0 + 64 * i + lane;
            if (NOCONTEND) slot[i] &= 255u;
            if (e < wcnt && (int64_t)slot[i] < p.stride) {
                const uint32_t qp = pk_q[e];
                const int64_t at = (int64_t)(qp >> 11) * p.stride + slot[i];
                lmf_store_u64(p.keys + at, pk_keys[e]);
                lmf_store_u16(p.cand_pr + at, qp & 2047u);
            }
        }
    }
    wcnt = 0;
}
// dense pass over the staged records: parked candidates go to pk_keys / pk_q (wcnt of them so far; `flush`
// empties them).  Round 6: LANE l looks at RECORD l -- its 16 scores arrive as four 16-byte reads, one (prefetched) per step, and a
// step appends the rows that pass of four score positions.  The first version walked the records four at a time, 16 lanes each, and
// paid two dependent LDS round trips per step through an LDS that eight wavefronts gather from: ~ 30 of them per 64 records.
template <int METRIC, int PARK, int NST, typename Flush>
__device__ __forceinline__ void lmf_stage_expand(LmfStage<NST>& st, int lane, u64* pk_keys, uint32_t* pk_q, int& wcnt, Flush&& flush) {
    // score positions per step: as many as always fit behind a flush (4 with the sweeps' 256 .. 1024 parked candidates per wave; the
    // two-copy codebook leaves room for 64 candidates and 32 records: 2)
    constexpr int EPS = PARK >= 4 * NST ? 4 : PARK >= 2 * NST ? 2 : 1;
    static_assert(NST <= 64 && PARK >= NST, "a record per lane; a score position of every record fits");
    const bool valid = lane < st.cnt;
    const int rc = valid ? lane : 0;
    const float tq = valid ? st.tq()[rc] : __builtin_nanf(""); // (nothing passes NaN)
    const uint32_t pos0 = st.pos()[rc], qpr = st.qpr()[rc];
    const float xh = st.xh()[rc];
    f32x4 nx = *(const f32x4*)(st.plane(0) + 4 * rc);
#pragma unroll 1
    for (int g = 0; g < 4; ++g) {
        const f32x4 s4 = nx;
        nx = *(const f32x4*)(st.plane(min(g + 1, 3)) + 4 * rc);
#pragma unroll
        for (int e0 = 0; e0 < 4; e0 += EPS) {
            bool ps[EPS];
            unsigned long long bl[EPS];
            int first[EPS + 1];
            first[0] = 0;
#pragma unroll
            for (int e = 0; e < EPS; ++e) {
                ps[e] = s4[e0 + e] >= tq;
                bl[e] = __ballot(ps[e]);
                first[e + 1] = first[e] + __popcll(bl[e]);
            }
            const int n = first[EPS];
            if (n == 0) continue;
            if (wcnt + n > PARK) flush();
#pragma unroll
            for (int e = 0; e < EPS; ++e) {
                if (ps[e]) {
                    const int at = wcnt + first[e] + lmf_rank_in(bl[e]);
                    // row of score 4 g + e of a lane: 8 g + e rows behind the lane's first
                    pk_keys[at] = ((u64)ordkey<METRIC>(lmf_to_est<METRIC>(s4[e0 + e] + xh)) << 32) | (pos0 + (uint32_t)(8 * g + e0 + e));
                    pk_q[at] = qpr;
                }
            }
            wcnt += n;
        }
    }
    st.cnt = 0;
}
// the lanes of a (row block, query block) pair whose best score reaches their query's threshold stage their 16 scores
template <int NST, typename Expand>
__device__ __forceinline__ void lmf_collect_pair(LmfStage<NST>& st, int lane, bool hit, const f32x16& a, float tq, uint32_t pos0,
                                                 uint32_t qpr, float xh, Expand&& expand) {
    if (NST >= 64) {
        const unsigned long long bal = __ballot(hit);
        if (!bal) return; // (wave-uniform)
        if (st.cnt + __popcll(bal) > NST) expand();
        lmf_stage_push(st, hit, bal, a, tq, pos0, qpr, xh);
    } else { // (a half of the lanes at a time: <= 32 records)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const bool hh = hit && (lane >> 5) == half;
            const unsigned long long bal = __ballot(hh);
            if (!bal) continue;
            if (st.cnt + __popcll(bal) > NST) expand();
            lmf_stage_push(st, hh, bal, a, tq, pos0, qpr, xh);
        }
    }
}

// Work items of a sweep, drawn per XCD.  The plan emits the items in list order -- the query groups of one (list, row
// chunk) next to each other -- and a list that meets more queries than an item holds is read once per group: the item range
// is cut into eight contiguous parts, the wavefronts of XCD x (block b runs on XCD b % 8: observed, a matter of speed only)
// draw from part x first, so that the groups of a list run at about the same time behind ONE L2 and the second group's
// rows hit it (IVFFlat nb = 10M: 4.06 GB of L2 misses per sweep against 2.6 GB of unique bytes with one global counter).
// An XCD that runs dry takes the next part's items.  One counter per XCD also keeps the dequeue rate per word low
// (a single word saturates near 88 dequeues / us; the nb = 1M sweeps draw ~50 items / us).
struct LmfDraw {
    uint32_t* ctr; // counters of this sweep, 32 words apart
    uint32_t it0, n, per;
    int cur, tried;
    __device__ LmfDraw(uint32_t* item_bounds, int sweep, uint32_t it0_, uint32_t it1)
            : ctr(item_bounds + kLmXcdCtr + sweep * 8 * 32), it0(it0_), n(it1 - it0_), per((it1 - it0_ + 7) / 8),
              cur((int)(blockIdx.x & 7)), tried(0) {}
    // next item of the wavefront (wave-uniform), or >= it1 when every part is exhausted
    __device__ __forceinline__ uint32_t next(int lane) {
        while (tried < 8) {
            uint32_t i = 0;
            if (lane == 0) i = atomicAdd(ctr + cur * 32, 1u);
            i = (uint32_t)__builtin_amdgcn_readfirstlane((int)i);
            const uint32_t g = (uint32_t)cur * per + i;
            if (i < per && g < n) return it0 + g;
            cur = (cur + 1) & 7;
            ++tried;
        }
        return 0xffffffffu;
    }
};

// ------------------------------------------------------------------ IVFFlat sweep
// One WAVEFRONT per work item, items drawn from a counter; A operands global -> registers one 32-row block ahead, refilled
// right behind the MFMAs that consumed them (the walk of ivf_lm_flat_reg_kernel).  KS: k-steps of a row the loops are
// unrolled for (8: d <= 128, 16 / 24 / 32: d <= 256 / 384 / 512 with fewer query blocks per item); FULL: ldh == 16 KS.
// The A operands wait in a ring of R pieces (a whole block at d <= 128, half a block beyond: the registers go to the B
// operands): the shadow is operand-major, so the ring simply runs R KB ahead of the MFMAs through the k-steps of this block
// and the next one looked at.
// PAIRB (scalar quantizer): the B operands and the query terms are per (query, probe) PAIR -- pair16 / pair_xh, prepared once per
// search by lmf_sq_prepare_kernel -- instead of per query; the A operands are the fp16 copy of the centred codes.
// PAIR (round 6): two-wave workgroups that walk the two query groups of a (list, row chunk) IN LOCK-STEP.  A list
// probed by more than 96 queries of the batch is an item per group of 96, and every group streams the chunk's rows again: 1.41 x
// the unique bytes per sweep at nb = 10M although sibling items are drawn back to back on one XCD -- two free-running wavefronts
// drift apart by more than the few microseconds a line survives in a 4 MB L2 that turns over at 0.6 TB/s.  Here the workgroup draws
// ONE item; if the next item is its sibling (same list, same chunk, next group) wave 1 takes that one and the two waves meet at a
// barrier before every 32-row block: the second read of a block hits the CU's L1 / the XCD's L2.  Without a sibling the two waves
// split the item's rows.  (The drawer of a sibling item skips it: it belongs to the workgroup that drew the item in front.)
template <int METRIC, int MODE, int NQB, int KS, bool FULL, bool SEL, bool PAIRB, bool PAIR = false>
__global__ void __launch_bounds__(PAIR ? 128 : LF_THREADS, 2) ivf_lmf_flat_kernel(IvfLmParams p) {
    static_assert(KS == 8 || (FULL && (KS == 16 || KS == 24 || KS == 32)), "k-steps");
    constexpr int NW = PAIR ? 2 : 4; // wavefronts per workgroup
    constexpr int R = KS == 8 ? 8 : KS / 2; // (KS % R == 0: slot s % R holds k-step s of the block at its start)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const int j = lane & 31;
    const int np = p.nprobe;
    const int nks = FULL ? KS : (int)(p.ldh >> 4);
    const int gsh = __builtin_ctz((unsigned)p.gran_blocks); // blocks per granule: a power of two
    const _Float16* xq16 = (const _Float16*)p.xq16;
    const _Float16* arena_h = (const _Float16*)p.arena_h;
    u64* pk_keys = (u64*)smem + wave * LF_PARK;
    uint32_t* pk_q = (uint32_t*)(smem + NW * LF_PARK * 8) + wave * LF_PARK;
    // |y|^2 of the rows of TWO consecutive blocks of this wave (see the block loop)
    __shared__ float rn_lds_all[NW][64];
    __shared__ uint32_t pair_it; // PAIR: the item the workgroup drew
    float* rn_lds = rn_lds_all[wave];
    int wcnt = 0; // (wave-uniform) parked candidates
    LmfStage<64> st{smem + NW * LF_PARK * 12 + wave * LS_WAVE, 0};
    auto flush = [&]() __attribute__((always_inline)) { lmf_flush_parked(p, lane, pk_keys, pk_q, wcnt); };
    auto expand = [&]() __attribute__((always_inline)) { lmf_stage_expand<METRIC, LF_PARK, 64>(st, lane, pk_keys, pk_q, wcnt, flush); };

    const uint32_t it0 = p.item_bounds[1], it1 = p.item_bounds[2];
    LmfDraw draw(p.item_bounds, MODE == MODE_MIN ? 0 : 1, it0, it1);
    for (;;) {
        uint32_t it;
        if (PAIR) {
            if (wave == 0) {
                it = draw.next(lane);
                if (lane == 0) pair_it = it;
            }
            __syncthreads();
            it = pair_it;
            __syncthreads(); // (wave 0 must not draw again before wave 1 has read the word)
        } else {
            it = draw.next(lane);
        }
        if (it >= it1) break;
        IvfLmItem item = p.items[it];
        bool paired = false;
        if (PAIR) {
            // (all of this is wave- and workgroup-uniform: both waves see the same items)
            if ((item.qt & 1) && it > it0) {
                const IvfLmItem pv = p.items[it - 1];
                const bool follower = pv.bucket == item.bucket && pv.rt == item.rt && pv.both == item.both && pv.qt + 1 == item.qt;
                if (__builtin_amdgcn_readfirstlane((int)follower)) continue;
            }
            if (!(item.qt & 1) && it + 1 < it1) {
                const IvfLmItem nx = p.items[it + 1];
                paired = nx.bucket == item.bucket && nx.rt == item.rt && nx.both == item.both && nx.qt == item.qt + 1;
                paired = __builtin_amdgcn_readfirstlane((int)paired) != 0;
                if (paired && wave == 1) item = nx;
            }
        }
        const int bk = __builtin_amdgcn_readfirstlane(item.bucket);
        const int qt = __builtin_amdgcn_readfirstlane(item.qt);
        const int rt = __builtin_amdgcn_readfirstlane(item.rt);
        const int list = bk >> 1;
        const int len = (int)p.list_len[list];
        const int64_t start = p.list_start[list];
        const uint32_t pb = p.bucket_start[bk];
        const int npair = min(32 * NQB, (int)(p.bucket_start[bk + 1 + item.both] - pb) - qt * (32 * NQB));
        const int r0 = rt * p.rows_per_item;
        const int r1 = min(len, r0 + p.rows_per_item);
        // sweep 1 may look at a prefix of the chunk only (IvfLmParams::sample_rows): rows [r0, rend); its granule slots are
        // then numbered over the sampled granules of the list (ivf_lmf_list_granules)
        const bool smp = MODE == MODE_MIN && p.sample_rows > 0 && p.sample_rows < p.rows_per_item;
        const int rend_item = smp ? min(r1, r0 + p.sample_rows) : r1;
        const int gbase = smp ? rt * (p.sample_rows >> (5 + gsh)) - ((r0 >> 5) >> gsh) : 0;
        // PAIR without a sibling: the two waves split the rows at a granule boundary (a granule's minimum has ONE writer)
        int ra = r0, rend = rend_item;
        if (PAIR && !paired) {
            const int gr = (32 << gsh) * (MODE == MODE_MIN ? p.min_stride : 1);
            const int mid = min(rend_item, r0 + (((rend_item - r0 + 1) / 2 + gr - 1) / gr) * gr);
            if (wave == 0) rend = mid;
            else ra = mid;
        }

        // ---- this lane's queries: one per 32-query block
        LmfLane L[NQB];
        half8 bq[NQB][KS];
#pragma unroll
        for (int b = 0; b < NQB; ++b) {
            const int my = b * 32 + j;
            L[b].qv = my < npair;
            const uint32_t pi = p.pairs[pb + (uint32_t)(qt * (32 * NQB)) + (uint32_t)(L[b].qv ? my : 0)];
            const int q = (int)(pi / (uint32_t)np);
            const int pr = (int)(pi - (uint32_t)q * (uint32_t)np);
            const _Float16* qrow = (PAIRB ? (const _Float16*)p.pair16 + (int64_t)pi * p.ldq16 : xq16 + (int64_t)q * p.ldq16) + 8 * h;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (FULL || s < nks) bq[b][s] = *(const half8*)(qrow + 16 * s);
                else bq[b][s] = half8{0, 0, 0, 0, 0, 0, 0, 0};
            }
            L[b].xh = PAIRB ? p.pair_xh[pi] : METRIC == METRIC_L2 ? -0.5f * p.xqn[q] : 0.f;
            L[b].base_pos = p.prefix[(int64_t)q * (np + 1) + pr];
            L[b].qpr = ((uint32_t)q << 11) | (uint32_t)pr;
            L[b].tq = __builtin_nanf(""); // nothing passes a NaN threshold -- not even a score of +inf (a query beyond the fp16 range)
            L[b].gm = -INFINITY;
            L[b].gq = nullptr;
            L[b].kq = nullptr;
            if (MODE == MODE_MIN) L[b].gq = p.gmin + (int64_t)q * p.gstride + p.prefixg[(int64_t)q * (np + 1) + pr] + h;
            if (MODE == MODE_COLLECT && L[b].qv) L[b].tq = lmf_collect_threshold<METRIC>(p.thr_f[q], L[b].xh);
            if (MODE == MODE_DUMP) L[b].kq = p.keys + (int64_t)q * p.stride + L[b].base_pos;
        }

        int t = PAIR ? ra : r0;
        while (t < rend) {
            // ---- (re-)entry: the rows of block t.  Rows behind the end of the list belong to the next list or the
            // arena's padding: loaded, never looked at.
            // (operand-major shadow: k-step s of block b is the KB at (b * nks + s) * 1024, lane l its 16-byte piece l)
            const _Float16* arow = arena_h + ((start + t) >> 5) * (int64_t)(nks * 512) + lane * 8;
            half8 a[R];
#pragma unroll
            for (int s = 0; s < R; ++s) {
                if (FULL || s < nks) a[s] = *(const half8*)(arow + 512 * s);
                else a[s] = half8{0, 0, 0, 0, 0, 0, 0, 0};
                asm volatile("" ::: "memory");
            }
            // sweep 1 may look at every min_stride-th block only (a SAMPLE of the rows still bounds the k-th best estimate
            // from above; fewer rows -> a looser bound -> more candidates in sweep 2)
            const int bstep = MODE == MODE_MIN ? 32 * p.min_stride : 32;
            // |y|^2 of the rows, for the epilogue.  Every lane of a half needs the same 16 of a block's 32 values, and it needs
            // them a memory latency EARLIER than the epilogue of a 24-MFMA block can wait (loaded inside the block, as the f32
            // kernels do, each block stalled ~3000 cycles on them: the sweeps ran at a fifth of the matrix pipe with every
            // unit idle, profiles/r04_f_pmc_*).  So: lane l fetches ONE value per block (row l & 31), four blocks ahead, through a
            // register ring; at the start of a block the values go to the wave's LDS slice and the epilogue reads its 16 back
            // as four broadcast ds_read_b128.
            const int tb = t; // first block of this run (runs restart behind a flush)
            auto rn_fetch = [&](int blk) __attribute__((always_inline)) -> float {
                // (the load stays unconditional; rows behind the chunk get |y|^2 = +inf: their scores become -inf by themselves)
                const int row = tb + blk * bstep + (lane & 31);
                if (METRIC != METRIC_L2) return 0.f;
                const float v = p.arena_rn[start + min(row, r1 - 1)];
                return row < r1 ? v : INFINITY;
            };
            float pf0 = rn_fetch(0), pf1 = rn_fetch(1), pf2 = rn_fetch(2), pf3 = rn_fetch(3);
            int bi = 0; // blocks of this run so far
            for (; t < rend; t += bstep, ++bi) {
                if (PAIR && paired) __syncthreads(); // lock-step with the sibling item's wave (same rows, same trip count)
                const _Float16* acur = arow; // (KS > R: the ring refills from this block first)
                arow += (bstep >> 5) * nks * 512;
                if (METRIC == METRIC_L2) { // (no branch: unconditional loads keep the compiler counting them)
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_wave_barrier(); // (the reads of the previous block were issued: LDS keeps a wave's order)
                    rn_lds[lane] = pf0;              // (both halves write the block's 32 values)
                    pf0 = pf1, pf1 = pf2, pf2 = pf3;
                    pf3 = rn_fetch(bi + 4);
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                }
                const int row_b = t + 4 * h;    // row of the list of acc[.][4 g + e]: row_b + 8 g + e
                const bool tail = t + 32 > r1;  // (wave-uniform) the block reaches past the end of the chunk
                // IDSelector: one bit per arena row (launch_selector_mask); a block = one aligned word of the mask (lists
                // start on multiples of 32 rows).  Rows the selector excludes take no part in the bound nor in the collection.
                const uint32_t mw = SEL ? p.sel_mask[(start + t) >> 5] >> (4 * h) : 0u;
                f32x16 acc[NQB];
#pragma unroll
                for (int b = 0; b < NQB; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
                f32x4 rn[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) rn[g] = f32x4{0.f, 0.f, 0.f, 0.f};
                // (the refills are UNCONDITIONAL -- behind the last block they read the rows that follow the list)
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    if (FULL || s < nks) {
#pragma unroll
                        for (int b = 0; b < NQB; ++b)
                            acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s % R], bq[b][s], acc[b], 0, 0, 0);
                        a[s % R] = s + R < KS ? *(const half8*)(acur + 512 * (s + R)) : *(const half8*)(arow + 512 * (s + R - KS));
                    }
                    if (s == 5 && METRIC == METRIC_L2) { // the block's norms from the wave's slice (rows 8 g + 4 h + e)
#pragma unroll
                        for (int g = 0; g < 4; ++g) rn[g] = *(const f32x4*)(rn_lds + 8 * g + 4 * h);
                    }
                }
                if (FULL) {
                    // the instruction order above IS the schedule: NQB MFMAs, one load, KS times
#pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        __builtin_amdgcn_sched_group_barrier(0x008, NQB, 0); // MFMA
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (MODE == MODE_MIN) {
#pragma unroll
                    for (int b = 0; b < NQB; ++b) {
                        lmf_scores<METRIC, SEL>(acc[b], rn, tail, row_b, r1, mw);
                        {
                        const float lm = lmf_lane_max(acc[b]);
                        L[b].gm = lmf_max3(L[b].gm, lm, lm); // (no canonicalising v_max x, x around it)
                    }
                    }
                    const int blk = t >> 5;
                    // (wave-uniform) the granule ends with this block: the next block looked at lies in another one
                    if ((((t + bstep) >> 5) >> gsh) != (blk >> gsh) || t + bstep >= rend) {
#pragma unroll
                        for (int b = 0; b < NQB; ++b) {
                            if (L[b].qv)
                                lmf_store_u32(L[b].gq + 2 * ((blk >> gsh) + gbase), ordkey<METRIC>(lmf_to_est<METRIC>(L[b].gm + L[b].xh)));
                            L[b].gm = -INFINITY;
                        }
                    }
                } else if constexpr (MODE == MODE_DUMP) {
#pragma unroll
                    for (int b = 0; b < NQB; ++b) {
                        lmf_scores<METRIC, SEL>(acc[b], rn, tail, row_b, r1, mw);
#pragma unroll
                        for (int g = 0; g < 4; ++g)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int rowl = row_b + 8 * g + e;
                                if (L[b].qv && rowl < r1)
                                    lmf_store_u64(L[b].kq + rowl,
                                                  ((u64)ordkey<METRIC>(lmf_to_est<METRIC>(acc[b][4 * g + e] + L[b].xh)) << 32) |
                                                          (u64)(L[b].base_pos + (uint32_t)rowl));
                            }
                    }
                } else {
                    // ---- lanes whose best score of query block b reaches their query's threshold stage their 16 scores (the scores
                    // of all query blocks first: the row norms are dead by the time a dense pass needs registers)
                    bool hit[NQB];
#pragma unroll
                    for (int b = 0; b < NQB; ++b) {
                        lmf_scores<METRIC, SEL>(acc[b], rn, tail, row_b, r1, mw);
                        hit[b] = lmf_lane_max(acc[b]) >= L[b].tq;
                    }
#pragma unroll
                    for (int b = 0; b < NQB; ++b)
                        lmf_collect_pair(st, lane, hit[b], acc[b], L[b].tq, L[b].base_pos + (uint32_t)row_b, L[b].qpr, L[b].xh, expand);
                }
            }
        }
    }
    if (MODE == MODE_COLLECT) {
        if (st.cnt > 0) expand();
        if (wcnt > 0) flush();
    }
}

// ------------------------------------------------------------------ IVFPQ sweep, fp16 codebook in LDS
// The codebook [M][256][dsub] as fp16 (d * 512 bytes: 64 KB at d = 128) lives in LDS for the life of a persistent 8-wave
// workgroup; a WAVEFRONT works alone on an item (list, up to 96 queries, row chunk): B operands = fp16 (q - centroid)
// (inner product: fp16 q) of its queries, the block's code bytes global -> registers (one block ahead) -> a private
// 32-row LDS slice (un-rotated on the way in, kernels.h pq_code_offset), then per k-step one aligned read of the operand's
// code bytes and 1 .. 8 codebook gathers: ONE decoded operand feeds the MFMAs of all query blocks.
// DS: 1 / 2 / 4 / 8 = dsub itself (8 also: any multiple of 8).
constexpr int LP_THREADS = 512;
constexpr int LP_BR = 32;    // rows per block
constexpr int LP_PARK = 256; // parked candidates per wave
constexpr int LP_AHEAD = 3;  // k-steps the codebook gathers run ahead of the MFMAs (ring of 4 operands)
// TWOC (the two-copy codebook of PQ64 over d = 128, lmf_code_choice_kernel): 128 KB of tables leave 30 KB for the eight
// wavefronts' parked candidates (64 each) and staged records (32 each: pairs are staged half by half)
constexpr int LP_PARK2 = 64, LP_NST2 = 32;
struct LpLayout {
    int cb_bytes, off_park, off_stage, total, off_rn;
};
// fastg (round 6, PQ64 over d = 128): the 64 tables of 1 KB in two halves -- sub-quantizers with (m & 4) == 0 at LDS bytes [0, 32 K),
// the others at [64 K, 96 K) -- so that the table of k-step s, slot u is the IMMEDIATE offset (4 s + u) KB of a ds_read_b32 for both
// halves of a wavefront (lanes 32 .. 63 carry bit 16 in their address register); parked candidates + row norms sit in the hole
// [32 K, 64 K), the staged records behind the upper tables
__host__ __device__ static inline LpLayout lp_layout(int d, int M, bool twoc = false, bool fastg = false) {
    LpLayout L;
    if (fastg) {
        L.cb_bytes = 98304;
        L.off_park = 32768;
        L.off_rn = 32768 + 8 * LP_PARK * (8 + 4);
        L.off_stage = 98304;
        L.total = L.off_stage + 8 * LmfStage<64>::BYTES;
        return L;
    }
    L.off_rn = 0;
    L.cb_bytes = d * 256 * 2 * (twoc ? 2 : 1);
    L.off_park = (L.cb_bytes + 15) & ~15;
    L.off_stage = L.off_park + 8 * (twoc ? LP_PARK2 : LP_PARK) * (8 + 4);
    L.total = L.off_stage + 8 * (twoc ? LmfStage<LP_NST2>::BYTES : LmfStage<64>::BYTES); // (the staged records of sweep 2)
    return L;
}
// code dwords a lane holds per 32-row block (IvfLmParams::cs_bpl / 4, at most): 8 k-steps x (8 / dsub) codes; TWOC: 8 x 5 bytes
template <int DS, bool TWOC>
struct LpCodes {
    static constexpr int ND = TWOC ? 10 : DS == 1 ? 16 : DS == 2 ? 8 : DS == 4 ? 4 : 2;
};

// FULLK: d == 128 (8 k-steps, no runtime bound in the operand pipeline) and, for DS == 2, M == 64 (two 16-byte code pieces
// per lane and block): the block loop then holds no conditional code around its loads and LDS reads -- with runtime bounds
// hipcc closed every k-step with lgkmcnt(0) / vmcnt(0) (the gathers of step s + 2 were waited for before the MFMAs of step
// s, the code prefetch before the next instruction): 8400 cycles per 24-MFMA block, every unit idle (profiles/r04_g_pmc_*).
// the four codebook gathers of one k-step of the FG sweeps (below): S_ = k-step inside the block, tables 8 S_ + 4 h + u at the
// immediate offsets (4 S_ + u) KB; c4 = the four code bytes; va0 / va1 = address registers whose high half holds the lane's
// table half (h << 16; the SDWA shifts rewrite the low half only)
template <int S_>
__device__ __forceinline__ void lp_gather(unsigned c4, unsigned (&dst)[4], unsigned& va0, unsigned& va1, unsigned two) {
    asm volatile(
            "v_lshlrev_b32_sdwa %4, %7, %6 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_0\n\t"
            "v_lshlrev_b32_sdwa %5, %7, %6 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_1\n\t"
            "ds_read_b32 %0, %4 offset:%8\n\t"
            "ds_read_b32 %1, %5 offset:%9\n\t"
            "v_lshlrev_b32_sdwa %4, %7, %6 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_2\n\t"
            "v_lshlrev_b32_sdwa %5, %7, %6 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_3\n\t"
            "ds_read_b32 %2, %4 offset:%10\n\t"
            "ds_read_b32 %3, %5 offset:%11"
            : "=&v"(dst[0]), "=&v"(dst[1]), "=&v"(dst[2]), "=&v"(dst[3]), "+v"(va0), "+v"(va1)
            : "v"(c4), "v"(two), "n"((4 * S_) * 1024), "n"((4 * S_ + 1) * 1024), "n"((4 * S_ + 2) * 1024), "n"((4 * S_ + 3) * 1024)
            : "memory");
}
// the gathers of a ring slot have landed (at most 12 younger LDS operations outstanding); ties the registers to the wait
__device__ __forceinline__ half8 lp_landed(unsigned (&dst)[4]) {
    asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(dst[0]), "+v"(dst[1]), "+v"(dst[2]), "+v"(dst[3])::"memory");
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    return __builtin_bit_cast(half8, u32x4{dst[0], dst[1], dst[2], dst[3]});
}
// FG (round 6, "fast gathers"; FULLK, DS == 2, one copy): a codebook gather is ONE VALU instruction + one ds_read_b32 -- v_lshlrev_b32_sdwa
// moves (code byte u) << 2 into the low half of an address register whose high half holds the lane's table half, the table itself is
// the read's immediate offset (lp_layout) -- instead of the 2.3 instructions per gather hipcc makes of the C++ (v_bfe / v_and + shifts
// + v_lshl_add: 75 of the 108 VALU instructions of a 24-MFMA block, profiles/r6_pq_sweep_isa.txt).  The reads are issued from asm
// statements, so their waits are counted by hand: the operands of k-step s are complete when at most 12 LDS operations (the gathers of
// the three k-steps issued behind them) are outstanding; LDS returns in order, so operations the compiler issues in between only
// make that wait conservative, and its own waits (which do not count the asm reads) wait for more than they need, never for less.
//
// ABL (tools/pq_sweep_ablation.py; instantiated only in the variant library built with -DFAISS_AMD_LMF_ABLATE, results are WRONG): the
// FG sweep with units taken out, to see which of them the others wait for -- a MASK of
//     1  no codebook gathers (the A operands are the code bytes themselves, masked to finite halfs: same VALU count, no LDS reads)
//     2  the MFMAs of k-step 0 only (3 of 24 per block)
//     4  no epilogue (no score / maximum / threshold / parking code behind the MFMAs)
//     8  conflict-free gathers (the low five bits of every code byte replaced by the lane's row: same reads, no bank conflicts)
//    16  no code loads from memory (the next block's code bytes are computed from this block's)
//    32  the whole epilogue, but no score ever reaches a threshold (NaN thresholds)
//    64  no row-norm term (the scores are the accumulators)
//   128  parked candidates are dropped instead of flushed to memory
//   256  staged records are dropped instead of expanded
//   512  the flush's atomics go to words of their own (no two on one address: what the per-query counters' contention costs)
template <int METRIC, int MODE, int NQB, int DS, bool SEL, bool FULLK, bool TWOC, bool FG = false, int ABL = 0>
__global__ void __launch_bounds__(LP_THREADS, 2) ivf_lmf_pq_kernel(IvfLmParams p) {
    static_assert(ABL == 0 || (FG && MODE != MODE_DUMP), "ablations: the FG sweeps");
    static_assert(!TWOC || (FULLK && DS == 2), "the two-copy codebook serves PQ64 over d = 128");
    static_assert(!FG || (FULLK && DS == 2 && !TWOC), "fast gathers: PQ64 over d = 128, one copy");
    constexpr int PARK = TWOC ? LP_PARK2 : LP_PARK, NST = TWOC ? LP_NST2 : 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const int j = lane & 31;
    const int np = p.nprobe;
    const int M = p.M, dsub = p.dsub;
    const int nks = FULLK ? 8 : (p.d >> 4);
    const int gsh = __builtin_ctz((unsigned)p.gran_blocks); // blocks per granule: a power of two
    const LpLayout LY = lp_layout(p.d, M, TWOC, FG);
    const _Float16* cb = (const _Float16*)smem;
    if (FG) {
        // (the immediate offsets are absolute LDS addresses: the dynamic segment must start at 0, i.e. no static __shared__ here)
        if ((unsigned)(size_t)(const __attribute__((address_space(3))) char*)smem != 0u) __builtin_trap();
        const uint4* src = (const uint4*)p.pq16;
        for (int i = tid; i < 64 * 64; i += LP_THREADS) { // 64 tables of 64 x 16 bytes
            const int m = i >> 6;
            const int tb = ((m & 4) ? 65536 : 0) + (((m >> 3) << 2) + (m & 3)) * 1024;
            *(uint4*)(smem + tb + (i & 63) * 16) = src[i];
        }
    } else if (TWOC) {
        // [m][copy][256] entries of two halfs: copy 0 holds entry c at slot c, copy 1 at slot rotr8(c, 3)
        const uint32_t* src = (const uint32_t*)p.pq16;
        uint32_t* dst = (uint32_t*)smem;
        for (int e = tid; e < 64 * 256; e += LP_THREADS) {
            const uint32_t v = src[e];
            const int m = e >> 8;
            const unsigned c = (unsigned)e & 255u;
            dst[(m << 9) + (int)c] = v;
            dst[(m << 9) + 256 + (int)lmf_rotr8(c, 3)] = v;
        }
    } else {
        const uint4* src = (const uint4*)p.pq16;
        uint4* dst = (uint4*)smem;
        for (int i = tid; i < p.d * 32; i += LP_THREADS) dst[i] = src[i];
    }
    // (the upper 32 tables through a base of their own: 64 KB above the first, beyond the reach of a 16-bit offset.  Typed
    // LDS pointers: behind the opaque copy a generic pointer lost its address space and every gather became a flat load
    // with 64-bit address arithmetic -- + 36 % VALU instructions, profiles/r5f_pmc_two_copies_*)
    typedef const __attribute__((address_space(3))) _Float16* lds_half;
    typedef const __attribute__((address_space(3))) half2v* lds_half2;
    const lds_half cb3 = (lds_half)cb;
    lds_half cb_hi = cb3 + 32 * 512 * 2;
    asm volatile("" : "+v"(cb_hi));
    u64* pk_keys = (u64*)(smem + LY.off_park) + wave * PARK;
    uint32_t* pk_q = (uint32_t*)(smem + LY.off_park + 8 * PARK * 8) + wave * PARK;
    float* rn_lds; // |r^|^2 of the rows of the block in hand, per wave (see the flat kernel)
    if constexpr (FG) {
        rn_lds = (float*)(smem + LY.off_rn) + wave * 64;
    } else {
        __shared__ float rn_lds_all[8][64];
        rn_lds = rn_lds_all[wave];
    }
    int wcnt = 0;
    LmfStage<NST> st{smem + LY.off_stage + wave * LmfStage<NST>::BYTES, 0};
    auto flush = [&]() __attribute__((always_inline)) {
        if constexpr ((ABL & 128) != 0) wcnt = 0;
        else lmf_flush_parked<(ABL & 512) != 0>(p, lane, pk_keys, pk_q, wcnt);
    };
    auto expand = [&]() __attribute__((always_inline)) {
        if constexpr ((ABL & 256) != 0) st.cnt = 0;
        else lmf_stage_expand<METRIC, PARK, NST>(st, lane, pk_keys, pk_q, wcnt, flush);
    };
    __syncthreads();

    // operand-major code shadow (IvfLmParams::arena_cs): a block = npiece pieces of 64 lanes x cs_piece bytes
    constexpr int ND = LpCodes<DS, TWOC>::ND;
    const bool x4 = p.cs_piece == 16;
    const int npiece = (p.cs_bpl + p.cs_piece - 1) / p.cs_piece;
    const int64_t blk_bytes = TWOC ? (int64_t)kLmfChoiceBlockBytes : (int64_t)64 * npiece * p.cs_piece;
    const uint32_t it0 = p.item_bounds[1], it1 = p.item_bounds[2];
    LmfDraw draw(p.item_bounds, MODE == MODE_MIN ? 0 : 1, it0, it1);
    for (;;) {
        const uint32_t it = draw.next(lane);
        if (it >= it1) break;
        const IvfLmItem item = p.items[it];
        const int bk = __builtin_amdgcn_readfirstlane(item.bucket);
        const int qt = __builtin_amdgcn_readfirstlane(item.qt);
        const int rt = __builtin_amdgcn_readfirstlane(item.rt);
        const int list = bk >> 1;
        const int len = (int)p.list_len[list];
        const int64_t start = p.list_start[list];
        const uint32_t pb = p.bucket_start[bk];
        const int npair = min(32 * NQB, (int)(p.bucket_start[bk + 1 + item.both] - pb) - qt * (32 * NQB));
        const int r0 = rt * p.rows_per_item;
        const int r1 = min(len, r0 + p.rows_per_item);
        // sweep 1 may look at a prefix of the chunk only (see the flat kernel)
        const bool smp = MODE == MODE_MIN && p.sample_rows > 0 && p.sample_rows < p.rows_per_item;
        const int rend = smp ? min(r1, r0 + p.sample_rows) : r1;
        const int gbase = smp ? rt * (p.sample_rows >> (5 + gsh)) - ((r0 >> 5) >> gsh) : 0;

        // the code bytes of this lane's operands for block t: global -> registers, TWO blocks ahead (a 24-MFMA block is
        // shorter than a memory latency)
        unsigned cw[ND], cn[ND], cn2[ND];
#pragma unroll
        for (int i = 0; i < ND; ++i) cw[i] = cn[i] = cn2[i] = 0u;
        auto fetch = [&](int t, unsigned (&dst)[ND]) __attribute__((always_inline)) {
            if constexpr ((ABL & 16) != 0) {
#pragma unroll
                for (int i = 0; i < ND; ++i) dst[i] = cw[i] * 0x9E3779B1u + (unsigned)(t + lane);
                return;
            }
            const uint8_t* bp = p.arena_cs + ((start + t) >> 5) * blk_bytes;
            if (TWOC) { // 40 bytes per lane: two 16-byte pieces and one of 8
                const uint4 v0 = *(const uint4*)(bp + (int64_t)lane * 16), v1 = *(const uint4*)(bp + 1024 + (int64_t)lane * 16);
                const uint2 v2 = *(const uint2*)(bp + 2048 + (int64_t)lane * 8);
                dst[0] = v0.x, dst[1] = v0.y, dst[2] = v0.z, dst[3] = v0.w;
                dst[4] = v1.x, dst[5] = v1.y, dst[6] = v1.z, dst[7] = v1.w;
                dst[8 % ND] = v2.x, dst[9 % ND] = v2.y;
            } else if (FULLK && DS == 2) { // two 16-byte pieces, unconditionally
                const uint4 v0 = *(const uint4*)(bp + (int64_t)lane * 16), v1 = *(const uint4*)(bp + ((int64_t)64 + lane) * 16);
                dst[0] = v0.x, dst[1] = v0.y, dst[2] = v0.z, dst[3] = v0.w;
                dst[4] = v1.x, dst[5] = v1.y, dst[6] = v1.z, dst[7] = v1.w;
            } else if (x4) {
#pragma unroll
                for (int c = 0; c < ND / 4; ++c) {
                    if (c < npiece) {
                        const uint4 v = *(const uint4*)(bp + ((int64_t)c * 64 + lane) * 16);
                        dst[4 * c] = v.x, dst[4 * c + 1] = v.y, dst[4 * c + 2] = v.z, dst[4 * c + 3] = v.w;
                    }
                }
            } else {
#pragma unroll
                for (int c = 0; c < ND; ++c)
                    if (c < npiece) dst[c] = *(const unsigned*)(bp + ((int64_t)c * 64 + lane) * 4);
            }
        };
        const int bstep = MODE == MODE_MIN ? LP_BR * p.min_stride : LP_BR; // (sweep 1 may sample the blocks, see the flat kernel)
        const int tlast = (r1 - 1) & ~31; // (prefetches behind the last block re-read it: loads stay unconditional)
        fetch(r0, cw);
        fetch(min(r0 + bstep, tlast), cn);
        // row norms: register ring -> LDS slice -> broadcast reads, as in the flat kernel
        // (one value per lane and block: lane l fetches row l & 31 of the block four blocks ahead -- both halves the same)
        auto rn_fetch = [&](int blk) __attribute__((always_inline)) -> float {
            const int row = r0 + blk * bstep + (lane & 31);
            if (METRIC != METRIC_L2) return 0.f;
            const float v = p.arena_rn[start + min(row, r1 - 1)];
            return row < r1 ? v : INFINITY; // (rows behind the chunk: scores of -inf without a test in the epilogue)
        };
        float pf0 = rn_fetch(0), pf1 = rn_fetch(1), pf2 = rn_fetch(2), pf3 = rn_fetch(3);

        // The item's query blocks that hold queries (1 .. NQB): the block loop is instantiated per count, so that a list probed
        // by <= 32 / <= 64 of the batch's queries costs 8 / 16 MFMAs and one / two epilogues per 32-row block instead of 24 and
        // three -- chosen per ITEM, outside the loop (a bound inside it costs the compiler its count of the loads in flight).
        auto run_item = [&](auto nb_c) __attribute__((always_inline)) {
        constexpr int NB = decltype(nb_c)::value;
        // ---- this lane's queries: B operands = fp16 of the residual query (L2) / of the query (inner product), prepared
        // once per search (lmf_pq_prepare_kernel)
        LmfLane L[NB];
        half8 bq[NB][8];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int my = b * 32 + j;
            L[b].qv = my < npair;
            const uint32_t pi = p.pairs[pb + (uint32_t)(qt * (32 * NQB)) + (uint32_t)(L[b].qv ? my : 0)];
            const int q = (int)(pi / (uint32_t)np);
            const int pr = (int)(pi - (uint32_t)q * (uint32_t)np);
            const _Float16* qrow = (const _Float16*)p.pair16 + (int64_t)(METRIC == METRIC_L2 ? pi : (uint32_t)q) * p.d + 8 * h;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                if (FULLK || s < nks) bq[b][s] = *(const half8*)(qrow + 16 * s);
                else bq[b][s] = half8{0, 0, 0, 0, 0, 0, 0, 0};
            }
            L[b].xh = METRIC == METRIC_L2 ? p.pair_xh[pi] : p.coarse_dis[pi];
            L[b].base_pos = p.prefix[(int64_t)q * (np + 1) + pr];
            L[b].qpr = ((uint32_t)q << 11) | (uint32_t)pr;
            L[b].tq = __builtin_nanf(""); // (see the flat kernel)
            L[b].gm = -INFINITY;
            L[b].gq = nullptr;
            L[b].kq = nullptr;
            if (MODE == MODE_MIN) L[b].gq = p.gmin + (int64_t)q * p.gstride + p.prefixg[(int64_t)q * (np + 1) + pr] + h;
            if (MODE == MODE_COLLECT && L[b].qv) L[b].tq = lmf_collect_threshold<METRIC>(p.thr_f[q], L[b].xh);
            if ((ABL & 32) != 0) L[b].tq = __builtin_nanf("");
            if (MODE == MODE_DUMP) L[b].kq = p.keys + (int64_t)q * p.stride + L[b].base_pos;
        }

        // the A operand of k-step s: coordinates 16 s + 8 h .. + 7 of this lane's row, gathered from the codebook by
        // the code bytes in cw
        auto operand_of = [&](const unsigned (&cw)[ND], int s_) __attribute__((always_inline)) -> half8 {
            half8 a = half8{0, 0, 0, 0, 0, 0, 0, 0};
            if (!FULLK && s_ >= nks) return a;
            const int kb = 16 * s_ + 8 * h; // first coordinate
            if (DS == 8) {
                const int m = kb / dsub, off = kb - m * dsub;
                const unsigned c = (cw[s_ >> 2] >> (8 * (s_ & 3))) & 255u;
                a = *(const half8*)(cb + ((m << 8) + (int)c) * dsub + off);
            } else if (DS == 4) {
                const unsigned c2 = (cw[s_ >> 1] >> (16 * (s_ & 1))) & 0xffffu;
                const int m0 = kb >> 2;
                const half4v lo = *(const half4v*)(cb + ((m0 << 8) + (int)(c2 & 255u)) * 4);
                const half4v hi = *(const half4v*)(cb + (((m0 + 1) << 8) + (int)(c2 >> 8)) * 4);
                a = half8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            } else if (TWOC) {
                // the 40 bits of k-step s_ (bytes 5 s_ .. 5 s_ + 4 of the lane's string): four 9-bit fields copy << 8 | slot
                // (32-bit pieces by hand: a 64-bit shift of two array elements sent the array to scratch memory)
                const int o = 5 * s_, i = o >> 2, r = o & 3, i4 = (o + 4) >> 2, r4 = (o + 4) & 3;
                const unsigned lo32 = r ? __builtin_amdgcn_alignbyte(cw[(i + 1) % ND], cw[i % ND], (unsigned)r) : cw[i % ND];
                const unsigned hi4 = (cw[i4 % ND] >> (8 * r4)) & 15u;
                const unsigned fields[4] = {lo32 & 511u, (lo32 >> 9) & 511u, (lo32 >> 18) & 511u, (lo32 >> 27) | (hi4 << 5)};
                const int m0 = kb >> 1; // sub-quantizers m0 .. m0 + 3: all below 32 or all from 32 on
                const lds_half bm = m0 < 32 ? cb3 : cb_hi;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned field = fields[u];
                    const half2v e2 = *(lds_half2)(bm + ((((m0 + u) & 31) << 9) + (int)field) * 2);
                    a[2 * u] = e2[0];
                    a[2 * u + 1] = e2[1];
                }
            } else if (DS == 2) {
                const unsigned c4 = cw[s_];
                const int m0 = kb >> 1;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const half2v v = *(const half2v*)(cb + (((m0 + u) << 8) + (int)((c4 >> (8 * u)) & 255u)) * 2);
                    a[2 * u] = v[0];
                    a[2 * u + 1] = v[1];
                }
            } else {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const unsigned c = (cw[2 * s_ + (u >> 2)] >> (8 * (u & 3))) & 255u;
                    a[u] = cb[((kb + u) << 8) + (int)c];
                }
            }
            return a;
        };
        int bi = 0;
        half8 av[4]; // ring of decoded A operands (k-step s in slot s % 4)
        // FG: the four dwords of a ring slot are written by asm ds_reads; va0 / va1 = address registers (high half: the lane's table half)
        unsigned ar[4][4];
        unsigned va0 = (unsigned)h << 16, va1 = (unsigned)h << 16;
        const unsigned two = 2u;
        // (ABL 8: code bytes whose low five bits are the lane's row -> 32 distinct banks per half wavefront)
        auto abl_code = [&](unsigned c4) __attribute__((always_inline)) -> unsigned {
            return (ABL & 8) ? (c4 & 0xE0E0E0E0u) | ((unsigned)j * 0x01010101u) : c4;
        };
        if constexpr (FG && !(ABL & 1)) {
            lp_gather<0>(abl_code(cw[0]), ar[0], va0, va1, two);
            lp_gather<1>(abl_code(cw[1]), ar[1], va0, va1, two);
            lp_gather<2>(abl_code(cw[2]), ar[2], va0, va1, two);
        } else if constexpr (!FG) {
#pragma unroll
            for (int s = 0; s < LP_AHEAD; ++s) av[s] = operand_of(cw, s);
        }
        for (int t = r0; t < rend; t += bstep, ++bi) {
            if (METRIC == METRIC_L2) {
                asm volatile("" ::: "memory");
                __builtin_amdgcn_wave_barrier(); // (the epilogue reads of the previous block were issued: LDS keeps a wave's order)
                rn_lds[lane] = pf0;              // (both halves write the block's 32 values: rn_lds[l] == rn_lds[32 + l])
                pf0 = pf1, pf1 = pf2, pf2 = pf3;
                pf3 = rn_fetch(bi + 4);
                asm volatile("" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
            const int row_b = t + 4 * h;
            const bool tail = t + 32 > r1;
            const uint32_t mw = SEL ? p.sel_mask[(start + t) >> 5] >> (4 * h) : 0u; // IDSelector bits of the block's rows
            f32x16 acc[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
            f32x4 rn[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) rn[g] = f32x4{0.f, 0.f, 0.f, 0.f};
            // software pipeline: gathers of k-step s + LP_AHEAD | MFMAs of k-step s (one k-step ahead the gathers of a step
            // had only the 3 MFMAs of the step before -- ~100 cycles -- to come back from an LDS all eight wavefronts gather
            // from).  The pipeline runs ACROSS blocks: the last k-steps of a block gather the first operands of the next
            // block looked at (its code bytes wait in cn), so that a block does not open with LP_AHEAD exposed LDS latencies
            // and the epilogue runs with gathers in flight.
            auto step = [&](auto s_c) __attribute__((always_inline)) {
                constexpr int s = decltype(s_c)::value;
                if constexpr (FG && !(ABL & 1)) {
                    constexpr int sn = (s + LP_AHEAD) & 7;
                    lp_gather<sn>(abl_code(s + LP_AHEAD < 8 ? cw[sn] : cn[sn]), ar[(s + LP_AHEAD) % 4], va0, va1, two);
                } else if constexpr (!FG) {
                    if (s + LP_AHEAD < 8) av[(s + LP_AHEAD) % 4] = operand_of(cw, s + LP_AHEAD);
                    else av[(s + LP_AHEAD) % 4] = operand_of(cn, s + LP_AHEAD - 8);
                }
                if (s == 1) fetch(min(t + 2 * bstep, tlast), cn2);
                if (s == 5 && METRIC == METRIC_L2) { // |r^|^2 of the block's rows (lane: rows 8 g + 4 h + e) from the wave's slice
#pragma unroll
                    for (int g = 0; g < 4; ++g) rn[g] = *(const f32x4*)(rn_lds + 8 * g + 4 * h);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (FULLK || s < nks) {
                    half8 a;
                    if constexpr ((ABL & 1) != 0) {
                        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                        constexpr unsigned fin = 0x3bff3bffu; // (halfs below 1)
                        a = __builtin_bit_cast(half8, u32x4{cw[s] & fin, cw[(s + 1) & 7] & fin, cw[(s + 2) & 7] & fin, cw[(s + 3) & 7] & fin});
                    } else if constexpr (FG) a = lp_landed(ar[s % 4]);
                    else a = av[s % 4];
                    if (!(ABL & 2) || s == 0) {
#pragma unroll
                        for (int b = 0; b < NB; ++b)
                            acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bq[b][s], acc[b], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            step(std::integral_constant<int, 0>{});
            step(std::integral_constant<int, 1>{});
            step(std::integral_constant<int, 2>{});
            step(std::integral_constant<int, 3>{});
            step(std::integral_constant<int, 4>{});
            step(std::integral_constant<int, 5>{});
            step(std::integral_constant<int, 6>{});
            step(std::integral_constant<int, 7>{});
            if constexpr ((ABL & 4) != 0) { // (the accumulators stay alive through one instruction; the store below never happens)
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    L[b].gm = lmf_max3(L[b].gm, acc[b][0], acc[b][15]);
                    if (t + bstep >= rend && L[b].gm == 1.2345f) p.cnt[0] = 1u;
                }
            } else if constexpr (MODE == MODE_MIN) {
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    lmf_scores<METRIC, SEL>(acc[b], rn, tail, row_b, r1, mw);
                    {
                        const float lm = lmf_lane_max(acc[b]);
                        L[b].gm = lmf_max3(L[b].gm, lm, lm); // (no canonicalising v_max x, x around it)
                    }
                }
                const int blk = t >> 5;
                if ((((t + bstep) >> 5) >> gsh) != (blk >> gsh) || t + bstep >= rend) {
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        if (L[b].qv) L[b].gq[2 * ((blk >> gsh) + gbase)] = ordkey<METRIC>(lmf_to_est<METRIC>(L[b].gm + L[b].xh));
                        L[b].gm = -INFINITY;
                    }
                }
            } else if constexpr (MODE == MODE_DUMP) {
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    lmf_scores<METRIC, SEL>(acc[b], rn, tail, row_b, r1, mw);
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int rowl = row_b + 8 * g + e;
                            if (L[b].qv && rowl < r1)
                                L[b].kq[rowl] = ((u64)ordkey<METRIC>(lmf_to_est<METRIC>(acc[b][4 * g + e] + L[b].xh)) << 32) |
                                                (u64)(L[b].base_pos + (uint32_t)rowl);
                        }
                }
            } else {
                // ---- lanes whose best score of query block b reaches their query's threshold stage their 16 scores (the scores of
                // all query blocks first: the row norms are dead by the time a dense pass needs registers)
                bool hit[NB];
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    if constexpr (!(ABL & 64)) lmf_scores<METRIC, SEL>(acc[b], rn, tail, row_b, r1, mw);
                    hit[b] = lmf_lane_max(acc[b]) >= L[b].tq;
                }
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    lmf_collect_pair(st, lane, hit[b], acc[b], L[b].tq, L[b].base_pos + (uint32_t)row_b, L[b].qpr, L[b].xh, expand);
            }
#pragma unroll
            for (int i = 0; i < ND; ++i) cw[i] = cn[i], cn[i] = cn2[i];
        }
        };
        if (npair > 64) run_item(std::integral_constant<int, 3>{});
        else if (npair > 32) run_item(std::integral_constant<int, 2>{});
        else run_item(std::integral_constant<int, 1>{});
    }
    if (MODE == MODE_COLLECT) {
        if (st.cnt > 0) expand();
        if (wcnt > 0) flush();
    }
}

// ------------------------------------------------------------------ launchers of the sweeps
int ivf_lmf_grid_blocks(const IvfLmParams& p, int num_cus) {
    if (p.kind == 1 && !p.lmf_pairb) return num_cus; // one 8-wave workgroup per CU (codebook + slices in its LDS)
    return 2 * num_cus / 8 * 8;            // IVFFlat: two 4-wave workgroups per CU
}
template <int METRIC, int MODE, bool SEL, bool PAIRB>
static void lmf_flat_launch(const IvfLmParams& p, int grid_blocks, hipStream_t stream) {
    const int lds = MODE == MODE_COLLECT ? LF_LDS : 0;
    if constexpr (MODE != MODE_DUMP && !SEL) {
        // two-wave workgroups in lock-step over sibling items (round 6): twice the workgroups.  lmf_pair 1: sweep 1 only (the
        // default: measured faster there, slower in sweep 2 -- DESIGN 3.12), 2: both sweeps
        if ((p.lmf_pair == 2 || (p.lmf_pair == 1 && MODE == MODE_MIN)) && p.ldh == 128) {
            auto kern = ivf_lmf_flat_kernel<METRIC, MODE, kLmfQueryBlocks, 8, true, SEL, PAIRB, true>;
            HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LF_LDS / 2));
            hipLaunchKernelGGL(kern, dim3((unsigned)(2 * grid_blocks)), dim3(128), MODE == MODE_COLLECT ? LF_LDS / 2 : 0, stream, p);
            return;
        }
    }
#define FA_LF(NQB_, KS_, FULL_)                                                                                                \
    do {                                                                                                                       \
        HIP_CHECK(hipFuncSetAttribute((const void*)ivf_lmf_flat_kernel<METRIC, MODE, NQB_, KS_, FULL_, SEL, PAIRB>,            \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, LF_LDS));                                    \
        hipLaunchKernelGGL((ivf_lmf_flat_kernel<METRIC, MODE, NQB_, KS_, FULL_, SEL, PAIRB>), dim3((unsigned)grid_blocks),     \
                           dim3(LF_THREADS), lds, stream, p);                                                                  \
    } while (0)
    if (p.ldh == 128) FA_LF(kLmfQueryBlocks, 8, true);
    else if (p.ldh < 128) FA_LF(kLmfQueryBlocks, 8, false);
    else if (p.ldh == 256) FA_LF(2, 16, true);
    else if (p.ldh == 384) FA_LF(1, 24, true);
    else FA_LF(1, 32, true);
#undef FA_LF
}
template <int METRIC, int MODE, bool SEL>
static void lmf_pq_launch(const IvfLmParams& p, int grid_blocks, hipStream_t stream) {
    constexpr int NQB = kLmfQueryBlocks;
    const bool twoc = p.cs_choice != 0;
    const int ds = p.dsub >= 8 ? 8 : p.dsub;
    const bool bench_shape = !twoc && ds == 2 && p.d == 128 && p.M == 64 && p.cs_piece == 16;
    const bool fastg = bench_shape && p.lmf_fast_gather != 0;
    const int lds = lp_layout(p.d, p.M, twoc, fastg).total;
#define FA_LP(DS_, FK_, TC_, ...)                                                                                              \
    do {                                                                                                                       \
        HIP_CHECK(hipFuncSetAttribute((const void*)ivf_lmf_pq_kernel<METRIC, MODE, NQB, DS_, SEL, FK_, TC_, ##__VA_ARGS__>,    \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds));                                       \
        hipLaunchKernelGGL((ivf_lmf_pq_kernel<METRIC, MODE, NQB, DS_, SEL, FK_, TC_, ##__VA_ARGS__>), dim3((unsigned)grid_blocks), \
                           dim3(LP_THREADS), lds, stream, p);                                                                  \
    } while (0)
    if (twoc) {
        FA_THROW_IF_NOT(ds == 2 && ivf_lmf_choice_shape(p.d, p.M));
        FA_LP(2, true, true); // PQ64 over d = 128 with the two-copy codebook
    }
#ifdef FAISS_AMD_LMF_ABLATE
    else if (const char* abl = fastg && METRIC == METRIC_L2 && !SEL && MODE != MODE_DUMP ? experiment_env("FAISS_AMD_LMF_ABLATE") : nullptr;
             abl && atoi(abl) >= 1) {
        if constexpr (METRIC == METRIC_L2 && !SEL && MODE != MODE_DUMP) {
            switch (atoi(abl)) {
            case 1: FA_LP(2, true, false, true, 1); break;
            case 2: FA_LP(2, true, false, true, 2); break;
            case 4: FA_LP(2, true, false, true, 4); break;
            case 8: FA_LP(2, true, false, true, 8); break;
            case 16: FA_LP(2, true, false, true, 16); break;
            case 32: FA_LP(2, true, false, true, 32); break;
            case 64: FA_LP(2, true, false, true, 64); break;
            case 128: FA_LP(2, true, false, true, 128); break;
            case 256: FA_LP(2, true, false, true, 256); break;
            case 5: FA_LP(2, true, false, true, 5); break;
            case 6: FA_LP(2, true, false, true, 6); break;
            case 20: FA_LP(2, true, false, true, 20); break;
            case 7: FA_LP(2, true, false, true, 7); break;
            case 21: FA_LP(2, true, false, true, 21); break;
            case 22: FA_LP(2, true, false, true, 22); break;
            case 23: FA_LP(2, true, false, true, 23); break;
            case 33: FA_LP(2, true, false, true, 33); break;
            case 34: FA_LP(2, true, false, true, 34); break;
            case 48: FA_LP(2, true, false, true, 48); break;
            case 96: FA_LP(2, true, false, true, 96); break;
            case 512: FA_LP(2, true, false, true, 512); break;
            default: FA_THROW_IF_NOT(!"FAISS_AMD_LMF_ABLATE: a mask that is not instantiated");
            }
        }
    }
#endif
    else if (fastg) FA_LP(2, true, false, true); // PQ64 over d = 128, one copy, one-instruction gathers (round 6)
    else if (bench_shape) FA_LP(2, true, false); // the same shape, gathers as hipcc compiles them
    else if (ds == 1) FA_LP(1, false, false);
    else if (ds == 2) FA_LP(2, false, false);
    else if (ds == 4) FA_LP(4, false, false);
    else FA_LP(8, false, false);
#undef FA_LP
}
template <int METRIC, bool SEL>
static void lmf_launch_sel(const IvfLmParams& p, int mode, int grid_blocks, hipStream_t stream) {
    if (p.kind == 0) {
        if (mode == MODE_MIN) lmf_flat_launch<METRIC, MODE_MIN, SEL, false>(p, grid_blocks, stream);
        else if (mode == MODE_COLLECT) lmf_flat_launch<METRIC, MODE_COLLECT, SEL, false>(p, grid_blocks, stream);
        else lmf_flat_launch<METRIC, MODE_DUMP, SEL, false>(p, grid_blocks, stream);
    } else if (p.kind == 2 || p.lmf_pairb) {
        if (mode == MODE_MIN) lmf_flat_launch<METRIC, MODE_MIN, SEL, true>(p, grid_blocks, stream);
        else if (mode == MODE_COLLECT) lmf_flat_launch<METRIC, MODE_COLLECT, SEL, true>(p, grid_blocks, stream);
        else lmf_flat_launch<METRIC, MODE_DUMP, SEL, true>(p, grid_blocks, stream);
    } else {
        if (mode == MODE_MIN) lmf_pq_launch<METRIC, MODE_MIN, SEL>(p, grid_blocks, stream);
        else if (mode == MODE_COLLECT) lmf_pq_launch<METRIC, MODE_COLLECT, SEL>(p, grid_blocks, stream);
        else lmf_pq_launch<METRIC, MODE_DUMP, SEL>(p, grid_blocks, stream);
    }
}
template <int METRIC>
static void lmf_launch_mode(const IvfLmParams& p, int mode, int grid_blocks, hipStream_t stream) {
    // (the test dump never runs with a selector: one instantiation less)
    if (p.sel_mask && mode != MODE_DUMP) lmf_launch_sel<METRIC, true>(p, mode, grid_blocks, stream);
    else lmf_launch_sel<METRIC, false>(p, mode, grid_blocks, stream);
}
void launch_ivf_lmf_sweep(const IvfLmParams& p, int mode, int grid_blocks, hipStream_t stream) {
    if (p.nq == 0) return;
    const int skind = p.lmf_pairb ? 2 : p.kind; // the sweeps' flavour: decoded IVFPQ residuals run the pair-operand IVFFlat kernel
    FA_THROW_IF_NOT(p.filter && mode >= 1 && mode <= 3 && grid_blocks > 0);
    FA_THROW_IF_NOT(p.lmf_pairb && p.kind == 1 ? ivf_lmf_pq_decoded_supported(p.d, p.dpad, p.M)
                                               : ivf_lmf_supported(p.kind, p.d, p.dpad, p.kind == 2 ? p.sq_ct : p.M));
    FA_THROW_IF_NOT(p.min_stride >= 1 && p.min_stride <= 8);
    FA_THROW_IF_NOT(p.qpi == ivf_lmf_queries_per_item(skind, p.d) && p.nq < (1 << 21) && p.nprobe <= 2048 &&
                    p.gran_blocks >= 1 && (p.gran_blocks & (p.gran_blocks - 1)) == 0);
    if (skind == 2) {
        FA_THROW_IF_NOT(p.pair16 && p.pair_xh && p.arena_h && p.ldh == ivf_lmf_row_halfs(p.d) && p.ldh <= 512 && p.ldq16 == p.ldh);
        FA_THROW_IF_NOT(p.metric != METRIC_L2 || p.arena_rn);
    } else if (p.kind == 0) {
        FA_THROW_IF_NOT(p.xq16 && p.arena_h && p.ldh == ivf_lmf_row_halfs(p.d) && p.ldh <= 512 && p.ldq16 >= p.ldh &&
                        p.ldq16 % 8 == 0);
        FA_THROW_IF_NOT(p.metric != METRIC_L2 || (p.arena_rn && p.xqn));
    } else {
        FA_THROW_IF_NOT(p.pq16 && p.arena_cs && p.cs_bpl > 0 && (p.cs_piece == 4 || p.cs_piece == 16) && p.centroids &&
                        p.ldq % 4 == 0 && p.ldc % 4 == 0);
        FA_THROW_IF_NOT(lp_layout(p.d, p.M, p.cs_choice != 0).total + 2048 <= 160 * 1024 && (p.metric != METRIC_L2 || p.arena_rn));
    }
    if (p.metric == METRIC_L2) lmf_launch_mode<METRIC_L2>(p, mode, grid_blocks, stream);
    else lmf_launch_mode<METRIC_INNER_PRODUCT>(p, mode, grid_blocks, stream);
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------ bound: k-th best granule estimate + error band
// One workgroup per query: radix select (8 bits per pass) over the query's granule slots.
template <int METRIC>
__global__ void __launch_bounds__(256) lmf_bound_kernel(IvfLmParams p, const float* __restrict__ xn_bound) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t sel_prefix, sel_need;
    const int q = blockIdx.x;
    const int tid = threadIdx.x;
    const int np = p.nprobe;
    const uint32_t S = p.prefixg[(int64_t)q * (np + 1) + np];
    const uint32_t* g = p.gmin + (int64_t)q * p.gstride;
    // (IVFPQ: the B operands are fp16 (q - c): every coordinate is below sqrt(max |q - c|^2), which must stay in range)
    if ((p.qflags && p.qflags[q]) || (p.coarse_bad && p.coarse_bad[q]) || (p.kind != 0 && !(xn_bound[q] <= 9.0e8f))) {
        // outside the fp16 range / NaN / no coarse assignment yet: nothing is collected, the query is redone by the query-major scan
        if (tid == 0) {
            p.thr_f[q] = METRIC == METRIC_L2 ? -INFINITY : INFINITY;
            const uint32_t s = atomicAdd(&p.ovf[0], 1u);
            p.ovf[1 + s] = (uint32_t)q;
        }
        return;
    }
    if (tid == 0 && p.err_f) p.err_f[q] = INFINITY; // (no band known yet: the tightening keeps every candidate)
    if (S < (uint32_t)p.k) { // fewer granules than results: everything is a candidate
        if (tid == 0) p.thr_f[q] = lmf_worst<METRIC>();
        return;
    }
    // The estimates of one query share sign and exponent and most of them the leading mantissa bits: a digit of the raw key
    // puts every slot into one or two bins (256 threads queueing on the same LDS word) and the top pass decides nothing.  So
    // the digits are those of key - min over the bits in which the query's valid keys differ at all: ceil(bits / 8) passes
    // (three for the usual 2^22 .. 2^24 spread) over evenly filled bins.  Empty slots (0xffffffff) take no part: with fewer
    // than k valid slots the k-th best is an empty one.
    __shared__ uint32_t red_min[4], red_max[4], red_cnt[4];
    // Up to 2048 slots live in registers (eight per thread, loaded together): the passes below then cost no memory round trips
    // (a loop of dependent L2 reads per pass -- six rounds at nb = 10M, four passes -- was most of the kernel's 0.09 ms).
    constexpr int BR = 8;
    const bool inreg = S <= 256u * BR; // (workgroup-uniform)
    uint32_t vals[BR];
#pragma unroll
    for (int u = 0; u < BR; ++u) {
        const uint32_t i = (uint32_t)tid + 256u * u;
        vals[u] = inreg && i < S ? g[i] : 0xffffffffu;
    }
    auto for_slots = [&](auto fn) __attribute__((always_inline)) {
        if (inreg) {
#pragma unroll
            for (int u = 0; u < BR; ++u) fn(vals[u]);
        } else {
            for (uint32_t i = tid; i < S; i += 256) fn(g[i]);
        }
    };
    uint32_t vmin = 0xffffffffu, vmax = 0u, nval = 0u;
    for_slots([&](uint32_t v) {
        if (v < kInvalidOrdKey) {
            vmin = min(vmin, v);
            vmax = max(vmax, v);
            ++nval;
        }
    });
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        vmin = min(vmin, (uint32_t)__shfl_xor((int)vmin, off, 64));
        vmax = max(vmax, (uint32_t)__shfl_xor((int)vmax, off, 64));
        nval += (uint32_t)__shfl_xor((int)nval, off, 64);
    }
    if ((tid & 63) == 0) red_min[tid >> 6] = vmin, red_max[tid >> 6] = vmax, red_cnt[tid >> 6] = nval;
    if (tid == 0) {
        sel_prefix = 0u;
        sel_need = (uint32_t)p.k;
    }
    __syncthreads();
    vmin = min(min(red_min[0], red_min[1]), min(red_min[2], red_min[3]));
    vmax = max(max(red_max[0], red_max[1]), max(red_max[2], red_max[3]));
    nval = red_cnt[0] + red_cnt[1] + red_cnt[2] + red_cnt[3];
    const bool enough = nval >= (uint32_t)p.k; // (workgroup-uniform)
    const uint32_t range = enough ? vmax - vmin : 0u;
    const int npass = range == 0u ? 0 : (32 - __builtin_clz(range) + 7) >> 3;
    for (int pass = npass - 1; pass >= 0; --pass) {
        hist[tid] = 0u;
        __syncthreads();
        const uint32_t pre = sel_prefix;
        for_slots([&](uint32_t v) {
            const uint32_t w = v - vmin;
            const bool in = v < kInvalidOrdKey && (pass == 3 || (w >> (8 * (pass + 1))) == pre);
            if (in) atomicAdd(&hist[(w >> (8 * pass)) & 255u], 1u);
        });
        __syncthreads();
        if (tid < 64) {
            // bucket holding the sel_need-th smallest: exclusive prefix over 256 bins, 4 bins per lane
            uint32_t c[4], sum = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                c[u] = hist[4 * tid + u];
                sum += c[u];
            }
            const uint32_t inc = wave_incl_scan(sum);
            uint32_t before = inc - sum;
            const uint32_t need = sel_need;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (before < need && need <= before + c[u]) {
                    sel_prefix = (pre << 8) | (uint32_t)(4 * tid + u);
                    sel_need = need - before;
                }
                before += c[u];
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        const uint32_t tk = enough ? vmin + sel_prefix : 0xffffffffu;
        float thr;
        if (tk >= kInvalidOrdKey) {
            thr = lmf_worst<METRIC>();
        } else {
            const float T = unordkey<METRIC>(tk);
            float extra = 0.f;
            if (p.kind == 1) {
                // IVFPQ: the exact path's table grid (M entries rounded to delta <= 2^-23 sum_m max_c |<q_m, cb_mc>|), its
                // per-row term |r^|^2 + 2 <c, r^> and the coarse term, all below (|q| + |c| + |r^|)^2 in magnitude
                const float sroot = sqrtf(p.xn_full[q]) + sqrtf(p.cn_max) + sqrtf(p.yn_max);
                // + the fp32 chains of those terms themselves (d products each: d 2^-24 |c| |r^| and the like), ADVICE r4
                extra = (4.8e-7f * (float)p.M + 1.2e-7f * (float)(p.d + 8)) * sroot * sroot;
            }
            if (p.kind == 2) {
                // scalar quantizer: xn_bound / yn_max are the norms of the matrix pipe's operands (a o s and the centred codes);
                // on top, the fp32 chains of |a|^2 and |s o code'|^2, the exact path's own chains over d dimensions and the
                // one rounding of b' = b + mid s per dimension, all in terms of the magnitudes they act on
                const float an = p.an_bound[q], sa = sqrtf(an), sr = sqrtf(p.rn_max), sb = sqrtf(p.bn);
                if (METRIC == METRIC_L2) extra = 2.4e-7f * (float)(p.d + 8) * (an + p.rn_max) + 4.8e-7f * sb * (sa + sr);
                else extra = 2.4e-7f * (float)(p.d + 8) * (sa * sb + 2.f * sqrtf(xn_bound[q]) * (sqrtf(p.yn_max) + sqrtf(p.cmid2)));
            }
            // (kind 2: the operands' norms are in code units -- d 127.5^2 for 8-bit codes -- and say nothing about the magnitudes
            // the fp32 chains act on: the generic bound's norm terms are replaced by `extra` above)
            const float E = p.kind == 2 ? 1.25f * ((METRIC == METRIC_L2 ? 2.f : 1.f) * ivf_filter_err_mfma(p.d, xn_bound[q], p.yn_max) + extra) + 1e-30f
                                        : ivf_filter_err_bound(METRIC, p.d, xn_bound[q], p.yn_max, extra);
            if (p.band_out) p.band_out[q] = E;
            if (p.err_f) p.err_f[q] = E;
            thr = METRIC == METRIC_L2 ? T + 2.f * E : T - 2.f * E;
            if (thr != thr) thr = lmf_worst<METRIC>();
        }
        p.thr_f[q] = thr;
    }
}
void launch_ivf_lmf_bound(const IvfLmParams& p, const float* xn_bound, hipStream_t stream) {
    if (p.nq == 0) return;
    if (!p.pre_cleared) HIP_CHECK(hipMemsetAsync(p.ovf, 0, 4, stream));
    if (p.metric == METRIC_L2) hipLaunchKernelGGL(lmf_bound_kernel<METRIC_L2>, dim3((unsigned)p.nq), dim3(256), 0, stream, p, xn_bound);
    else hipLaunchKernelGGL(lmf_bound_kernel<METRIC_INNER_PRODUCT>, dim3((unsigned)p.nq), dim3(256), 0, stream, p, xn_bound);
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------ tighten: the smallest superset the band allows
// One workgroup per query, behind sweep 2.  The collected candidates of a query are ALL its rows with estimate <= thr_f, and
// thr_f >= (k-th best estimate of the query) by construction, so the k-th best estimate among them, T2, is the k-th best
// estimate of all the query's rows; a row of the exact top-k has estimate <= T2 + 2 E_q (the superset
// argument with the sharpest T).  Candidates above that leave the segment (compacted in place through LDS) before
// anything exact is computed: the rerank then works on ~ k + (rows inside the band) candidates whatever sample of the rows
// the first sweep's bound came from.  Also does what lm_clamp_kernel does for the other scans (cnt > stride: clamp + ovf).
constexpr int LT_CAP = 2048; // kept candidates a workgroup can stage (24 KB of LDS); more: the segment stays as it is
template <int METRIC>
__global__ void __launch_bounds__(256) lmf_tighten_kernel(IvfLmParams p, int fin_cap) {
    __shared__ u64 keep_k[LT_CAP];
    __shared__ uint16_t keep_p[LT_CAP];
    __shared__ uint32_t hist[256];
    __shared__ uint32_t sel_prefix, sel_need, nkeep;
    const int q = blockIdx.x;
    const int tid = threadIdx.x;
    const uint32_t raw = p.cnt[q];
    // (queries the bound kernel handed to the redo path -- threshold "nothing" -- are listed already)
    const float thr0 = p.thr_f[q];
    const bool listed = METRIC == METRIC_L2 ? thr0 == -INFINITY : thr0 == INFINITY;
    auto overflow = [&]() {
        if (listed) return;
        const uint32_t s = atomicAdd(&p.ovf[0], 1u);
        p.ovf[1 + s] = (uint32_t)q;
    };
    if ((int64_t)raw > p.stride) { // (workgroup-uniform) the segment overflowed: redone by the caller
        if (tid == 0) {
            p.cnt[q] = (uint32_t)p.stride;
            overflow();
        }
        return;
    }
    const uint32_t n = raw;
    u64* kq = p.keys + (int64_t)q * p.stride;
    uint16_t* cpr = p.cand_pr + (int64_t)q * p.stride;
    const float E = p.err_f[q];
    if (n > (uint32_t)p.k && E < INFINITY) {
        // (as in lmf_bound_kernel: up to 2048 candidates wait in registers -- one round of loads for the selection passes and the
        // compaction --, and the digits are those of key - min over the bits in which the query's estimate keys differ)
        constexpr int BR = 8;
        const bool inreg = n <= 256u * BR; // (workgroup-uniform)
        u64 kv[BR];
        uint16_t pv[BR];
#pragma unroll
        for (int u = 0; u < BR; ++u) {
            const uint32_t i = (uint32_t)tid + 256u * u;
            const bool ok = inreg && i < n;
            kv[u] = ok ? kq[i] : ~0ull;
            pv[u] = ok ? cpr[i] : (uint16_t)0;
        }
        auto for_est = [&](auto fn) __attribute__((always_inline)) { // fn(estimate key, it is a candidate's)
            if (inreg) {
#pragma unroll
                for (int u = 0; u < BR; ++u) fn((uint32_t)(kv[u] >> 32), (uint32_t)tid + 256u * u < n);
            } else {
                for (uint32_t i = tid; i < n; i += 256) fn((uint32_t)(kq[i] >> 32), true);
            }
        };
        __shared__ uint32_t red_min[4], red_max[4];
        uint32_t vmin = 0xffffffffu, vmax = 0u;
        for_est([&](uint32_t v, bool ok) {
            if (ok) vmin = min(vmin, v), vmax = max(vmax, v);
        });
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            vmin = min(vmin, (uint32_t)__shfl_xor((int)vmin, off, 64));
            vmax = max(vmax, (uint32_t)__shfl_xor((int)vmax, off, 64));
        }
        if ((tid & 63) == 0) red_min[tid >> 6] = vmin, red_max[tid >> 6] = vmax;
        if (tid == 0) {
            sel_prefix = 0u;
            sel_need = (uint32_t)p.k;
            nkeep = 0u;
        }
        __syncthreads();
        vmin = min(min(red_min[0], red_min[1]), min(red_min[2], red_min[3]));
        vmax = max(max(red_max[0], red_max[1]), max(red_max[2], red_max[3]));
        const uint32_t range = vmax - vmin;
        const int npass = range == 0u ? 0 : (32 - __builtin_clz(range) + 7) >> 3;
        for (int pass = npass - 1; pass >= 0; --pass) { // radix select of the k-th smallest estimate key
            hist[tid] = 0u;
            __syncthreads();
            const uint32_t pre = sel_prefix;
            for_est([&](uint32_t v, bool ok) {
                const uint32_t w = v - vmin;
                if (ok && (pass == 3 || (w >> (8 * (pass + 1))) == pre)) atomicAdd(&hist[(w >> (8 * pass)) & 255u], 1u);
            });
            __syncthreads();
            if (tid < 64) {
                uint32_t c[4], sum = 0;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    c[u] = hist[4 * tid + u];
                    sum += c[u];
                }
                const uint32_t inc = wave_incl_scan(sum);
                uint32_t before = inc - sum;
                const uint32_t need = sel_need;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (before < need && need <= before + c[u]) {
                        sel_prefix = (pre << 8) | (uint32_t)(4 * tid + u);
                        sel_need = need - before;
                    }
                    before += c[u];
                }
            }
            __syncthreads();
        }
        const float T2 = unordkey<METRIC>(vmin + sel_prefix);
        float thr = METRIC == METRIC_L2 ? T2 + 2.f * E : T2 - 2.f * E;
        if (thr != thr) thr = lmf_worst<METRIC>();
        const uint32_t tkey = ordkey<METRIC>(thr);
        // (an estimate key of a collected row is never the invalid key; tkey >= the k-th key, so at least k rows stay)
        auto keep = [&](u64 key, uint16_t pr) __attribute__((always_inline)) {
            if ((uint32_t)(key >> 32) <= tkey) {
                const uint32_t s = atomicAdd(&nkeep, 1u);
                if (s < (uint32_t)LT_CAP) {
                    keep_k[s] = key;
                    keep_p[s] = pr;
                }
            }
        };
        if (inreg) {
#pragma unroll
            for (int u = 0; u < BR; ++u)
                if ((uint32_t)tid + 256u * u < n) keep(kv[u], pv[u]);
        } else {
            for (uint32_t i = tid; i < n; i += 256) keep(kq[i], cpr[i]);
        }
        __syncthreads();
        const uint32_t kept = nkeep;
        if (kept <= (uint32_t)LT_CAP) { // (workgroup-uniform; every read of the segment happened before the barrier)
            for (uint32_t i = tid; i < kept; i += 256) {
                kq[i] = keep_k[i];
                cpr[i] = keep_p[i];
            }
            if (tid == 0) p.cnt[q] = kept;
            if (tid == 0 && fin_cap > 0 && kept > (uint32_t)fin_cap) overflow();
            return;
        }
    }
    if (tid == 0 && fin_cap > 0 && n > (uint32_t)fin_cap) overflow();
}
void launch_ivf_lmf_tighten(const IvfLmParams& p, int fin_cap, hipStream_t stream) {
    if (p.nq == 0) return;
    FA_THROW_IF_NOT(p.err_f && p.cand_pr && p.keys && p.cnt && p.ovf);
    if (p.metric == METRIC_L2) hipLaunchKernelGGL(lmf_tighten_kernel<METRIC_L2>, dim3((unsigned)p.nq), dim3(256), 0, stream, p, fin_cap);
    else hipLaunchKernelGGL(lmf_tighten_kernel<METRIC_INNER_PRODUCT>, dim3((unsigned)p.nq), dim3(256), 0, stream, p, fin_cap);
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------ IVFPQ: per-search preparation
// The sweeps' B operands and query terms, once per search instead of at every work item (a list of 24 000 rows is cut into
// four items, each of which loaded 96 fp32 queries + the centroid and rounded the differences: 48 loads and ~200 VALU
// instructions per query block and item -- half of an item's life at nb = 1M):
//   L2: pair16[(q, probe)][d] = fp16 of (q - centroid), pair_xh[(q, probe)] = -|q - c|^2 / 2;
//   inner product: pair16[q][d] = fp16 of q (the coarse term of a pair is read from coarse_dis).
// xn_bound[q] = max over the probes of |q - c|^2 (inner product: |q|^2): the |q'|^2 of the error band.
// Sixteen lanes per (query, probe): lane (s, h) owns the coordinates 16 s + 8 h .. + 7 -- one operand piece of the sweeps --
// so a pair reads its query and centroid rows and writes its fp16 row as contiguous 512 / 256 bytes.  |q - c|^2 is the
// tree sum of the sixteen 8-term chains (any order serves: the value only enters estimates, and the error band covers the
// fp32 chains of the norms with (d + 8) 2^-23 of their magnitude).
__global__ void __launch_bounds__(256) lmf_pq_prepare_kernel(IvfLmParams p, float* __restrict__ xn_bound) {
    const int np = p.nprobe;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t pair = t >> 4;
    const int sub = (int)(t & 15);
    if (pair >= (int64_t)p.nq * np) return; // (whole groups of 16 lanes leave together)
    const int q = (int)(pair / np), pr = (int)(pair - (int64_t)q * np);
    const bool l2 = p.metric == METRIC_L2;
    if (!l2 && pr != 0) return; // (inner product: one row per query)
    const int64_t l = p.coarse_ids[pair];
    float acc = 0.f;
    if (8 * sub < p.d) {
        const float* x = p.xq + (int64_t)q * p.ldq + 8 * sub;
        const f32x4 v0 = *(const f32x4*)x, v1 = *(const f32x4*)(x + 4);
        f32x4 c0 = f32x4{0.f, 0.f, 0.f, 0.f}, c1 = c0;
        if (l2 && l >= 0) {
            const float* c = p.centroids + l * p.ldc + 8 * sub;
            c0 = *(const f32x4*)c;
            c1 = *(const f32x4*)(c + 4);
        }
        half8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float r0 = v0[e] - c0[e], r1 = v1[e] - c1[e];
            acc = __fmaf_rn(r0, r0, acc);
            acc = __fmaf_rn(r1, r1, acc);
            o[e] = (_Float16)r0;
            o[4 + e] = (_Float16)r1;
        }
        *(half8*)((_Float16*)p.pair16 + (l2 ? pair : (int64_t)q) * p.d + 8 * sub) = o;
    }
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) acc += __shfl_xor(acc, off, 64);
    if (sub == 0) {
        if (l2) p.pair_xh[pair] = -0.5f * acc;
        if (l >= 0 || !l2) atomicMax((unsigned*)xn_bound + q, __float_as_uint(acc * 1.0001f)); // (non-negative floats order as integers)
    }
}
void launch_ivf_lmf_pq_prepare(const IvfLmParams& p, float* xn_bound, hipStream_t stream) {
    if (p.nq == 0) return;
    FA_THROW_IF_NOT(p.kind == 1 && p.centroids && p.pair16 && p.pair_xh && p.d % 16 == 0 && p.d <= 128 && p.ldq % 4 == 0 && p.ldc % 4 == 0);
    if (!p.pre_cleared) HIP_CHECK(hipMemsetAsync(xn_bound, 0, (size_t)p.nq * 4, stream));
    hipLaunchKernelGGL(lmf_pq_prepare_kernel, dim3((unsigned)div_up((size_t)p.nq * p.nprobe * 16, 256)), dim3(256), 0, stream, p,
                       xn_bound);
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------ scalar quantizer: per-search preparation
// B operands and query terms of every (query, probe) pair (kernels.h IvfLmParams, kind 2), with a_j = (q_j [- centroid_j]) - b'_j:
//   L2: pair16 = fp16(a o s), pair_xh = -|a|^2 / 2;        IP: pair16 = fp16(q o s), pair_xh = <q, b'> [+ coarse term]
// xn_bound[q] = max over the probes of |B|^2 (the operand's norm: the matrix pipe's share of the error band), an_bound[q] = max of
// |a|^2 (L2) / |q|^2 (IP) (the fp32 chains' share); qflags[q] |= 1 when a B coordinate leaves the fp16 range.  P = dh / 8 lanes
// (a power of two, 2 .. 64) per pair, one 8-coordinate operand piece each.
template <int P>
__global__ void __launch_bounds__(256) lmf_sq_prepare_kernel(IvfLmParams p, float* __restrict__ xn_bound, float* __restrict__ an_bound) {
    const int np = p.nprobe;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t pair = t / P; // (P a power of two: a shift)
    const int sub = (int)(t & (P - 1));
    if (pair >= (int64_t)p.nq * np) return; // (whole groups leave together: P divides 64)
    const int q = (int)(pair / np);
    const bool l2 = p.metric == METRIC_L2;
    const int64_t l = p.coarse_ids[pair];
    const bool res = p.sq_by_residual && l >= 0;
    float an = 0.f, bn = 0.f, qb = 0.f;
    bool bad = false;
    if (8 * sub < (int)p.ldh) {
        half8 o;
        // (rows of xq / centroids / the decoder tables are padded to multiples of 8 floats with zeros: whole 16-byte loads)
        float x[8], sj[8], bj[8], cj[8];
        const bool in = 8 * sub < p.dpad;
#pragma unroll
        for (int v4 = 0; v4 < 2; ++v4) {
            const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
            const f32x4 x4 = in ? *(const f32x4*)(p.xq + (int64_t)q * p.ldq + 8 * sub + 4 * v4) : z;
            const f32x4 s4 = in ? *(const f32x4*)(p.sq_s + 8 * sub + 4 * v4) : z;
            const f32x4 b4 = in ? *(const f32x4*)(p.sq_b + 8 * sub + 4 * v4) : z;
            const f32x4 c4 = in && res ? *(const f32x4*)(p.centroids + l * p.ldc + 8 * sub + 4 * v4) : z;
#pragma unroll
            for (int e = 0; e < 4; ++e) x[4 * v4 + e] = x4[e], sj[4 * v4 + e] = s4[e], bj[4 * v4 + e] = b4[e], cj[4 * v4 + e] = c4[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float B = 0.f;
            if (8 * sub + e < p.d) {
                if (l2) {
                    const float a = (x[e] - cj[e]) - bj[e];
                    an = __fmaf_rn(a, a, an);
                    B = a * sj[e];
                } else {
                    an = __fmaf_rn(x[e], x[e], an);
                    qb = __fmaf_rn(x[e], bj[e], qb);
                    B = x[e] * sj[e];
                }
            }
            if (!(fabsf(B) <= 65000.f)) bad = true;
            bn = __fmaf_rn(B, B, bn);
            o[e] = (_Float16)B;
        }
        *(half8*)((_Float16*)p.pair16 + pair * p.ldh + 8 * sub) = o;
    }
#pragma unroll
    for (int off = 1; off < P; off <<= 1) {
        an += __shfl_xor(an, off, 64);
        bn += __shfl_xor(bn, off, 64);
        qb += __shfl_xor(qb, off, 64);
        bad = bad || __shfl_xor((int)bad, off, 64);
    }
    if (sub == 0) {
        p.pair_xh[pair] = l2 ? -0.5f * an : qb + (p.sq_by_residual ? p.coarse_dis[pair] : 0.f);
        if (l >= 0) {
            atomicMax((unsigned*)xn_bound + q, __float_as_uint(bn * 1.0001f));
            atomicMax((unsigned*)an_bound + q, __float_as_uint(an * 1.0001f));
        }
        // (a pair without a list -- a preassigned -1 column -- is never scanned: its operand must not send the query to the redo)
        if (l >= 0 && (bad || !(an <= 3.0e38f))) atomicOr(const_cast<uint32_t*>(p.qflags) + q, 1u);
    }
}
void launch_ivf_lmf_sq_prepare(const IvfLmParams& p, float* xn_bound, float* an_bound, hipStream_t stream) {
    if (p.nq == 0) return;
    FA_THROW_IF_NOT((p.kind == 2 || p.lmf_pairb) && p.pair16 && p.pair_xh && p.qflags && p.sq_s && p.sq_b && p.centroids &&
                    p.ldh % 16 == 0 && p.ldh <= 512);
    FA_THROW_IF_NOT(p.dpad % 8 == 0 && p.ldq % 4 == 0 && p.ldc % 4 == 0);
    if (!p.pre_cleared) {
        HIP_CHECK(hipMemsetAsync(xn_bound, 0, (size_t)p.nq * 4, stream));
        HIP_CHECK(hipMemsetAsync(an_bound, 0, (size_t)p.nq * 4, stream));
    }
    // the scalar quantizer's flags come from this launch alone; the decoded-residual IVFPQ sweeps (kind 1, pair operands) run
    // launch_prep_queries first, whose NaN / fp16-range flags of the raw queries stay: this launch only ORs into them
    if (p.kind == 2 && !p.pre_cleared) HIP_CHECK(hipMemsetAsync(const_cast<uint32_t*>(p.qflags), 0, (size_t)p.nq * 4, stream));
    const int pieces = (int)p.ldh / 8; // 2 .. 64
    const int P = pieces <= 16 ? 16 : pieces <= 32 ? 32 : 64;
    const dim3 grid((unsigned)div_up((size_t)p.nq * p.nprobe * P, 256)), block(256);
    if (P == 16) hipLaunchKernelGGL(lmf_sq_prepare_kernel<16>, grid, block, 0, stream, p, xn_bound, an_bound);
    else if (P == 32) hipLaunchKernelGGL(lmf_sq_prepare_kernel<32>, grid, block, 0, stream, p, xn_bound, an_bound);
    else hipLaunchKernelGGL(lmf_sq_prepare_kernel<64>, grid, block, 0, stream, p, xn_bound, an_bound);
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------ rerank: exact distances of the candidates
// One workgroup per query, eight lanes per candidate row (the first version ran a wavefront per query: ~19 rounds of
// dependent loads each, 0.18 ms at nb = 1M and 10M alike -- latency, not bytes).
// IVFFlat: the arithmetic of ivfflat_fused_kernel -- lane ln of the group owns the 16-byte chunks ln, ln + 8, ... of the row
// and keeps one sequential fmaf chain of (q - y)^2 (inner product: q * y) over them; the eight partial sums meet in the
// xor butterfly ((p0 + p1) + (p2 + p3)) + ((p4 + p5) + (p6 + p7)).
template <int METRIC>
__global__ void __launch_bounds__(256) lmf_rerank_flat_kernel(IvfLmParams p) {
    __shared__ u64 sel_k[kLmfFusedSelectN];
    __shared__ uint32_t sel_wk[kLmfFusedSelectK];
    __shared__ int64_t sel_wl[kLmfFusedSelectK];
    const bool fin = p.fin_dis != nullptr; // (launch_ivf_lmf_rerank checked k and stride)
    const int q = blockIdx.x;
    const int ln = threadIdx.x & 7, grp = threadIdx.x >> 3; // 32 groups: 32 candidates per round, their loads in flight together
    const int np = p.nprobe;
    int n = (int)min((int64_t)p.cnt[q], p.stride);
    if (fin) n = min(n, kLmfFusedSelectN); // (more: the tightening launch listed the query for the redo; its rows here are dropped)
    const int nch = p.dpad >> 2;
    u64* kq = p.keys + (int64_t)q * p.stride;
    const uint16_t* cpr = p.cand_pr + (int64_t)q * p.stride;
    const float* qrow = p.xq + (int64_t)q * p.ldq;
    // Two phases per 256 candidates (round 5): first every thread resolves ONE candidate's row (key -> probe -> row base: two
    // dependent loads, all 256 chains in flight together), then the groups of eight lanes walk the rows.  (One phase -- every group
    // resolving its own candidate before reading it -- paid the chain once per round of 32 candidates: 0.154 ms at nb = 1M.)
    __shared__ int64_t s_row[256];
    __shared__ uint32_t s_pos[256];
    for (int cbase = 0; cbase < n; cbase += 256) {
        {
            const int i = cbase + (int)threadIdx.x;
            if (i < n) {
                const uint32_t pos = (uint32_t)kq[i];
                const int pr = (int)cpr[i];
                s_row[threadIdx.x] = p.row_base[(int64_t)q * np + pr] + (int64_t)pos;
                s_pos[threadIdx.x] = pos;
            }
        }
        __syncthreads();
        const int cn = min(256, n - cbase);
        for (int base = 0; base < cn; base += 32) {
            const int ci = base + grp;
            const bool valid = ci < cn;
            float a = 0.f;
            uint32_t pos = 0;
            if (valid) {
                pos = s_pos[ci];
                const float* rowp = p.arena_vecs + s_row[ci] * p.ldv;
                for (int c4 = ln; c4 < nch; c4 += 8) {
                    const f32x4 y4 = *(const f32x4*)(rowp + 4 * c4);
                    const f32x4 q4 = *(const f32x4*)(qrow + 4 * c4);
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
            a = a + __shfl_xor(a, 1, 64);
            a = a + __shfl_xor(a, 2, 64);
            a = a + __shfl_xor(a, 4, 64);
            if (valid && ln == 0) {
                const u64 key = ((u64)ordkey<METRIC>(a) << 32) | (u64)pos;
                if (fin) sel_k[cbase + ci] = key;
                else kq[cbase + ci] = key;
            }
        }
        __syncthreads(); // (s_row / s_pos are rewritten by the next 256 candidates; kq[i] of this chunk was read in phase one)
    }
    if (fin) lmf_select_tail<256>(p, q, n, sel_k, cpr, sel_wk, sel_wl);
}
// IVFPQ: the arithmetic of ivfpq_fused_kernel.  The workgroup first builds the query's table exactly as the query-major
// scan does (oracle orc_ivf_search_ex arith 0): entries <q_m, cb[m][c]> as sequential fmaf chains from 0, B = sum_m max_c
// |entry| in sub-quantizer order, the power-of-two grid pq_lut_grid(B), every entry rounded to it -- M x 256 floats in LDS
// (the first version recomputed the entries per candidate from the L2-resident codebook: 0.7 ms at nb = 10M / 100M).  Then
// S = sum_m table[m][code_m] (every partial sum exact in fp32, so any order gives these bits; lane ln of a candidate's
// group takes the sub-quantizers ln, ln + 8, ..., the partial sums meet in a butterfly), L2: fmaf(-2, S, coarse + t2[row]),
// inner product: coarse + S.  Without a grid (NaN / inf / all-zero tables) the sum runs in sub-quantizer order on one
// lane, like the oracle.
constexpr int RRQ_THREADS = 512; // 64 candidate groups of 8 lanes per round
template <int METRIC>
__global__ void __launch_bounds__(RRQ_THREADS) lmf_rerank_pq_kernel(IvfLmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int q = blockIdx.x;
    const int tid = threadIdx.x;
    const int ln = tid & 7, grp = tid >> 3;
    const int np = p.nprobe, M = p.M, dsub = p.dsub;
    int n = (int)min((int64_t)p.cnt[q], p.stride);
    if (p.fin_dis) n = min(n, kLmfFusedSelectN);
    float* lut = (float*)smem;                      // [256][M]
    uint32_t* colmax = (uint32_t*)(lut + M * 256);  // [M]
    float* grid = (float*)(colmax + M);             // delta, 1 / delta, on
    // fused selection (IvfLmParams::fin_dis): keys + winners behind the table
    const bool fin = p.fin_dis != nullptr;
    u64* sel_k = (u64*)(smem + ((M * 1024 + M * 4 + 16 + 15) & ~15));
    int64_t* sel_wl = (int64_t*)(sel_k + kLmfFusedSelectN);
    uint32_t* sel_wk = (uint32_t*)(sel_wl + kLmfFusedSelectK);
    if (n == 0) { // (workgroup-uniform)
        if (fin) lmf_select_tail<RRQ_THREADS>(p, q, 0, sel_k, nullptr, sel_wk, sel_wl);
        return;
    }
    u64* kq = p.keys + (int64_t)q * p.stride;
    const uint16_t* cpr = p.cand_pr + (int64_t)q * p.stride;
    const float* x = p.xq + (int64_t)q * p.ldq;
    for (int m = tid; m < M; m += RRQ_THREADS) colmax[m] = 0u;
    __syncthreads();
    // table entries e = c * M + m in the order of the transposed codebook pq_t [256][M][dsub] (coalesced reads, conflict-free
    // LDS writes).  When M divides the workgroup size a thread meets one sub-quantizer only: its maximum stays in a
    // register; dsub = 2 keeps the codebook loads of 8 entries in flight (one entry at a time the loop was a chain of
    // dependent L2 round trips: 0.7 ms for 10 000 tables).
    const int ne = M * 256;
    if (RRQ_THREADS % M == 0) {
        const int m = tid % M;
        uint32_t mx = 0u;
        if (dsub == 2) {
            const float x0 = x[2 * m], x1 = x[2 * m + 1];
            for (int e0 = tid; e0 < ne; e0 += 8 * RRQ_THREADS) {
                float2 c[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = e0 + u * RRQ_THREADS;
                    c[u] = e < ne ? *(const float2*)(p.pq_t + (size_t)e * 2) : float2{0.f, 0.f};
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = e0 + u * RRQ_THREADS;
                    if (e < ne) {
                        const float acc = __fmaf_rn(x1, c[u].y, __fmaf_rn(x0, c[u].x, 0.f));
                        lut[e] = acc;
                        mx = max(mx, __float_as_uint(fabsf(acc))); // (bit patterns: NaN beats every number, like the oracle)
                    }
                }
            }
        } else {
            for (int e = tid; e < ne; e += RRQ_THREADS) {
                const float* cen = p.pq_t + (size_t)e * dsub;
                float acc = 0.f;
                for (int jd = 0; jd < dsub; ++jd) acc = __fmaf_rn(x[m * dsub + jd], cen[jd], acc);
                lut[e] = acc;
                mx = max(mx, __float_as_uint(fabsf(acc)));
            }
        }
        atomicMax(&colmax[m], mx);
    } else {
        for (int e = tid; e < ne; e += RRQ_THREADS) {
            const int m = e % M;
            const float* cen = p.pq_t + (size_t)e * dsub;
            float acc = 0.f;
            for (int jd = 0; jd < dsub; ++jd) acc = __fmaf_rn(x[m * dsub + jd], cen[jd], acc);
            lut[e] = acc;
            atomicMax(&colmax[m], __float_as_uint(fabsf(acc)));
        }
    }
    __syncthreads();
    if (tid == 0) {
        float B = 0.f;
        for (int m = 0; m < M; ++m) B = B + __uint_as_float(colmax[m]);
        float delta = 0.f, inv = 0.f;
        const bool on = pq_lut_grid(B, &delta, &inv);
        grid[0] = delta;
        grid[1] = inv;
        grid[2] = on ? 1.f : 0.f;
    }
    __syncthreads();
    // (the entries are rounded to the grid where they are looked up: ~10 000 lookups per query against 16 384 entries, and no
    // second pass over the table)
    const bool on = grid[2] != 0.f;
    const float delta = grid[0], inv = grid[1];
    // candidates: lane ln of a group takes the sub-quantizers ln, ln + 8, ...; the groups of a wavefront start at different
    // ones (rotation by the group number), so that their gathers fall on different LDS banks (bank = m mod 32 in [c][m])
    const int nm8 = (M + 7) >> 3;
    const bool fastc = pq_chunk_bytes(M) == 16 && (M == 32 || M == 64 || M == 128);
    const int B8 = M >> 3; // fastc: stored bytes per lane of a candidate's group
    for (int base = 0; base < n; base += RRQ_THREADS / 8) {
        const int i = base + grp;
        const bool valid = i < n;
        float s = 0.f, dis0 = 0.f, t2 = 0.f;
        uint32_t pos = 0;
        if (valid) {
            pos = (uint32_t)kq[i];
            const int pr = (int)cpr[i];
            const int64_t row = p.row_base[(int64_t)q * np + pr] + (int64_t)pos;
            dis0 = p.coarse_dis[(int64_t)q * np + pr];
            if (METRIC == METRIC_L2) t2 = p.arena_t2[row];
            if (on && fastc) {
                // the row's M stored bytes in ONE load per lane: lane ln takes the stored bytes ln * M / 8 .. (a piece of a
                // 16-byte chunk of the rotated block layout, kernels.h pq_code_offset); stored byte x is sub-quantizer
                // (x + row) mod M.  (Byte by byte through pq_code_offset the 64 dependent loads per candidate made this
                // kernel cost 1.2 ms at nb = 10M.)
                const int x0 = ln * B8;
                const uint8_t* src = p.arena_codes + (size_t)(row >> 6) * 64 * M + (size_t)(x0 >> 4) * 1024 + (size_t)(row & 63) * 16 + (x0 & 15);
                unsigned w[4] = {0u, 0u, 0u, 0u};
                if (B8 == 16) {
                    const uint4 v = *(const uint4*)src;
                    w[0] = v.x, w[1] = v.y, w[2] = v.z, w[3] = v.w;
                } else if (B8 == 8) {
                    const uint2 v = *(const uint2*)src;
                    w[0] = v.x, w[1] = v.y;
                } else {
                    w[0] = *(const unsigned*)src;
                }
                const int lrot = (int)(row & 63);
#pragma unroll
                for (int b = 0; b < 16; ++b) {
                    if (b < B8) {
                        const int m = (x0 + b + lrot) & (M - 1);
                        const unsigned code = (w[b >> 2] >> (8 * (b & 3))) & 255u;
                        s = s + __builtin_rintf(lut[(int)code * M + m] * inv) * delta;
                    }
                }
            } else if (on) {
                for (int i8 = 0; i8 < nm8; ++i8) {
                    const int m = ln + 8 * ((i8 + grp) % nm8);
                    if (m < M) s = s + __builtin_rintf(lut[(int)p.arena_codes[pq_code_offset(M, row, m)] * M + m] * inv) * delta;
                }
            } else if (ln == 0) {
                for (int m = 0; m < M; ++m) s = s + lut[(int)p.arena_codes[pq_code_offset(M, row, m)] * M + m];
            }
        }
        if (on) {
            s = s + __shfl_xor(s, 1, 64);
            s = s + __shfl_xor(s, 2, 64);
            s = s + __shfl_xor(s, 4, 64);
        }
        if (valid && ln == 0) {
            const float dis = METRIC == METRIC_L2 ? __fmaf_rn(-2.f, s, dis0 + t2) : dis0 + s;
            const u64 key = ((u64)ordkey<METRIC>(dis) << 32) | (u64)pos;
            if (fin) sel_k[i] = key;
            else kq[i] = key;
        }
    }
    if (fin) lmf_select_tail<RRQ_THREADS>(p, q, n, sel_k, cpr, sel_wk, sel_wl);
}
// PQ64 over d = 128 (round 5): one WAVEFRONT per query, the fp32 codebook in LDS for the life of a 16-wave workgroup per CU.
// The kernel above gives every query a 512-thread workgroup that re-reads the 128 KB codebook from L2 and writes the query's
// 64 KB table into LDS before it looks at ~150 candidates: 0.24 ms per 10 000 queries whatever the database size -- a
// quarter of the IVF4096,PQ64 search at nb = 1M.  Here no table is stored: a query needs (a) the grid, i.e. B = sum_m max_c
// |entry(m, c)| -- lane m walks the 256 entries of sub-quantizer m (codebook reads from LDS, the running maximum in a
// register, no atomics, no barrier) -- and (b) the 64 entries its candidates' codes select, recomputed per candidate from the
// LDS codebook: lane = candidate, the row's 64 stored bytes in four 16-byte loads, all of a pass's metadata loads in flight
// together.  Same arithmetic as above and as ivfpq_fused_kernel: entries as fmaf chains from 0, B summed in sub-quantizer
// order, entries rounded to the grid where they are looked up, the sum order-free.
constexpr int RW_THREADS = 1024;
template <int METRIC>
__global__ void __launch_bounds__(RW_THREADS) lmf_rerank_pq64_kernel(IvfLmParams p) {
    constexpr int M = 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* cbt = (float2*)smem;                 // [256][64] entries (two coordinates each): 128 KB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float2* qx = (float2*)(smem + 256 * 64 * 8) + wave * 64; // this wave's query, coordinate pairs by sub-quantizer
    {
        const uint4* src = (const uint4*)p.pq_t;
        uint4* dst = (uint4*)smem;
        for (int i = tid; i < 256 * 64 * 8 / 16; i += RW_THREADS) dst[i] = src[i];
    }
    __syncthreads();
    const int np = p.nprobe;
    // Latency, not arithmetic, bounds this kernel (skipping all of (b)'s arithmetic moved 0.21 ms to 0.17): a query's
    // candidates hang off a chain of dependent loads (count -> key / probe number -> row base -> codes), and a wave that walks
    // the chain once per 64 candidates and once per query spends its time waiting.  So: the NEXT query's number, count and
    // coordinates are fetched while this one is worked on; the key / probe number / row base / norm loads of up to four
    // passes are issued together BEFORE the walk over the codebook entries (a), which hides them; and the code bytes of pass
    // j + 1 are in flight while pass j is summed.
    constexpr int NPF = 4; // passes of 64 candidates whose metadata is fetched together
    // the workgroup's share of the queries, handed out to its waves through a counter in LDS (ten thousand atomics on ONE
    // device-wide counter serialise at the memory side: ~ 17 ns each, which WAS the kernel's duration)
    uint32_t* wq = (uint32_t*)(smem + 256 * 64 * 8 + (RW_THREADS / 64) * 64 * 8);
    const int q_end = (int)(((int64_t)p.nq * (blockIdx.x + 1)) / gridDim.x);
    if (tid == 0) *wq = (uint32_t)(((int64_t)p.nq * blockIdx.x) / gridDim.x);
    __syncthreads();
    int q = 0;
    if (lane == 0) q = (int)atomicAdd(wq, 1u);
    q = __builtin_amdgcn_readfirstlane(q);
    int n = 0;
    float2 x2 = make_float2(0.f, 0.f);
    if (q < q_end) {
        n = (int)min((int64_t)p.cnt[q], p.stride);
        x2 = *(const float2*)(p.xq + (int64_t)q * p.ldq + 2 * lane);
    }
    while (q < q_end) {
        int qn_raw = 0;
        if (lane == 0) qn_raw = (int)atomicAdd(wq, 1u); // (read after (a))
        u64* kq = p.keys + (int64_t)q * p.stride;
        const uint16_t* cpr = p.cand_pr + (int64_t)q * p.stride;
        const int64_t* rbq = p.row_base + (int64_t)q * np;
        const float* cdq = p.coarse_dis + (int64_t)q * np;
        float delta = 0.f, inv = 0.f;
        bool on = false;
        int qn = 0, nn = 0;
        float2 x2n = make_float2(0.f, 0.f);
        auto fetch_next = [&]() __attribute__((always_inline)) {
            qn = __builtin_amdgcn_readfirstlane(qn_raw);
            if (qn < q_end) {
                nn = (int)min((int64_t)p.cnt[qn], p.stride);
                x2n = *(const float2*)(p.xq + (int64_t)qn * p.ldq + 2 * lane);
            }
        };
        if (n == 0) fetch_next();
        for (int base0 = 0; base0 < n; base0 += 64 * NPF) {
            // ---- metadata of the group's passes (indices clamped: every lane loads, only valid lanes store)
            uint32_t pos[NPF];
            int pr[NPF];
#pragma unroll
            for (int j = 0; j < NPF; ++j) {
                const int i = min(base0 + 64 * j + lane, n - 1);
                pos[j] = (uint32_t)kq[i];
                pr[j] = (int)cpr[i];
            }
            int64_t row[NPF];
            float dis0[NPF], t2[NPF];
#pragma unroll
            for (int j = 0; j < NPF; ++j) {
                row[j] = rbq[pr[j]] + (int64_t)pos[j];
                dis0[j] = cdq[pr[j]];
            }
#pragma unroll
            for (int j = 0; j < NPF; ++j) t2[j] = METRIC == METRIC_L2 ? p.arena_t2[row[j]] : 0.f;
            if (base0 == 0) {
                __builtin_amdgcn_wave_barrier(); // (the reads of the previous query's pairs were issued: LDS keeps a wave's order)
                qx[lane] = x2;
                // ---- (a) max_c |entry(m = lane, c)|, as bit patterns (NaN beats every number, like the oracle)
                uint32_t mx = 0u;
#pragma unroll 8
                for (int c = 0; c < 256; ++c) {
                    const float2 e = cbt[c * M + lane];
                    const float acc = __fmaf_rn(x2.y, e.y, __fmaf_rn(x2.x, e.x, 0.f));
                    mx = max(mx, __float_as_uint(fabsf(acc)));
                }
                float B = 0.f; // in sub-quantizer order, in every lane
#pragma unroll
                for (int m = 0; m < M; ++m) B = B + __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)mx, m));
                on = pq_lut_grid(B, &delta, &inv);
                // the next query: its number arrived during (a); its count and coordinates travel during the passes
                fetch_next();
            }
            // ---- (b) candidates, 64 per pass
            if (on) {
                // a row's 64 stored bytes: chunk ch holds the stored bytes 16 ch .. 16 ch + 15 (kernels.h pq_code_offset);
                // stored byte x is the code of sub-quantizer (x + row) mod 64
                auto codes_of = [&](int64_t r, uint4 (&w)[4]) __attribute__((always_inline)) {
                    const uint8_t* rp = p.arena_codes + (size_t)(r >> 6) * 64 * M + (size_t)(r & 63) * 16;
#pragma unroll
                    for (int ch = 0; ch < 4; ++ch) w[ch] = *(const uint4*)(rp + ch * 1024);
                };
                uint4 w[4];
                codes_of(row[0], w);
#pragma unroll
                for (int j = 0; j < NPF; ++j) {
                    if (base0 + 64 * j >= n) break; // (wave-uniform)
                    uint4 wn[4] = {};
                    if (j + 1 < NPF && base0 + 64 * (j + 1) < n) codes_of(row[j + 1 < NPF ? j + 1 : j], wn);
                    const int lrot = (int)(row[j] & 63);
                    float s = 0.f;
#pragma unroll
                    for (int ch = 0; ch < 4; ++ch) {
                        const unsigned ww[4] = {w[ch].x, w[ch].y, w[ch].z, w[ch].w};
#pragma unroll
                        for (int b = 0; b < 16; ++b) {
                            const int m = (16 * ch + b + lrot) & (M - 1);
                            const unsigned code = (ww[b >> 2] >> (8 * (b & 3))) & 255u;
                            const float2 e = cbt[(int)code * M + m];
                            const float2 xm = qx[m];
                            const float ent = __fmaf_rn(xm.y, e.y, __fmaf_rn(xm.x, e.x, 0.f));
                            s = s + __builtin_rintf(ent * inv) * delta;
                        }
                    }
                    const float dis = METRIC == METRIC_L2 ? __fmaf_rn(-2.f, s, dis0[j] + t2[j]) : dis0[j] + s;
                    const int i = base0 + 64 * j + lane;
                    if (i < n) kq[i] = ((u64)ordkey<METRIC>(dis) << 32) | (u64)pos[j];
#pragma unroll
                    for (int ch = 0; ch < 4; ++ch) w[ch] = wn[ch];
                }
            } else {
                // no grid (NaN / inf / all-zero tables): the plain sum in sub-quantizer order, like the oracle
#pragma unroll
                for (int j = 0; j < NPF; ++j) {
                    const int i = base0 + 64 * j + lane;
                    if (i >= n) continue;
                    float s = 0.f;
                    for (int m = 0; m < M; ++m) {
                        const unsigned code = p.arena_codes[pq_code_offset(M, row[j], m)];
                        const float2 e = cbt[(int)code * M + m];
                        const float2 xm = qx[m];
                        s = s + __fmaf_rn(xm.y, e.y, __fmaf_rn(xm.x, e.x, 0.f));
                    }
                    const float dis = METRIC == METRIC_L2 ? __fmaf_rn(-2.f, s, dis0[j] + t2[j]) : dis0[j] + s;
                    kq[i] = ((u64)ordkey<METRIC>(dis) << 32) | (u64)pos[j];
                }
            }
        }
        q = qn, n = nn, x2 = x2n;
    }
}
void launch_ivf_lmf_rerank(const IvfLmParams& p, hipStream_t stream) {
    if (p.nq == 0) return;
    // (fused selection: launch_ivf_lmf_tighten left at most kLmfFusedSelectN candidates or listed the query for the redo)
    FA_THROW_IF_NOT(!p.fin_dis || (p.fin_ids && p.arena_ids && p.k <= kLmfFusedSelectK));
    FA_THROW_IF_NOT(p.row_base != nullptr); // (written by launch_ivf_lm_plan)
    const dim3 grid((unsigned)p.nq), block(256);
    if (p.kind == 0) {
        if (p.metric == METRIC_L2) hipLaunchKernelGGL(lmf_rerank_flat_kernel<METRIC_L2>, grid, block, 0, stream, p);
        else hipLaunchKernelGGL(lmf_rerank_flat_kernel<METRIC_INNER_PRODUCT>, grid, block, 0, stream, p);
    } else if (p.M == 64 && p.dsub == 2 && !p.fin_dis && p.rr_blocks > 0) {
        // the bench shape: a wavefront per query, the codebook in LDS (lmf_rerank_pq64_kernel)
        FA_THROW_IF_NOT((p.metric != METRIC_L2 || p.arena_t2) && p.pq_t && p.ldq % 2 == 0);
        const int lds = 256 * 64 * 8 + (RW_THREADS / 64) * 64 * 8 + 16;
        const int blocks = std::max(1, std::min(p.rr_blocks, (p.nq + RW_THREADS / 64 - 1) / (RW_THREADS / 64)));
        if (p.metric == METRIC_L2) {
            HIP_CHECK(hipFuncSetAttribute((const void*)lmf_rerank_pq64_kernel<METRIC_L2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            hipLaunchKernelGGL(lmf_rerank_pq64_kernel<METRIC_L2>, dim3((unsigned)blocks), dim3(RW_THREADS), lds, stream, p);
        } else {
            HIP_CHECK(hipFuncSetAttribute((const void*)lmf_rerank_pq64_kernel<METRIC_INNER_PRODUCT>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            hipLaunchKernelGGL(lmf_rerank_pq64_kernel<METRIC_INNER_PRODUCT>, dim3((unsigned)blocks), dim3(RW_THREADS), lds, stream, p);
        }
    } else {
        FA_THROW_IF_NOT((p.metric != METRIC_L2 || p.arena_t2) && p.pq_t);
        const int lds = ((p.M * 1024 + p.M * 4 + 16 + 15) & ~15) + (p.fin_dis ? kLmfFusedSelectN * 8 + kLmfFusedSelectK * 12 : 0);
        FA_THROW_IF_NOT(lds <= 160 * 1024);
        if (p.metric == METRIC_L2) {
            HIP_CHECK(hipFuncSetAttribute((const void*)lmf_rerank_pq_kernel<METRIC_L2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            hipLaunchKernelGGL(lmf_rerank_pq_kernel<METRIC_L2>, grid, dim3(RRQ_THREADS), lds, stream, p);
        } else {
            HIP_CHECK(hipFuncSetAttribute((const void*)lmf_rerank_pq_kernel<METRIC_INNER_PRODUCT>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            hipLaunchKernelGGL(lmf_rerank_pq_kernel<METRIC_INNER_PRODUCT>, grid, dim3(RRQ_THREADS), lds, stream, p);
        }
    }
    HIP_CHECK(hipGetLastError());
}

} // namespace faiss_amd
