// faiss_amd/csrc/flat_small.hip -- exact k-nearest rows of a SMALL database in ONE launch: the coarse quantizer of the
// IVF searches (nlist centroids, k = nprobe), where the five launches of the general filter path (query preparation
// aside: maxima, tighten, collect, re-rank -- flat_filter.hip) cost 0.16 ms of a 1.34 ms IVF4096,PQ64 search although
// they move next to nothing (tools/ivfpq_only.py, round 2).
//
// STATUS: written at the end of round 2 after the GPU budget was spent; compiles for gfx950, NOT YET RUN ON HARDWARE.
// Off unless FAISS_AMD_FLAT_SMALL=1 (GpuIndexFlat::search_tile_); the default path is untouched.
//
// Same contract as the filter path: fp16 MFMA scores select a candidate SUPERSET of the exact top-k, exact fp32 distances
// (the fmaf chain of flat_scan_kernel / flat_rerank_kernel / the oracle) decide, results bit-identical to the fp32 scan.
//   workgroup = 4 wavefronts = 32 queries (one MFMA column block, fp16 coordinates in 32 VGPRs per lane);
//   wavefront w owns the 32-row blocks w, w + 4, ... of the database (fp16 rows straight from L2: the table is 1 MB).
//   pass 1  scores of every row (v_mfma_f32_32x32x16_f16, accumulators start from -|y|^2/2); every lane keeps the
//           running maximum of each of its 16 fragment positions: 4 waves x 2 lane halves x 16 = 128 chunk maxima per
//           query, chunks = disjoint row sets.  Grouped four by four they give 32 maxima of disjoint row sets, so their
//           MINIMUM is a lower bound of the 32nd best score: no selection needed (k <= 32).
//   pass 2  the same scores again; rows above  bound - 2 e_q  (the rigorous fp16 error band, flat_filter_err_bound) go
//           to the query's candidate list in LDS with their scores: ~3 % of the rows (135 of 4096 on the bench's centroids).
//   narrow  eight threads per query: the k-th best score among the candidates (every key counts the better ones), and
//           only the rows within 2 e_q of it stay (k plus a handful) -- the re-rank kernel's band.
//   exact   the fp32 chain for those rows (512 bytes each from L2), key = (ordkey(distance) << 32) | row.
//   rank    every key counts the smaller ones: its rank is its output position.
// A query whose list overflows (or that left the fp16 range) is handed to the exact scan like the filter path does.
#include "kernels.h"

namespace faiss_amd {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

constexpr int FS_Q = 32;      // queries per workgroup
constexpr int FS_WAVES = 4;
constexpr int FS_THREADS = FS_WAVES * 64;
constexpr int FS_CAP = 512;   // candidates per query (simulated on IVF4096 centroids of the bench data: mean 135, 99th
                              // percentile 272, maximum 422 of 4096 rows; beyond the capacity the exact scan takes the query)
constexpr int FS_KMAX = 32;   // groups of chunk maxima = 32: the bound holds for k <= 32

// (the threshold rule of flat_filter.hip band_threshold: strictly below t_k - 2e, ties of the k-th score stay inside)
__device__ __forceinline__ float fs_band_threshold(float tk, float e) {
    return tk - 2.f * e - 9.6e-7f * fabsf(tk) - 1e-37f;
}

size_t flat_small_lds_bytes(int dpad) {
    return (size_t)FS_Q * FS_CAP * 8 + (size_t)FS_Q * dpad * 4 + (size_t)FS_Q * 8 * 4 + (size_t)FS_Q * 4 * 4 + (size_t)FS_Q * 8 + 64;
}

template <int METRIC>
__global__ void __launch_bounds__(FS_THREADS) flat_small_kernel(FlatSmallParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64* cand = (u64*)smem;                              // [FS_Q][FS_CAP] rows, then exact keys
    float* qs = (float*)(cand + FS_Q * FS_CAP);          // [FS_Q][dpad] fp32 queries
    float* gmin = qs + FS_Q * p.dpad;                    // [FS_Q][8] minimum of the group maxima per (wave, lane half)
    unsigned* lcnt = (unsigned*)(gmin + FS_Q * 8);       // [FS_Q] candidates
    float* lthr = (float*)(lcnt + FS_Q);                 // [FS_Q]
    unsigned* lbad = (unsigned*)(lthr + FS_Q);           // [FS_Q] 1 = hand the query to the exact path
    float* lerr = (float*)(lbad + FS_Q);                 // [FS_Q] e_q
    u64* lkth = (u64*)(lerr + FS_Q);                     // [FS_Q] k-th best candidate key
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31;
    const int q0 = blockIdx.x * FS_Q;
    const int nblk = (p.nb + 31) >> 5; // 32-row blocks (the fp16 rows and their start values are padded by a tile)

    // ---- this lane's query as B operands; fp32 queries, counters
    const int q = q0 + j;
    const int qc = q < p.nq ? q : p.nq - 1;
    half8 bq[8];
    {
        const _Float16* qrow = p.xqh + (int64_t)qc * p.ldqh;
#pragma unroll
        for (int s = 0; s < 8; ++s) bq[s] = *(const half8*)(qrow + s * 16 + h * 8);
    }
    for (int t = tid; t < FS_Q * p.dpad; t += FS_THREADS) {
        const int qi = t / p.dpad, c = t - qi * p.dpad;
        qs[t] = q0 + qi < p.nq ? p.xq[(int64_t)(q0 + qi) * p.ldq + c] : 0.f;
    }
    if (tid < FS_Q) {
        lcnt[tid] = 0;
        lbad[tid] = (q0 + tid < p.nq && p.flags[q0 + tid]) ? 1u : 0u;
    }

    // scores of the 32-row block rb for this lane's query: acc[4 g + e] = row 32 rb + 8 g + 4 h + e
    auto block_scores = [&](int rb) __attribute__((always_inline)) {
        const int r0 = rb * 32;
        f32x16 acc;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b4 = *(const f32x4*)(p.xbhn + r0 + 8 * g + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[4 * g + e] = b4[e];
        }
        const _Float16* yrow = p.xbh + (int64_t)(r0 + j) * p.ldbh + h * 8;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const half8 a = *(const half8*)(yrow + s * 16);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bq[s], acc, 0, 0, 0);
        }
        return acc;
    };

    // ---- pass 1: chunk maxima
    float mx[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) mx[i] = -INFINITY;
    for (int rb = wave; rb < nblk; rb += FS_WAVES) {
        const f32x16 acc = block_scores(rb);
#pragma unroll
        for (int i = 0; i < 16; ++i) mx[i] = fmaxf(mx[i], acc[i]);
    }
    {
        // four groups per lane (the positions 4 g .. 4 g + 3), their minimum -> LDS
        float lm = INFINITY;
#pragma unroll
        for (int g = 0; g < 4; ++g) lm = fminf(lm, fmaxf(fmaxf(mx[4 * g], mx[4 * g + 1]), fmaxf(mx[4 * g + 2], mx[4 * g + 3])));
        gmin[j * 8 + wave * 2 + h] = lm;
    }
    __syncthreads();
    if (tid < FS_Q) {
        float tk = INFINITY;
#pragma unroll
        for (int i = 0; i < 8; ++i) tk = fminf(tk, gmin[tid * 8 + i]);
        float thr = -INFINITY, e = 0.f;
        if (q0 + tid < p.nq) {
            e = flat_filter_err_bound(METRIC, p.d, p.xqn[q0 + tid], p.yn_max, false);
            if (!(e < FLT_MAX)) lbad[tid] = 1u; // (NaN / overflowing norms: the exact path)
            else if (tk > -INFINITY) thr = fs_band_threshold(tk, e);
        }
        lerr[tid] = e;
        lthr[tid] = thr;
        lkth[tid] = ~0ull;
    }
    __syncthreads();

    // ---- pass 2: rows above the threshold -> candidate list of the query
    {
        const float thr = lthr[j];
        const bool live = q < p.nq && lbad[j] == 0u;
        for (int rb = wave; rb < nblk; rb += FS_WAVES) {
            const f32x16 acc = block_scores(rb);
            unsigned mask = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) mask |= acc[i] > thr ? 1u << i : 0u;
            if (!live) mask = 0;
            if (mask) {
                const unsigned base = atomicAdd(&lcnt[j], (unsigned)__popc(mask));
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if ((mask >> i) & 1u) {
                        const unsigned slot = base + (unsigned)__popc(mask & ((1u << i) - 1u));
                        // (score key: larger score = smaller key, the order of ordkey<METRIC_INNER_PRODUCT>)
                        if (slot < (unsigned)FS_CAP)
                            cand[j * FS_CAP + slot] = ((u64)ordkey<METRIC_INNER_PRODUCT>(acc[i]) << 32) |
                                                      (unsigned)(rb * 32 + 8 * (i >> 2) + 4 * h + (i & 3));
                    }
                }
            }
        }
    }
    __syncthreads();

    // ---- eight threads per query from here on
    const int qi = tid >> 3, l8 = tid & 7;
    const int qq = q0 + qi;
    int n = (int)lcnt[qi];
    const bool bad = qq < p.nq && (lbad[qi] != 0u || n > FS_CAP);
    if (bad) n = 0;
    if (qq >= p.nq) n = 0;
    // ---- narrow: the k-th best score among the candidates (rank = number of better keys; keys are unique) ...
    if (n > p.k) {
        const u64* keys = cand + qi * FS_CAP;
        for (int c = l8; c < n; c += 8) {
            const u64 key = keys[c];
            int r = 0;
            for (int i = 0; i < n; ++i) r += keys[i] < key ? 1 : 0;
            if (r == p.k - 1) lkth[qi] = key;
        }
    }
    __syncthreads();
    // ... and the band below it: rows outside cannot be among the exact k best (flat_rerank_kernel's rule)
    u64 key_thr = ~0ull;
    if (n > p.k) {
        const float tk = unordkey<METRIC_INNER_PRODUCT>((uint32_t)(lkth[qi] >> 32));
        const float thr2 = fs_band_threshold(tk, lerr[qi]);
        key_thr = ((u64)ordkey<METRIC_INNER_PRODUCT>(thr2) << 32) | 0xffffffffull;
    }
    // ---- exact distances of the rows inside the band
    {
        const float* qr = qs + qi * p.dpad;
        const float xn = METRIC == METRIC_L2 && qq < p.nq ? p.xqn[qq] : 0.f;
        for (int c = l8; c < n; c += 8) {
            const u64 akey = cand[qi * FS_CAP + c];
            if (akey > key_thr) {
                cand[qi * FS_CAP + c] = ~0ull; // outside the band: behind every result
                continue;
            }
            const unsigned row = (unsigned)akey;
            const float* yr = p.xb + (int64_t)row * p.ldb;
            float acc = 0.f;
            // the chain of flat_scan_kernel / flat_rerank_kernel: 8-float steps, e and 4 + e interleaved; the loads of a
            // 32-float stretch of the row are in flight together
            int s = 0;
            for (; s + 32 <= p.dpad; s += 32) {
                f32x4 y[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) y[u] = *(const f32x4*)(yr + s + 4 * u);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const f32x4 x0 = *(const f32x4*)(qr + s + 8 * u), x1 = *(const f32x4*)(qr + s + 8 * u + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc = __fmaf_rn(y[2 * u][e], x0[e], acc);
                        acc = __fmaf_rn(y[2 * u + 1][e], x1[e], acc);
                    }
                }
            }
            for (; s < p.dpad; s += 8) {
                const f32x4 y0 = *(const f32x4*)(yr + s), y1 = *(const f32x4*)(yr + s + 4);
                const f32x4 x0 = *(const f32x4*)(qr + s), x1 = *(const f32x4*)(qr + s + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc = __fmaf_rn(y0[e], x0[e], acc);
                    acc = __fmaf_rn(y1[e], x1[e], acc);
                }
            }
            float dis;
            if (METRIC == METRIC_L2) {
                dis = __fmaf_rn(-2.f, acc, xn + p.xbn[row]);
                dis = dis < 0.f ? 0.f : dis;
            } else {
                dis = acc;
            }
            cand[qi * FS_CAP + c] = ((u64)ordkey<METRIC>(dis) << 32) | row;
        }
    }
    __syncthreads();

    // ---- every key to its rank (keys are unique): the k best, ordered by (distance, id)
    if (qq < p.nq && !bad) {
        const float pad = neutral_distance(METRIC);
        const u64* keys = cand + qi * FS_CAP;
        int live = 0; // keys inside the band (the others are ~0: never ranked, never written)
        for (int i = 0; i < n; ++i) live += keys[i] != ~0ull ? 1 : 0;
        for (int c = l8; c < n; c += 8) {
            const u64 key = keys[c];
            if (key == ~0ull) continue;
            int r = 0;
            for (int i = 0; i < n; ++i) r += keys[i] < key ? 1 : 0;
            if (r < p.k) {
                const uint32_t wk = (uint32_t)(key >> 32);
                const bool ok = wk < kInvalidOrdKey;
                p.out_dis[(int64_t)qq * p.k + r] = ok ? unordkey<METRIC>(wk) : pad;
                p.out_ids[(int64_t)qq * p.k + r] = ok ? (int64_t)(uint32_t)key + p.id_base : -1;
            }
        }
        for (int i = live + l8; i < p.k; i += 8) {
            p.out_dis[(int64_t)qq * p.k + i] = pad;
            p.out_ids[(int64_t)qq * p.k + i] = -1;
        }
    }
    if (bad && l8 == 0) p.ovf_list[atomicAdd(p.ovf_cnt, 1u)] = (uint32_t)qq;
}

bool flat_small_supported(int nb, int dh, int dpad, int k) {
    return nb >= 2048 && nb <= 8192 && dh == kFilterSlab && k <= FS_KMAX && flat_small_lds_bytes(dpad) <= 156 * 1024;
}

void launch_flat_small(const FlatSmallParams& p, hipStream_t stream) {
    if (p.nq == 0) return;
    FA_THROW_IF_NOT(flat_small_supported(p.nb, p.dh, p.dpad, p.k) && p.dpad % 8 == 0 && p.ldqh % 8 == 0 && p.ldbh % 8 == 0);
    const size_t lds = flat_small_lds_bytes(p.dpad);
    dim3 grid((unsigned)div_up((size_t)p.nq, FS_Q));
    if (p.metric == METRIC_L2) {
        HIP_CHECK(hipFuncSetAttribute((const void*)flat_small_kernel<METRIC_L2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds));
        hipLaunchKernelGGL((flat_small_kernel<METRIC_L2>), grid, dim3(FS_THREADS), lds, stream, p);
    } else {
        HIP_CHECK(hipFuncSetAttribute((const void*)flat_small_kernel<METRIC_INNER_PRODUCT>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((flat_small_kernel<METRIC_INNER_PRODUCT>), grid, dim3(FS_THREADS), lds, stream, p);
    }
    HIP_CHECK(hipGetLastError());
}

} // namespace faiss_amd
