// faiss_amd/csrc/flat_kernels.hip -- brute-force L2 / inner-product scan for gfx950 (MI355X).
//
// What it replaces in the reference: faiss/gpu/impl/Distance.cu:120-406 (runDistance: GEMM
// into a 512 x 262144 fp32 tile, l2SelectMinK re-reads the tile, sumAlongRows) and
// faiss/gpu/impl/L2Norm.cu.  Here the distance tile never exists in memory: every 32x32
// block of distances is produced in MFMA accumulators, compared against the running k-th
// best of its query in registers, and only the (rare) survivors are appended to a small
// per-(query, split) reservoir.
//
// Arithmetic contract (restated by oracle/faiss_oracle.c: orc_ip_chain / orc_flat_search):
//   ip(q, y)  = f32 fmaf chain over k in the order  for s: for e in 0..3: k = 8s+e, 8s+4+e
//               (v_mfma_f32_32x32x2_f32 is bit-for-bit a k-ordered fmaf chain)
//   L2: dis   = fmaf(-2, ip, |q|^2 + |y|^2);  if (dis < 0) dis = 0
//               (same formula as the CPU reference, faiss/utils/distances.cpp:480-495)
//   IP: dis   = ip
//   selection = k best under the total order (dis, id)  (faiss/impl/ResultHandler.h:276-281
//               strict admission + id-ascending scan; faiss/utils/ordered_key_value.h:74-76)
#include "kernels.h"
#include "wave_select.h"

namespace faiss_amd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------
// small kernels
// ---------------------------------------------------------------------------------
__global__ void l2_norms_kernel(const float* __restrict__ x, int64_t ld, int64_t n, int d,
                                float* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* r = x + i * ld;
    float acc = 0.f;
    for (int k = 0; k < d; ++k) {
        float v = r[k];
        acc = __fmaf_rn(v, v, acc);
    }
    out[i] = acc;
}

void launch_l2_norms(const float* x, int64_t ld, int64_t n, int d, float* out, hipStream_t stream) {
    if (n == 0) return;
    int bs = 256;
    hipLaunchKernelGGL(l2_norms_kernel, dim3((unsigned)div_up(n, bs)), dim3(bs), 0, stream, x, ld, n, d,
                       out);
}

__global__ void pad_rows_kernel(const float* __restrict__ src, int64_t ld_src, int64_t n, int d,
                                float* __restrict__ dst, int64_t ld_dst, int dpad) {
    int64_t total = n * dpad;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        int64_t i = t / dpad;
        int c = (int)(t - i * dpad);
        dst[i * ld_dst + c] = c < d ? src[i * ld_src + c] : 0.f;
    }
}

void launch_pad_rows(const float* src, int64_t ld_src, int64_t n, int d, float* dst, int64_t ld_dst,
                     int dpad, hipStream_t stream) {
    if (n == 0) return;
    int64_t total = n * dpad;
    int bs = 256;
    unsigned grid = (unsigned)std::min<int64_t>(div_up(total, bs), 65535 * 16);
    hipLaunchKernelGGL(pad_rows_kernel, dim3(grid), dim3(bs), 0, stream, src, ld_src, n, d, dst, ld_dst,
                       dpad);
}

// out[i][0..d) = (x ? x[i][0..d) : 0) -/= rows[key[i]][0..d): residuals (x given; faiss/gpu/impl/VectorResidual.cu:26-60,
// a key of -1 yields a row of NaNs like there) or a gather of stored rows (x null; FlatIndex::reconstruct by ids)
__global__ void rows_by_key_kernel(const float* __restrict__ x, int64_t ld_x, const int64_t* __restrict__ keys,
                                   int64_t n, int d, const float* __restrict__ rows, int64_t ld_rows, int64_t nrows,
                                   float* __restrict__ out, int64_t ld_out, const _Float16* __restrict__ rows16,
                                   int64_t ld_rows16) {
    const int64_t total = n * d;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = t / d;
        const int c = (int)(t - i * d);
        const int64_t key = keys[i];
        float v;
        if (key < 0 || key >= nrows) {
            v = __builtin_nanf("");
        } else {
            const float r = rows ? rows[key * ld_rows + c] : (float)rows16[key * ld_rows16 + c];
            v = x ? x[i * ld_x + c] - r : r;
        }
        out[i * ld_out + c] = v;
    }
}
void launch_rows_by_key(const float* x, int64_t ld_x, const int64_t* keys, int64_t n, int d, const float* rows,
                        int64_t ld_rows, int64_t nrows, float* out, int64_t ld_out, hipStream_t stream,
                        const _Float16* rows16, int64_t ld_rows16) {
    if (n == 0) return;
    const int64_t total = n * d;
    unsigned grid = (unsigned)std::min<int64_t>(div_up(total, 256), 65535 * 16);
    hipLaunchKernelGGL(rows_by_key_kernel, dim3(grid), dim3(256), 0, stream, x, ld_x, keys, n, d, rows, ld_rows, nrows,
                       out, ld_out, rows16, ld_rows16);
    HIP_CHECK(hipGetLastError());
}

__global__ void convert_matrix_kernel(const void* __restrict__ src, int type, bool row_major, int64_t n, int d,
                                      float* __restrict__ dst) {
    const int64_t total = n * d;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = t / d;
        const int c = (int)(t - i * d);
        const int64_t s = row_major ? t : (int64_t)c * n + i;
        float v;
        if (type == 1) {
            v = ((const float*)src)[s];
        } else if (type == 2) {
            v = (float)((const _Float16*)src)[s];
        } else {
            v = __uint_as_float((unsigned)((const unsigned short*)src)[s] << 16); // bf16 = upper half of an fp32
        }
        dst[t] = v;
    }
}
void launch_convert_matrix(const void* src, int type, bool row_major, int64_t n, int d, float* dst, hipStream_t stream) {
    if (n == 0) return;
    unsigned grid = (unsigned)std::min<int64_t>(div_up(n * d, 256), 65535 * 16);
    hipLaunchKernelGGL(convert_matrix_kernel, dim3(grid), dim3(256), 0, stream, src, type, row_major, n, d, dst);
    HIP_CHECK(hipGetLastError());
}
__global__ void i64_to_i32_kernel(const int64_t* __restrict__ src, int64_t n, int32_t* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (int32_t)src[i];
}
void launch_i64_to_i32(const int64_t* src, int64_t n, int32_t* dst, hipStream_t stream) {
    if (n == 0) return;
    hipLaunchKernelGGL(i64_to_i32_kernel, dim3((unsigned)div_up(n, 256)), dim3(256), 0, stream, src, n, dst);
    HIP_CHECK(hipGetLastError());
}

__global__ void round_f16_inplace_kernel(float* __restrict__ x, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        x[i] = (float)(_Float16)x[i];
}
void launch_round_f16_inplace(float* x, int64_t n, hipStream_t stream) {
    if (n == 0) return;
    unsigned grid = (unsigned)std::min<int64_t>(div_up(n, 256), 65535 * 16);
    hipLaunchKernelGGL(round_f16_inplace_kernel, dim3(grid), dim3(256), 0, stream, x, n);
    HIP_CHECK(hipGetLastError());
}
__global__ void f16_rows_to_f32_kernel(const _Float16* __restrict__ src, int64_t ld_src, int64_t n, int dpad,
                                       float* __restrict__ dst) {
    const int64_t total = n * dpad;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = t / dpad;
        const int c = (int)(t - i * dpad);
        dst[t] = (float)src[i * ld_src + c];
    }
}
void launch_f16_rows_to_f32(const _Float16* src, int64_t ld_src, int64_t n, int dpad, float* dst, hipStream_t stream) {
    if (n == 0) return;
    unsigned grid = (unsigned)std::min<int64_t>(div_up(n * dpad, 256), 65535 * 16);
    hipLaunchKernelGGL(f16_rows_to_f32_kernel, dim3(grid), dim3(256), 0, stream, src, ld_src, n, dpad, dst);
    HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------
// fused MFMA scan
//
// Workgroup = 512 threads = 8 waves; wave w owns 32 queries for the whole kernel (their d
// coordinates live in 64 VGPRs as MFMA B operands), the 8 waves share one stream of
// 64-row database tiles staged through LDS (double buffered, 16-byte XOR swizzle so the
// ds_read_b128 of 32 different rows at one column is bank-conflict free).
// MFMA D[i][j]: i = database row of the 32-row block, j = query => each lane holds 16
// distances of ONE query, so the threshold test needs no cross-lane traffic.
// ---------------------------------------------------------------------------------
constexpr int QPW = 32;
constexpr int WAVES = 8;
constexpr int QPB = QPW * WAVES; // kFlatQueriesPerBlock
constexpr int TR = kFlatTileRows; // 64
constexpr int KS = 128;           // k-slab staged per step
constexpr int TILE_BYTES = TR * KS * 4; // 32768
constexpr int LDS_TILES = 0;
constexpr int LDS_BIAS = 2 * TILE_BYTES;           // 2 x 64 floats
constexpr int LDS_HIST = LDS_BIAS + 2 * TR * 4;    // 8 waves x 256 uint
constexpr int LDS_TOTAL = LDS_HIST + WAVES * 256 * 4;

size_t flat_scan_lds_bytes() {
    return LDS_TOTAL;
}

// FULL: dpad is a multiple of 128 (every k-slab is complete).  That variant has no data-dependent
// branch in the MFMA loop, keeps the query operands of a single-slab problem (d <= 128) in
// registers for the whole kernel, and contains no global load between the tile prefetch and
// its consumer, so the compiler never has to drain the prefetch (s_waitcnt vmcnt(0)) early.
template <int METRIC, bool DUMP, bool FULL, bool SINGLE>
__global__ void __launch_bounds__(512, 2) flat_scan_kernel(FlatScanParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const int j = lane & 31;

    // ---- block -> (split, query group); consecutive blocks on one XCD share a split
    int split, grp;
    {
        int b = blockIdx.x;
        if ((p.nsplit & 7) == 0) {
            int xcd = b & 7;
            int t = b >> 3;
            grp = t % p.ngroups;
            split = (t / p.ngroups) * 8 + xcd;
        } else {
            split = b % p.nsplit;
            grp = b / p.nsplit;
        }
    }
    const int r0 = split * p.rows_per_split;
    const int r1 = min(p.nb, r0 + p.rows_per_split);
    if (r0 >= r1) {
        // empty split: publish zero counts
        if (!DUMP) {
            int q = grp * QPB + tid;
            if (tid < QPB && q < p.nq) p.res_cnt[(int64_t)q * p.nsplit + split] = 0;
        }
        return;
    }
    const int ntiles = (r1 - r0 + TR - 1) / TR;
    const int nslab = SINGLE ? 1 : (p.dpad + KS - 1) / KS; // SINGLE: d <= 128, one k-slab per tile
    const int nsteps = ntiles * nslab;

    // ---- this lane's query
    const int qbase = grp * QPB + wave * QPW; // wave-uniform
    const int q = qbase + j;
    const bool qvalid = q < p.nq;
    const int qc = qvalid ? q : p.nq - 1;
    const float* qrow = p.xq + (int64_t)qc * p.ldq;
    float xn = 0.f;
    if (METRIC == METRIC_L2) xn = p.xqn[qc];
    float tau;
    if (METRIC == METRIC_L2)
        tau = qvalid ? FLT_MAX : -INFINITY;
    else
        tau = qvalid ? -FLT_MAX : INFINITY;
    int cnt = 0;
    u64* resq = nullptr;       // this lane's query reservoir
    u64* res_wave = nullptr;   // reservoir of query qbase (wave-uniform)
    if (!DUMP) {
        res_wave = p.res_keys + ((int64_t)qbase * p.nsplit + split) * p.cap;
        resq = res_wave + (int64_t)j * p.nsplit * p.cap;
    }
    unsigned* hist = (unsigned*)(smem + LDS_HIST) + wave * 256;
    const int cap_lim = p.cap - 32;

    // The two waves that share a SIMD would otherwise run their MFMA phases in lockstep and
    // then both sit in the (VALU-only) epilogue with the matrix pipe idle; a static priority
    // for one of them staggers the pair so one wave's epilogue overlaps the other's MFMAs.
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);

    // ---- staging registers
    f32x4 stg[4];
    float stg_bias = 0.f;
    bool stg_ok = false;
    auto stage_load = [&](int u) {
        const int t = u / nslab, sl = u - t * nslab;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int g = tid + 512 * i;
            int row = g >> 5, c = g & 31;
            int col = sl * KS + c * 4;
            int grow = min(r0 + t * TR + row, r1 - 1);
            if (FULL) {
                stg[i] = *(const f32x4*)(p.xb + (int64_t)grow * p.ldb + col);
            } else {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (col < p.dpad) v = *(const f32x4*)(p.xb + (int64_t)grow * p.ldb + col);
                stg[i] = v;
            }
        }
        {
            // every thread loads unconditionally (static vmcnt bookkeeping); rows >= r1 get a bias
            // that can never pass the threshold
            const int grow = r0 + t * TR + (tid & (TR - 1));
            stg_ok = grow < r1;
            // raw load only; the select happens in stage_store so that nothing waits on it here
            if (METRIC == METRIC_L2) stg_bias = p.xbn[min(grow, r1 - 1)];
            else if (p.ip_bias) stg_bias = p.ip_bias[min(grow, r1 - 1)]; // IDSelector: -inf for excluded rows
        }
    };
    auto stage_store = [&](int u) {
        const int t = u / nslab, sl = u - t * nslab;
        char* tile = smem + LDS_TILES + (u & 1) * TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int g = tid + 512 * i;
            int row = g >> 5, c = g & 31;
            *(f32x4*)(tile + row * 512 + ((c ^ (row & 15)) << 4)) = stg[i];
        }
        if (sl == 0 && tid < TR) {
            float b;
            if (METRIC == METRIC_L2) b = stg_ok ? stg_bias : INFINITY;
            else b = stg_ok ? stg_bias : -INFINITY; // (stg_bias stays 0 without p.ip_bias)
            ((float*)(smem + LDS_BIAS))[(t & 1) * TR + tid] = b;
        }
    };

    f32x4 bq[16];
    f32x16 acc0, acc1;

    if (FULL) {
        // slab 0 operands; for d <= 128 these are the only B loads of the kernel
#pragma unroll
        for (int s = 0; s < 16; ++s) bq[s] = *(const f32x4*)(qrow + 8 * s + 4 * h);
    }
    stage_load(0);
    stage_store(0);
    __syncthreads();

    for (int u = 0; u < nsteps; ++u) {
        const int t = u / nslab, sl = u - t * nslab;

        // ---- B operands (queries) for this slab, BEFORE the prefetch is issued: the counted
        // wait for them then leaves the younger prefetch loads in flight
        const int ns = FULL ? 16 : min(16, (p.dpad - sl * KS) >> 3); // 8-wide k steps in this slab
        if (FULL) {
            if (!SINGLE && u > 0) {
#pragma unroll
                for (int s = 0; s < 16; ++s) bq[s] = *(const f32x4*)(qrow + sl * KS + 8 * s + 4 * h);
            }
        } else {
            if (nslab > 1 || u == 0) {
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    if (s < ns) bq[s] = *(const f32x4*)(qrow + sl * KS + 8 * s + 4 * h);
                }
            }
        }
        if (u + 1 < nsteps) stage_load(u + 1);

        if (sl == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc0[r] = 0.f;
                acc1[r] = 0.f;
            }
        }
        // ---- MFMA: 2 row blocks x ns x 4
        {
            const char* tile = smem + LDS_TILES + (u & 1) * TILE_BYTES;
            const char* rowp0 = tile + j * 512;        // block 0: row j
            const char* rowp1 = tile + (32 + j) * 512; // block 1: row 32 + j  ((32+j)&15 == j&15)
            const int sw = j & 15;
            if (FULL) {
                // fragments of step s+1 are read (into the other register pair) before the MFMAs of
                // step s are issued, so the LDS latency hides under 8 x 64 matrix-pipe cycles
                f32x4 aA0 = *(const f32x4*)(rowp0 + (((0 + h) ^ sw) << 4));
                f32x4 aA1 = *(const f32x4*)(rowp1 + (((0 + h) ^ sw) << 4));
                f32x4 aB0, aB1;
#pragma unroll
                for (int s = 0; s < 16; s += 2) {
                    {
                        const int off = (((2 * (s + 1) + h) ^ sw) << 4);
                        aB0 = *(const f32x4*)(rowp0 + off);
                        aB1 = *(const f32x4*)(rowp1 + off);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(aA0[e], bq[s][e], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(aA1[e], bq[s][e], acc1, 0, 0, 0);
                    }
                    if (s + 2 < 16) {
                        const int off = (((2 * (s + 2) + h) ^ sw) << 4);
                        aA0 = *(const f32x4*)(rowp0 + off);
                        aA1 = *(const f32x4*)(rowp1 + off);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(aB0[e], bq[s + 1][e], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(aB1[e], bq[s + 1][e], acc1, 0, 0, 0);
                    }
                }
            } else {
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    if (s < ns) {
                        const int off = (((2 * s + h) ^ sw) << 4);
                        f32x4 a0 = *(const f32x4*)(rowp0 + off);
                        f32x4 a1 = *(const f32x4*)(rowp1 + off);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], bq[s][e], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], bq[s][e], acc1, 0, 0, 0);
                        }
                    }
                }
            }
        }
        // ---- epilogue after the last slab of the tile
        if (sl == nslab - 1) {
            const float* bias = (const float*)(smem + LDS_BIAS) + (t & 1) * TR;
            const int tile_row0 = r0 + t * TR;
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 b4 = *(const f32x4*)(bias + blk * 32 + 8 * g + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * g + e;
                        const float ip = blk == 0 ? acc0[r] : acc1[r];
                        float dis;
                        if (METRIC == METRIC_L2) {
                            dis = __fmaf_rn(-2.f, ip, xn + b4[e]);
                            dis = dis < 0.f ? 0.f : dis;
                        } else {
                            dis = ip + b4[e];
                        }
                        const int grow = tile_row0 + blk * 32 + 8 * g + 4 * h + e;
                        if (DUMP) {
                            if (qvalid && grow < r1) p.dump[(int64_t)q * p.nb + grow] = dis;
                        } else {
                            const bool pass = METRIC == METRIC_L2 ? dis < tau : dis > tau;
                            const u64 m = __ballot(pass);
                            if (m) {
                                const int lo = (int)((m >> j) & 1ull);
                                const int hi = (int)((m >> (j + 32)) & 1ull);
                                if (pass) {
                                    const int slot = cnt + (h ? lo : 0);
                                    resq[slot] = ((u64)ordkey<METRIC>(dis) << 32) | (unsigned)grow;
                                }
                                cnt += lo + hi;
                            }
                        }
                    }
                }
                if (!DUMP) {
                    u64 flagged = __ballot(cnt > cap_lim) & 0xffffffffull;
                    if (flagged) {
                        wave_mem_sync();
                        while (flagged) {
                            const int jq = __ffsll((long long)flagged) - 1;
                            flagged &= flagged - 1;
                            const int n = __shfl(cnt, jq, 64);
                            u64* base = res_wave + (int64_t)jq * p.nsplit * p.cap;
                            const u64 kth = wave_select_kth(base, n, p.k, hist);
                            const int kept = wave_compact(base, n, kth);
                            if (j == jq) {
                                cnt = kept;
                                tau = unordkey<METRIC>((uint32_t)(kth >> 32));
                            }
                        }
                    }
                }
            }
        }
        if (u + 1 < nsteps) stage_store(u + 1);
        __syncthreads();
    }

    if (!DUMP) {
        // final compaction so the merge kernel sees at most k keys per (query, split)
        u64 flagged = __ballot(cnt > p.k) & 0xffffffffull;
        if (flagged) wave_mem_sync();
        while (flagged) {
            const int jq = __ffsll((long long)flagged) - 1;
            flagged &= flagged - 1;
            const int n = __shfl(cnt, jq, 64);
            u64* base = res_wave + (int64_t)jq * p.nsplit * p.cap;
            const u64 kth = wave_select_kth(base, n, p.k, hist);
            const int kept = wave_compact(base, n, kth);
            if (j == jq) cnt = kept;
        }
        if (qvalid && h == 0) p.res_cnt[(int64_t)q * p.nsplit + split] = (uint32_t)cnt;
    }
}

void launch_flat_scan(const FlatScanParams& p, hipStream_t stream) {
    if (p.nq == 0 || p.nb == 0) return;
    FA_THROW_IF_NOT(p.dpad % 8 == 0 && p.ldq % 4 == 0 && p.ldb % 4 == 0);
    FA_THROW_IF_NOT(p.rows_per_split % TR == 0);
    FA_THROW_IF_NOT(p.dump || p.cap >= p.k + 32);
    dim3 grid((unsigned)(p.nsplit * p.ngroups)), block(512);
    size_t lds = LDS_TOTAL;
    const bool full = (p.dpad % KS) == 0;
    const bool single = p.dpad == KS;
#define FA_LAUNCH(M, D, F, S)                                                                     \
    do {                                                                                          \
        HIP_CHECK(hipFuncSetAttribute((const void*)flat_scan_kernel<M, D, F, S>,                  \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));     \
        hipLaunchKernelGGL((flat_scan_kernel<M, D, F, S>), grid, block, lds, stream, p);          \
    } while (0)
#define FA_LAUNCH_F(M, D)                          \
    do {                                           \
        if (single) FA_LAUNCH(M, D, true, true);   \
        else if (full) FA_LAUNCH(M, D, true, false); \
        else FA_LAUNCH(M, D, false, false);        \
    } while (0)
    if (p.metric == METRIC_L2) {
        if (p.dump) FA_LAUNCH_F(METRIC_L2, true);
        else FA_LAUNCH_F(METRIC_L2, false);
    } else {
        if (p.dump) FA_LAUNCH_F(METRIC_INNER_PRODUCT, true);
        else FA_LAUNCH_F(METRIC_INNER_PRODUCT, false);
    }
#undef FA_LAUNCH_F
#undef FA_LAUNCH
    HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------
// scalar cross-check kernel: identical arithmetic, no MFMA, no filtering
// ---------------------------------------------------------------------------------
template <int METRIC>
__global__ void flat_simple_kernel(const float* __restrict__ xq, const float* __restrict__ xqn,
                                   int64_t ldq, int nq, const float* __restrict__ xb,
                                   const float* __restrict__ xbn, int64_t ldb, int nb, int dpad,
                                   u64* __restrict__ keys) {
    const int q = blockIdx.y;
    const float* qr = xq + (int64_t)q * ldq;
    const float xn = METRIC == METRIC_L2 ? xqn[q] : 0.f;
    for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < nb; row += gridDim.x * blockDim.x) {
        const float* yr = xb + (int64_t)row * ldb;
        float acc = 0.f;
        for (int s = 0; s < dpad; s += 8) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc = __fmaf_rn(yr[s + e], qr[s + e], acc);
                acc = __fmaf_rn(yr[s + 4 + e], qr[s + 4 + e], acc);
            }
        }
        float dis;
        if (METRIC == METRIC_L2) {
            dis = __fmaf_rn(-2.f, acc, xn + xbn[row]);
            dis = dis < 0.f ? 0.f : dis;
        } else {
            dis = acc;
        }
        keys[(int64_t)q * nb + row] = ((u64)ordkey<METRIC>(dis) << 32) | (unsigned)row;
    }
}

// ---------------------------------------------------------------------------------
// k = 1 over a small database (k-means assignment of a product-quantizer sub-space: 256 centroids of 2-8 dims, tens of
// thousands of points, thousands of times per training): the whole database and its norms sit in LDS, one thread per
// query walks it with the very chain / formula / tie rule of the scan kernels (first minimum = lowest id), one launch
// instead of scan + select (146 + 71 us per call in round 1 -- 0.35 s of a 0.85 s IVFPQ training).
// ---------------------------------------------------------------------------------
template <int METRIC>
__global__ void __launch_bounds__(256) flat_assign_small_kernel(const float* __restrict__ xq, int64_t ldq, int nq,
                                                                const float* __restrict__ xb,
                                                                const float* __restrict__ xbn, int64_t ldb, int nb,
                                                                int dpad, float* __restrict__ out_dis,
                                                                int64_t* __restrict__ out_ids) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* yb = (float*)smem;          // [nb][dpad]
    float* yn = yb + (size_t)nb * dpad; // [nb]
    for (int t = threadIdx.x; t < nb * dpad; t += 256) yb[t] = xb[(int64_t)(t / dpad) * ldb + (t % dpad)];
    for (int t = threadIdx.x; t < nb; t += 256) yn[t] = METRIC == METRIC_L2 ? xbn[t] : 0.f;
    __syncthreads();
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    const float* qr = xq + (int64_t)q * ldq;
    float qv[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) qv[c] = c < dpad ? qr[c] : 0.f;
    float xn = 0.f;
    if (METRIC == METRIC_L2)
        for (int c = 0; c < dpad; ++c) xn = __fmaf_rn(qv[c], qv[c], xn); // (sequential chain = l2_norms_kernel)
    unsigned best_key = 0xffffffffu;
    int best = -1;
    for (int row = 0; row < nb; ++row) {
        const float* yr = yb + row * dpad; // every lane the same address: LDS broadcast
        float acc = 0.f;
#pragma unroll
        for (int s = 0; s < 32; s += 8) {
            if (s < dpad) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc = __fmaf_rn(yr[s + e], qv[s + e], acc);
                    acc = __fmaf_rn(yr[s + 4 + e], qv[s + 4 + e], acc);
                }
            }
        }
        float dis;
        if (METRIC == METRIC_L2) {
            dis = __fmaf_rn(-2.f, acc, xn + yn[row]);
            dis = dis < 0.f ? 0.f : dis;
        } else {
            dis = acc;
        }
        const unsigned key = ordkey<METRIC>(dis);
        if (key < best_key) { // strict: the first (lowest id) of equal distances stays
            best_key = key;
            best = row;
        }
    }
    const bool ok = best >= 0 && best_key < kInvalidOrdKey;
    out_dis[q] = ok ? unordkey<METRIC>(best_key) : neutral_distance(METRIC);
    out_ids[q] = ok ? best : -1;
}
bool flat_assign_small_supported(int nb, int dpad) {
    return nb >= 1 && dpad <= 32 && (size_t)nb * (dpad + 1) * 4 <= 48 * 1024;
}
void launch_flat_assign_small(int metric, const float* xq, int64_t ldq, int nq, const float* xb, const float* xbn, int64_t ldb,
                              int nb, int dpad, float* out_dis, int64_t* out_ids, hipStream_t stream) {
    if (nq == 0) return;
    FA_THROW_IF_NOT(flat_assign_small_supported(nb, dpad));
    const size_t lds = (size_t)nb * (dpad + 1) * 4;
    const dim3 grid((unsigned)div_up(nq, 256));
    if (metric == METRIC_L2)
        hipLaunchKernelGGL((flat_assign_small_kernel<METRIC_L2>), grid, dim3(256), lds, stream, xq, ldq, nq, xb, xbn, ldb, nb,
                           dpad, out_dis, out_ids);
    else
        hipLaunchKernelGGL((flat_assign_small_kernel<METRIC_INNER_PRODUCT>), grid, dim3(256), lds, stream, xq, ldq, nq, xb, xbn,
                           ldb, nb, dpad, out_dis, out_ids);
    HIP_CHECK(hipGetLastError());
}

void launch_flat_simple(int metric, const float* xq, const float* xqn, int64_t ldq, int nq,
                        const float* xb, const float* xbn, int64_t ldb, int nb, int dpad, u64* keys,
                        hipStream_t stream) {
    if (nq == 0 || nb == 0) return;
    dim3 grid((unsigned)std::min<int64_t>(div_up(nb, 256), 1024), (unsigned)nq), block(256);
    if (metric == METRIC_L2)
        hipLaunchKernelGGL((flat_simple_kernel<METRIC_L2>), grid, block, 0, stream, xq, xqn, ldq, nq, xb,
                           xbn, ldb, nb, dpad, keys);
    else
        hipLaunchKernelGGL((flat_simple_kernel<METRIC_INNER_PRODUCT>), grid, block, 0, stream, xq, xqn,
                           ldq, nq, xb, xbn, ldb, nb, dpad, keys);
    HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------
// "extra" metrics (faiss/gpu/impl/DistanceUtils.cuh:47-281 functors, faiss/utils/simd_impl/distances_autovec-inl.h:177-262
// on the CPU): no inner-product form, so no matrix pipe -- one thread per (query, row), the query in LDS, ONE sequential
// fp32 pass over the dimensions with every operation rounded once (IEEE division spelled out), which is what
// oracle/faiss_oracle.c general_distance restates.  All distances go to memory as keys and the select kernel picks
// the k best: the brute-force layout of the reference's general-distance path, good for the few GFLOP these metrics
// are used at.
// ---------------------------------------------------------------------------------
template <int GM>
__global__ void __launch_bounds__(256) flat_general_kernel(const float* __restrict__ xq, int64_t ldq, int nq,
                                                           const float* __restrict__ xb, int64_t ldb, int nb, int d,
                                                           float arg, u64* __restrict__ keys) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* qs = (float*)smem; // [d]
    const int q = blockIdx.y;
    for (int c = threadIdx.x; c < d; c += blockDim.x) qs[c] = xq[(int64_t)q * ldq + c];
    __syncthreads();
    for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < nb; row += gridDim.x * blockDim.x) {
        const float* yr = xb + (int64_t)row * ldb;
        float a = 0.f, b = 0.f;
        for (int i = 0; i < d; ++i) {
            const float xi = qs[i], yi = yr[i];
            if (GM == METRIC_L1) {
                a = a + fabsf(xi - yi);
            } else if (GM == METRIC_Linf) {
                a = fmaxf(a, fabsf(xi - yi));
            } else if (GM == METRIC_Lp) {
                a = a + powf(fabsf(xi - yi), arg);
            } else if (GM == METRIC_Canberra) {
                a = a + __fdiv_rn(fabsf(xi - yi), fabsf(xi) + fabsf(yi));
            } else if (GM == METRIC_BrayCurtis) {
                a = a + fabsf(xi - yi);
                b = b + fabsf(xi + yi);
            } else if (GM == METRIC_JensenShannon) {
                const float m = 0.5f * (xi + yi);
                const float kl1 = -xi * logf(__fdiv_rn(m, xi));
                const float kl2 = -yi * logf(__fdiv_rn(m, yi));
                a = a + (kl1 + kl2);
            } else { // METRIC_Jaccard
                a = a + fminf(xi, yi);
                b = b + fmaxf(xi, yi);
            }
        }
        float dis = a;
        if (GM == METRIC_BrayCurtis || GM == METRIC_Jaccard) dis = __fdiv_rn(a, b);
        if (GM == METRIC_JensenShannon) dis = 0.5f * a;
        const uint32_t ok = GM == METRIC_Jaccard ? ordkey<METRIC_INNER_PRODUCT>(dis) : ordkey<METRIC_L2>(dis);
        keys[(int64_t)q * nb + row] = ((u64)ok << 32) | (unsigned)row;
    }
}

void launch_flat_general(int metric, float metric_arg, const float* xq, int64_t ldq, int nq, const float* xb, int64_t ldb,
                         int nb, int d, u64* keys, hipStream_t stream) {
    if (nq == 0 || nb == 0) return;
    FA_THROW_IF_NOT_MSG(is_general_metric(metric), "not one of the extra metrics");
    FA_THROW_IF_NOT_MSG(nq <= 65535, "query tile too large for the general-distance kernel");
    dim3 grid((unsigned)std::min<int64_t>(div_up(nb, 256), 1024), (unsigned)nq), block(256);
    const size_t lds = (size_t)d * 4;
#define FA_GEN(M_)                                                                                                   \
    case M_:                                                                                                         \
        hipLaunchKernelGGL((flat_general_kernel<M_>), grid, block, lds, stream, xq, ldq, nq, xb, ldb, nb, d, metric_arg, \
                           keys);                                                                                    \
        break
    switch (metric) {
        FA_GEN(METRIC_L1);
        FA_GEN(METRIC_Linf);
        FA_GEN(METRIC_Lp);
        FA_GEN(METRIC_Canberra);
        FA_GEN(METRIC_BrayCurtis);
        FA_GEN(METRIC_JensenShannon);
        default: FA_GEN(METRIC_Jaccard);
    }
#undef FA_GEN
    HIP_CHECK(hipGetLastError());
}

} // namespace faiss_amd
