// faiss_amd/csrc/ivf_kernels.hip -- inverted-file scans (IVFFlat, IVFPQ) and the add path, gfx950.
//
// Layout (kernels.h, "IVF storage"): every inverted list owns a row range with slack of one device arena
// (list_start[l], list_len[l], capacity a multiple of the granule); vectors row-major fp32 (IVFFlat) or PQ codes in
// rotated 64-row blocks (IVFPQ, pq_code_offset), user ids (and the IVFPQ L2 term t2) in parallel per-row arenas.
// (The reference keeps one growable DeviceVector per list, faiss/gpu/impl/IVFBase.cuh:220-299; a single arena
// costs one allocation and lets 288 GB of HBM be sized once.)
//
// Arithmetic contract (restated by oracle/faiss_oracle.c):
//   IVFFlat L2 : eight partial fmaf chains of (q[k]-y[k])^2: chain ln runs over the 4-float chunks ln, ln+8, ...
//                (direct form as the CPU scanner, faiss/utils/simd_impl/IVFFlatScanner-inl.h:20-27, which also
//                keeps 8 partial sums); dis = ((p0+p1)+(p2+p3)) + ((p4+p5)+(p6+p7))
//   IVFFlat IP : the same with chains of fmaf(q[k], y[k], acc)
//   IVFPQ      : lut[m][c] = chain_j fmaf(q_mj, pq[m][c][j], acc)   (one table per query, both metrics), rounded
//                to the query's power-of-two grid (kernels.h pq_lut_grid); S = sum_m lut[m][c_m], exact in fp32
//                in any order
//   IVFPQ  L2  : dis = fmaf(-2, S, coarse_l2 + t2),  t2 = chain_k fmaf(r^_k, fmaf(2, c_k, r^_k), acc)
//                (r^ = decoded residual, c = list centroid; term decomposition of
//                 faiss/impl/pq_code_distance/IVFPQ_QueryTables.cpp:126-192)
//                (faiss/gpu/impl/PQCodeDistances-inl.cuh:29-285 semantics, by_residual,
//                 no precomputed table; CPU counterpart faiss/IndexIVFPQ.cpp scan_list_with_table)
//   IVFPQ  IP  : dis = coarse_ip + S
//   PQ encode  : code[m] = first argmin_c of the L2 lut expression above
//                (faiss/impl/ProductQuantizer.cpp compute_code)
#include "kernels.h"

namespace faiss_amd {

typedef unsigned long long u64;

// ---------------------------------------------------------------------------------
__global__ void ivf_prefix_kernel(const int64_t* __restrict__ coarse_ids, int nq, int nprobe,
                                  const uint32_t* __restrict__ list_len, uint32_t* __restrict__ prefix,
                                  uint32_t* __restrict__ total) {
    int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    uint32_t acc = 0;
    uint32_t* pr = prefix + (int64_t)q * (nprobe + 1);
    for (int p = 0; p < nprobe; ++p) {
        pr[p] = acc;
        int64_t l = coarse_ids[(int64_t)q * nprobe + p];
        if (l >= 0) acc += list_len[l];
    }
    pr[nprobe] = acc;
    total[q] = acc;
}

// caller-supplied probe lists (search_preassigned): anything outside [0, nlist) becomes -1 = "no list"
__global__ void ivf_sanitize_assign_kernel(int64_t* __restrict__ ids, int64_t n, int nlist) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const int64_t l = ids[i];
        if (l < 0 || l >= nlist) ids[i] = -1;
    }
}
void launch_ivf_sanitize_assign(int64_t* ids, int64_t n, int nlist, hipStream_t stream) {
    if (n == 0) return;
    hipLaunchKernelGGL(ivf_sanitize_assign_kernel, dim3((unsigned)div_up(n, 256)), dim3(256), 0, stream, ids, n, nlist);
    HIP_CHECK(hipGetLastError());
}

__global__ void ivf_probe_info_kernel(const int64_t* __restrict__ coarse_ids, int64_t n,
                                      const uint32_t* __restrict__ list_len, const int64_t* __restrict__ list_start,
                                      uint32_t* __restrict__ probe_len, int64_t* __restrict__ probe_start) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t l = coarse_ids[i];
    probe_len[i] = l >= 0 ? list_len[l] : 0u;
    probe_start[i] = l >= 0 ? list_start[l] : 0;
}
void launch_ivf_probe_info(const int64_t* coarse_ids, int64_t n, const uint32_t* list_len, const int64_t* list_start,
                           uint32_t* probe_len, int64_t* probe_start, hipStream_t stream) {
    if (n == 0) return;
    hipLaunchKernelGGL(ivf_probe_info_kernel, dim3((unsigned)div_up(n, 256)), dim3(256), 0, stream, coarse_ids, n, list_len,
                       list_start, probe_len, probe_start);
    HIP_CHECK(hipGetLastError());
}

void launch_ivf_prefix(const int64_t* coarse_ids, int nq, int nprobe, const uint32_t* list_len,
                       uint32_t* prefix, uint32_t* total, hipStream_t stream) {
    if (nq == 0) return;
    hipLaunchKernelGGL(ivf_prefix_kernel, dim3((unsigned)div_up(nq, 128)), dim3(128), 0, stream,
                       coarse_ids, nq, nprobe, list_len, prefix, total);
    HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------
// IVFFlat scan: one workgroup per (probe, query)
// ---------------------------------------------------------------------------------
template <int METRIC>
__global__ void __launch_bounds__(256) ivfflat_scan_kernel(IvfScanParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* qs = (float*)smem; // [dpad]
    const int pr = blockIdx.x, q = blockIdx.y;
    const int64_t list = p.coarse_ids[(int64_t)q * p.nprobe + pr];
    if (list < 0) return;
    const unsigned len = p.list_len[list];
    if (len == 0) return;
    const int64_t start = p.list_start[list];
    const uint32_t pos0 = p.prefix[(int64_t)q * (p.nprobe + 1) + pr];
    u64* out = p.keys + p.q_off[q] + pos0;
    for (int c = threadIdx.x; c < p.dpad; c += blockDim.x) qs[c] = p.xq[(int64_t)q * p.ldq + c];
    __syncthreads();
    for (unsigned i = threadIdx.x; i < len; i += blockDim.x) {
        const float* y = p.arena_vecs + (start + i) * p.ldv;
        // same order as the fused scan (ivf_fused.hip): partial sum ln over the 4-float chunks
        // ln, ln+8, ... and a pairwise combination of the eight partial sums
        float part[8];
#pragma unroll
        for (int ln = 0; ln < 8; ++ln) {
            float a = 0.f;
            for (int k = ln * 4; k < p.dpad; k += 32) {
                const float4 yv = *(const float4*)(y + k);
                const float4 qv = *(const float4*)(qs + k);
                if (METRIC == METRIC_L2) {
                    float t;
                    t = qv.x - yv.x; a = __fmaf_rn(t, t, a);
                    t = qv.y - yv.y; a = __fmaf_rn(t, t, a);
                    t = qv.z - yv.z; a = __fmaf_rn(t, t, a);
                    t = qv.w - yv.w; a = __fmaf_rn(t, t, a);
                } else {
                    a = __fmaf_rn(qv.x, yv.x, a);
                    a = __fmaf_rn(qv.y, yv.y, a);
                    a = __fmaf_rn(qv.z, yv.z, a);
                    a = __fmaf_rn(qv.w, yv.w, a);
                }
            }
            part[ln] = a;
        }
        const float acc = ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
        // (a row an IDSelector excludes keeps its slot with a key no result can have)
        const bool keep = !p.sel_mask || ((p.sel_mask[(start + i) >> 5] >> ((start + i) & 31)) & 1u);
        out[i] = ((u64)(keep ? ordkey<METRIC>(acc) : 0xffffffffu) << 32) | (u64)(pos0 + i);
    }
}

void launch_ivfflat_scan(const IvfScanParams& p, hipStream_t stream) {
    if (p.nq == 0 || p.nprobe == 0) return;
    dim3 grid((unsigned)p.nprobe, (unsigned)p.nq), block(256);
    size_t lds = (size_t)p.dpad * 4;
    if (p.metric == METRIC_L2)
        hipLaunchKernelGGL((ivfflat_scan_kernel<METRIC_L2>), grid, block, lds, stream, p);
    else
        hipLaunchKernelGGL((ivfflat_scan_kernel<METRIC_INNER_PRODUCT>), grid, block, lds, stream, p);
    HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------
// IVFPQ scan (unfused cross-check path): one workgroup per (probe, query); LUT [M][256] fp32 resident in LDS
// ---------------------------------------------------------------------------------
size_t ivfpq_scan_lds_bytes(int M, int dpad) {
    return (size_t)M * 256 * 4 + (size_t)dpad * 4 + (size_t)M * 4 + 16;
}

template <int METRIC>
__global__ void __launch_bounds__(256) ivfpq_scan_kernel(IvfScanParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* lut = (float*)smem;           // [M][256]
    float* rs = lut + (size_t)p.M * 256; // [dpad] query
    unsigned* colmax = (unsigned*)(rs + p.dpad); // [M] bits of max_c |lut[m][c]|
    float* grid = (float*)(colmax + p.M);        // delta, 1/delta, flag
    const int pr = blockIdx.x, q = blockIdx.y;
    const int64_t list = p.coarse_ids[(int64_t)q * p.nprobe + pr];
    if (list < 0) return;
    const unsigned len = p.list_len[list];
    if (len == 0) return;
    const int64_t start = p.list_start[list];
    const uint32_t pos0 = p.prefix[(int64_t)q * (p.nprobe + 1) + pr];
    u64* out = p.keys + p.q_off[q] + pos0;
    const int d = p.d, M = p.M;
    for (int c = threadIdx.x; c < d; c += blockDim.x) rs[c] = p.xq[(int64_t)q * p.ldq + c];
    for (int m = threadIdx.x; m < M; m += blockDim.x) colmax[m] = 0u;
    __syncthreads();
    // ---- lookup table of the query, then its grid (pq_lut_grid)
    const int dsub = p.dsub;
    for (int e = threadIdx.x; e < M * 256; e += blockDim.x) {
        const int m = e >> 8;
        const float* cen = p.pq_centroids + (size_t)e * dsub; // [m][c][dsub]
        const float* r = rs + m * dsub;
        float acc = 0.f;
        for (int jd = 0; jd < dsub; ++jd) acc = __fmaf_rn(r[jd], cen[jd], acc);
        lut[e] = acc;
        atomicMax(&colmax[m], __float_as_uint(fabsf(acc)));
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float B = 0.f;
        for (int m = 0; m < M; ++m) B = B + __uint_as_float(colmax[m]);
        float delta = 0.f, inv = 0.f;
        const bool ok = pq_lut_grid(B, &delta, &inv);
        grid[0] = delta;
        grid[1] = inv;
        grid[2] = ok ? 1.f : 0.f;
    }
    __syncthreads();
    if (grid[2] != 0.f) {
        const float delta = grid[0], inv = grid[1];
        for (int e = threadIdx.x; e < M * 256; e += blockDim.x) lut[e] = __builtin_rintf(lut[e] * inv) * delta;
    }
    __syncthreads();
    // ---- scan: one code per thread
    const float dis0 = p.coarse_dis[(int64_t)q * p.nprobe + pr];
    for (unsigned i = threadIdx.x; i < len; i += blockDim.x) {
        const int64_t row = start + i;
        float sum = 0.f;
        for (int m = 0; m < M; ++m) sum = sum + lut[m * 256 + p.arena_codes[pq_code_offset(M, row, m)]];
        const float acc = METRIC == METRIC_L2 ? __fmaf_rn(-2.f, sum, dis0 + p.arena_t2[row]) : dis0 + sum;
        const bool keep = !p.sel_mask || ((p.sel_mask[row >> 5] >> (row & 31)) & 1u);
        out[i] = ((u64)(keep ? ordkey<METRIC>(acc) : 0xffffffffu) << 32) | (u64)(pos0 + i);
    }
}

void launch_ivfpq_scan(const IvfScanParams& p, hipStream_t stream) {
    if (p.nq == 0 || p.nprobe == 0) return;
    dim3 grid((unsigned)p.nprobe, (unsigned)p.nq), block(256);
    size_t lds = ivfpq_scan_lds_bytes(p.M, p.dpad);
    if (p.metric == METRIC_L2) {
        HIP_CHECK(hipFuncSetAttribute((const void*)ivfpq_scan_kernel<METRIC_L2>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((ivfpq_scan_kernel<METRIC_L2>), grid, block, lds, stream, p);
    } else {
        HIP_CHECK(hipFuncSetAttribute((const void*)ivfpq_scan_kernel<METRIC_INNER_PRODUCT>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((ivfpq_scan_kernel<METRIC_INNER_PRODUCT>), grid, block, lds, stream, p);
    }
    HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------
// add path: stable counting sort of the new vectors by coarse label, entirely on the device
// ---------------------------------------------------------------------------------
__global__ void ivf_histogram_kernel(const int64_t* __restrict__ labels, int64_t n, int nlist, int chunk,
                                     uint32_t* __restrict__ hist) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t l = labels[i];
    if (l >= 0 && l < nlist) atomicAdd(&hist[(i / chunk) * nlist + l], 1u);
}
void launch_ivf_histogram(const int64_t* labels, int64_t n, int nlist, int chunk, uint32_t* hist, hipStream_t stream) {
    if (n == 0) return;
    hipLaunchKernelGGL(ivf_histogram_kernel, dim3((unsigned)div_up(n, 256)), dim3(256), 0, stream, labels, n, nlist,
                       chunk, hist);
    HIP_CHECK(hipGetLastError());
}

__global__ void ivf_chunk_scan_kernel(uint32_t* __restrict__ hist, int nchunks, int nlist,
                                      const uint32_t* __restrict__ list_len, uint32_t* __restrict__ new_len) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= nlist) return;
    uint32_t run = list_len[l];
    // eight chunks per round trip: the loads of a group are independent of the running offset
    int c = 0;
    for (; c + 8 <= nchunks; c += 8) {
        uint32_t t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = hist[(int64_t)(c + u) * nlist + l];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            hist[(int64_t)(c + u) * nlist + l] = run;
            run += t[u];
        }
    }
    for (; c < nchunks; ++c) {
        const uint32_t t = hist[(int64_t)c * nlist + l];
        hist[(int64_t)c * nlist + l] = run;
        run += t;
    }
    new_len[l] = run;
}
void launch_ivf_chunk_scan(uint32_t* hist, int nchunks, int nlist, const uint32_t* list_len, uint32_t* new_len,
                           hipStream_t stream) {
    hipLaunchKernelGGL(ivf_chunk_scan_kernel, dim3((unsigned)div_up(nlist, 256)), dim3(256), 0, stream, hist, nchunks,
                       nlist, list_len, new_len);
    HIP_CHECK(hipGetLastError());
}

// One wavefront per chunk walks its vectors 64 at a time, in order: the lanes that share a label take consecutive
// slots behind the chunk's running offset for that list (ballot prefix), so entries keep their insertion order
// (what testIVFEquality compares, faiss/gpu/test/TestUtils.h:111-142).  IN_LDS: the chunk's offset row lives in LDS.
template <bool IN_LDS>
__global__ void __launch_bounds__(64) ivf_rank_kernel(const int64_t* __restrict__ labels, int64_t n, int nlist,
                                                      int chunk, uint32_t* __restrict__ hist,
                                                      const int64_t* __restrict__ list_start,
                                                      int64_t* __restrict__ dest) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int c = blockIdx.x, lane = threadIdx.x;
    uint32_t* grow = hist + (int64_t)c * nlist;
    uint32_t* row = IN_LDS ? (uint32_t*)smem : grow;
    if (IN_LDS) {
        for (int l = lane; l < nlist; l += 64) row[l] = grow[l];
        __syncthreads();
    }
    const int64_t i0 = (int64_t)c * chunk, i1 = min(n, i0 + chunk);
    for (int64_t g0 = i0; g0 < i1; g0 += 64) {
        const int64_t i = g0 + lane;
        int64_t lab = i < i1 ? labels[i] : -1;
        if (lab >= nlist) lab = -1;
        int64_t dst = -1;
        u64 todo = __ballot(lab >= 0);
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            const int l0 = __shfl((int)lab, src, 64);
            const u64 m = __ballot((int)lab == l0 && lab >= 0);
            const uint32_t base = row[l0];
            if (lab >= 0 && (int)lab == l0) dst = list_start[l0] + base + __popcll(m & ((1ull << lane) - 1ull));
            if (lane == src) row[l0] = base + (uint32_t)__popcll(m);
            todo &= ~m;
        }
        if (i < i1) dest[i] = dst;
    }
}
void launch_ivf_rank(const int64_t* labels, int64_t n, int nlist, int chunk, uint32_t* hist,
                     const int64_t* list_start, int64_t* dest, hipStream_t stream) {
    if (n == 0) return;
    const unsigned nchunks = (unsigned)div_up(n, chunk);
    if ((size_t)nlist * 4 <= 64 * 1024) {
        hipLaunchKernelGGL((ivf_rank_kernel<true>), dim3(nchunks), dim3(64), (size_t)nlist * 4, stream, labels, n, nlist,
                           chunk, hist, list_start, dest);
    } else {
        hipLaunchKernelGGL((ivf_rank_kernel<false>), dim3(nchunks), dim3(64), 0, stream, labels, n, nlist, chunk, hist,
                           list_start, dest);
    }
    HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------
// k-means centroid update on the device (faiss/Clustering.cpp:307-324 compute_centroids): the points are
// counting-sorted by their assignment with the kernels above (stable: a cluster's members stay in index order), then
// one thread per (centroid, dimension) adds its members in that order in double precision -- the same additions in
// the same order as a sequential host loop, whatever the launch geometry.
// ---------------------------------------------------------------------------------
// start[l] = sum_{l' < l} cnt[l'] for l in [0, n]: one workgroup, segments of ceil(n / 1024) per thread
__global__ void __launch_bounds__(1024) exclusive_scan_kernel(const uint32_t* __restrict__ cnt, int n,
                                                             int64_t* __restrict__ start) {
    __shared__ int64_t part[1024];
    const int t = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int a = min(n, t * per), b = min(n, a + per);
    int64_t s = 0;
    for (int i = a; i < b; ++i) s += cnt[i];
    part[t] = s;
    __syncthreads();
    if (t == 0) {
        int64_t run = 0;
        for (int i = 0; i < 1024; ++i) {
            const int64_t v = part[i];
            part[i] = run;
            run += v;
        }
        start[n] = run;
    }
    __syncthreads();
    int64_t run = part[t];
    for (int i = a; i < b; ++i) {
        start[i] = run;
        run += cnt[i];
    }
}
void launch_exclusive_scan(const uint32_t* cnt, int n, int64_t* start, hipStream_t stream) {
    hipLaunchKernelGGL(exclusive_scan_kernel, dim3(1), dim3(1024), 0, stream, cnt, n, start);
    HIP_CHECK(hipGetLastError());
}
__global__ void invert_dest_kernel(const int64_t* __restrict__ dest, int64_t n, uint32_t* __restrict__ order) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && dest[i] >= 0) order[dest[i]] = (uint32_t)i;
}
void launch_invert_dest(const int64_t* dest, int64_t n, uint32_t* order, hipStream_t stream) {
    if (n == 0) return;
    hipLaunchKernelGGL(invert_dest_kernel, dim3((unsigned)div_up(n, 256)), dim3(256), 0, stream, dest, n, order);
    HIP_CHECK(hipGetLastError());
}
__global__ void kmeans_update_kernel(const float* __restrict__ x, int64_t ldx, int d, const uint32_t* __restrict__ order,
                                     const int64_t* __restrict__ start, const uint32_t* __restrict__ cnt,
                                     float* __restrict__ cen) {
    const int c = blockIdx.x;
    const uint32_t n = cnt[c];
    if (n == 0) return; // an empty cluster keeps its centroid (split on the host)
    const uint32_t* mem = order + start[c];
    for (int j = threadIdx.x; j < d; j += blockDim.x) {
        double s = 0.0;
        for (uint32_t t = 0; t < n; ++t) s += (double)x[(int64_t)mem[t] * ldx + j];
        cen[(int64_t)c * d + j] = (float)(s / (double)n);
    }
}
void launch_kmeans_update(const float* x, int64_t ldx, int d, const uint32_t* order, const int64_t* start,
                          const uint32_t* cnt, int k, float* centroids, hipStream_t stream) {
    const int threads = d >= 128 ? 128 : 64;
    hipLaunchKernelGGL(kmeans_update_kernel, dim3((unsigned)k), dim3(threads), 0, stream, x, ldx, d, order, start, cnt,
                       centroids);
    HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------
// Product-quantizer training as ONE k-means over all M sub-spaces (round 3): point (m, i) is the dsub columns
// [m dsub, (m+1) dsub) of residual row i, its label m * 256 + (nearest of the 256 centroids of sub-space m); the counting
// sort and the update above then run once per iteration over M * nt points and M * 256 clusters instead of M times.
// Every point sees the arithmetic of the per-sub-space loop (flat_assign_small_kernel: chains over the sub-vector padded
// to a multiple of 8 with zeros, norms as sequential chains, first minimum wins; kmeans_update_kernel: members added in
// index order in double precision): the codebook is bit-identical (tests/test_gpu_extras.py).
// Reference: faiss/impl/ProductQuantizer.cpp:140-190 (one Clustering per sub-quantizer), faiss/gpu/GpuIndexIVFPQ.cu:287-340.
// ---------------------------------------------------------------------------------
__global__ void pq_train_init_kernel(const float* __restrict__ res, int64_t ld, int M, int dsub,
                                     const uint32_t* __restrict__ sel, float* __restrict__ cen) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x; // (m, c, j)
    if (i >= M * 256 * dsub) return;
    const int j = i % dsub, mc = i / dsub, m = mc >> 8;
    cen[i] = res[(int64_t)sel[mc] * ld + (int64_t)m * dsub + j];
}
void launch_pq_train_init(const float* res, int64_t ld, int M, int dsub, const uint32_t* sel, float* cen, hipStream_t stream) {
    hipLaunchKernelGGL(pq_train_init_kernel, dim3((unsigned)div_up((size_t)M * 256 * dsub, 256)), dim3(256), 0, stream, res, ld,
                       M, dsub, sel, cen);
    HIP_CHECK(hipGetLastError());
}
// grid (point tiles of 256, M); the 256 centroids of the sub-space and their norms in LDS, one thread per point
__global__ void __launch_bounds__(256) pq_train_assign_kernel(const float* __restrict__ res, int64_t ld, int64_t nt, int dsub,
                                                              int dpad, const float* __restrict__ cen,
                                                              int64_t* __restrict__ labels) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* yb = (float*)smem;              // [256][dpad], zero padded
    float* yn = yb + (size_t)256 * dpad;   // [256]
    const int m = blockIdx.y;
    const float* cm = cen + (size_t)m * 256 * dsub;
    for (int t = threadIdx.x; t < 256 * dpad; t += 256) {
        const int c = t % dpad;
        yb[t] = c < dsub ? cm[(size_t)(t / dpad) * dsub + c] : 0.f;
    }
    __syncthreads();
    {
        // |y|^2 as l2_norms_kernel computes it for the rows of a flat index: one sequential chain
        float acc = 0.f;
        for (int c = 0; c < dpad; ++c) acc = __fmaf_rn(yb[threadIdx.x * dpad + c], yb[threadIdx.x * dpad + c], acc);
        yn[threadIdx.x] = acc;
    }
    __syncthreads();
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= nt) return;
    const float* qr = res + q * ld + (int64_t)m * dsub;
    float qv[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) qv[c] = c < dsub ? qr[c] : 0.f;
    float xn = 0.f;
    for (int c = 0; c < dpad; ++c) xn = __fmaf_rn(qv[c], qv[c], xn);
    unsigned best_key = 0xffffffffu;
    int best = -1;
    for (int row = 0; row < 256; ++row) {
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
        float dis = __fmaf_rn(-2.f, acc, xn + yn[row]);
        dis = dis < 0.f ? 0.f : dis;
        const unsigned key = ordkey<METRIC_L2>(dis);
        if (key < best_key) { // strict: the first (lowest id) of equal distances stays
            best_key = key;
            best = row;
        }
    }
    const bool ok = best >= 0 && best_key < kInvalidOrdKey;
    labels[(int64_t)m * nt + q] = ok ? (int64_t)m * 256 + best : -1;
}
bool pq_train_batched_supported(int dsub) {
    return dsub >= 1 && dsub <= 32;
}
void launch_pq_train_assign(const float* res, int64_t ld, int64_t nt, int M, int dsub, const float* cen, int64_t* labels,
                            hipStream_t stream) {
    FA_THROW_IF_NOT(pq_train_batched_supported(dsub));
    const int dpad = (int)round_up(dsub, 8);
    const size_t lds = (size_t)256 * (dpad + 1) * 4;
    hipLaunchKernelGGL(pq_train_assign_kernel, dim3((unsigned)div_up((size_t)nt, 256), (unsigned)M), dim3(256), lds, stream, res,
                       ld, nt, dsub, dpad, cen, labels);
    HIP_CHECK(hipGetLastError());
}
// one workgroup per cluster (m, c): thread j adds coordinate j of the members in index order, in double precision
__global__ void pq_train_update_kernel(const float* __restrict__ res, int64_t ld, int64_t nt, int dsub,
                                       const uint32_t* __restrict__ order, const int64_t* __restrict__ start,
                                       const uint32_t* __restrict__ cnt, float* __restrict__ cen) {
    const int c = blockIdx.x;
    const uint32_t n = cnt[c];
    if (n == 0) return; // an empty cluster keeps its centroid (split on the host)
    const int m = c >> 8;
    const uint32_t* mem = order + start[c];
    const float* base = res + (int64_t)m * dsub - (int64_t)m * nt * ld; // point p of sub-space m = row p - m nt
    for (int j = threadIdx.x; j < dsub; j += blockDim.x) {
        double s = 0.0;
        for (uint32_t t = 0; t < n; ++t) s += (double)base[(int64_t)mem[t] * ld + j];
        cen[(int64_t)c * dsub + j] = (float)(s / (double)n);
    }
}
void launch_pq_train_update(const float* res, int64_t ld, int64_t nt, int M, int dsub, const uint32_t* order,
                            const int64_t* start, const uint32_t* cnt, float* cen, hipStream_t stream) {
    hipLaunchKernelGGL(pq_train_update_kernel, dim3((unsigned)(M * 256)), dim3(64), 0, stream, res, ld, nt, dsub, order, start,
                       cnt, cen);
    HIP_CHECK(hipGetLastError());
}

__global__ void ivf_move_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                const IvfMoveJob* __restrict__ jobs, int bytes_per_row) {
    const IvfMoveJob jb = jobs[blockIdx.x];
    const int64_t words = jb.rows * (bytes_per_row >> 2);
    const uint32_t* s = (const uint32_t*)(src + jb.src * bytes_per_row);
    uint32_t* t = (uint32_t*)(dst + jb.dst * bytes_per_row);
    for (int64_t w = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; w < words; w += (int64_t)gridDim.y * blockDim.x)
        t[w] = s[w];
}
void launch_ivf_move(const uint8_t* arena_src, uint8_t* arena_dst, const IvfMoveJob* jobs, int njobs,
                     int bytes_per_row, hipStream_t stream) {
    if (njobs == 0) return;
    FA_THROW_IF_NOT(bytes_per_row % 4 == 0);
    // few long lists (IVF16 at 100M rows) as well as many short ones: spread each job over several workgroups
    const unsigned gy = njobs >= 1024 ? 1u : njobs >= 64 ? 8u : 64u;
    hipLaunchKernelGGL(ivf_move_kernel, dim3((unsigned)njobs, gy), dim3(256), 0, stream, arena_src, arena_dst, jobs,
                       bytes_per_row);
    HIP_CHECK(hipGetLastError());
}

__global__ void iota_i64_kernel(int64_t* __restrict__ out, int64_t n, int64_t base) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = base + i;
}
void launch_iota_i64(int64_t* out, int64_t n, int64_t base, hipStream_t stream) {
    if (n == 0) return;
    hipLaunchKernelGGL(iota_i64_kernel, dim3((unsigned)div_up(n, 256)), dim3(256), 0, stream, out, n, base);
    HIP_CHECK(hipGetLastError());
}
__global__ void fill_knn_kernel(float* __restrict__ D, int64_t* __restrict__ I, int64_t n, float pad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        D[i] = pad;
        I[i] = -1;
    }
}
void launch_fill_knn(float* D, int64_t* I, int64_t n, int metric, hipStream_t stream) {
    if (n == 0) return;
    hipLaunchKernelGGL(fill_knn_kernel, dim3((unsigned)div_up(n, 256)), dim3(256), 0, stream, D, I, n,
                       neutral_distance(metric));
    HIP_CHECK(hipGetLastError());
}

__global__ void ivfflat_append_kernel(const float* __restrict__ x, int64_t ldx, int n, int d,
                                      const int64_t* __restrict__ dest, float* __restrict__ arena,
                                      int64_t ldv, int dpad) {
    const int i = blockIdx.x;
    const int64_t dst = dest[i];
    if (dst < 0) return;
    for (int c = threadIdx.x; c < dpad; c += blockDim.x)
        arena[dst * ldv + c] = c < d ? x[(int64_t)i * ldx + c] : 0.f;
}

void launch_ivfflat_append(const float* x, int64_t ldx, int n, int d, const int64_t* dest,
                           float* arena_vecs, int64_t ldv, int dpad, hipStream_t stream) {
    if (n == 0) return;
    hipLaunchKernelGGL(ivfflat_append_kernel, dim3((unsigned)n), dim3(64), 0, stream, x, ldx, n, d, dest,
                       arena_vecs, ldv, dpad);
    HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------
// IVF scalar quantizer: encoder (bit-identical to faiss::ScalarQuantizer::compute_codes) and range training
// ---------------------------------------------------------------------------------
// impl/scalar_quantizer/quantizers.h:76-90 / 118-132: x -> [0, 1] with the trained range, clamped
__device__ __forceinline__ float sq_unit(float x, float vmin, float vdiff) {
    float xi = 0.f;
    if (vdiff != 0.f) {
        xi = __fdiv_rn(x - vmin, vdiff);
        if (xi < 0.f) xi = 0.f;
        if (xi > 1.f) xi = 1.f;
    }
    return xi;
}
// one thread per (vector, chunk of 16 components): the chunk's 16 / 8 / 12 / 32 code bytes
__global__ void ivfsq_encode_kernel(int qtype, const float* __restrict__ x, int64_t ldx, int n, int d,
                                    const int64_t* __restrict__ labels, const int64_t* __restrict__ dest,
                                    const float* __restrict__ centroids, int64_t ldc, int by_residual,
                                    const float* __restrict__ vmin, const float* __restrict__ vdiff,
                                    uint8_t* __restrict__ arena, int ld, int nch) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = idx / nch;
    const int c = (int)(idx - i * nch);
    if (i >= n) return;
    const int64_t dst = dest[i];
    if (dst < 0) return;
    const int64_t l = labels[i];
    unsigned code[16]; // component codes (fp16: the half's bits)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int dim = 16 * c + e;
        unsigned cv = 0;
        if (dim < d) {
            float v = x[i * ldx + dim];
            if (by_residual) v = v - centroids[l * ldc + dim]; // Index::compute_residual
            switch (qtype) {
                case QT_8bit:
                case QT_8bit_uniform:
                    cv = (unsigned)(int)(255.f * sq_unit(v, vmin[dim], vdiff[dim])); // codecs.h:29-34
                    break;
                case QT_4bit:
                case QT_4bit_uniform:
                    cv = (unsigned)(int)((double)sq_unit(v, vmin[dim], vdiff[dim]) * 15.0); // codecs.h:48-53
                    break;
                case QT_6bit:
                    cv = (unsigned)(int)((double)sq_unit(v, vmin[dim], vdiff[dim]) * 63.0); // codecs.h:67-72
                    break;
                case QT_fp16:
                    cv = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)v); // encode_fp16: round to nearest even
                    break;
                default: // QT_8bit_direct (quantizers.h Quantizer8bitDirect): the value itself as a byte
                    cv = (unsigned)(uint8_t)(int)v;
                    break;
            }
        }
        code[e] = cv;
    }
    // chunk c of arena row dst in the block layout (kernels.h sq_code_offset)
    const int chb = (qtype == QT_fp16) ? 32 : (qtype == QT_4bit || qtype == QT_4bit_uniform) ? 8 : qtype == QT_6bit ? 12 : 16;
    uint8_t* chunk = arena + sq_code_offset(dst, c * chb, ld, chb);
    if (qtype == QT_fp16) {
        uint32_t* o = (uint32_t*)chunk;
#pragma unroll
        for (int w = 0; w < 8; ++w) o[w] = code[2 * w] | (code[2 * w + 1] << 16);
    } else if (qtype == QT_4bit || qtype == QT_4bit_uniform) {
        // component i in byte i / 2, low nibble first (codecs.h:48-53)
        uint32_t* o = (uint32_t*)chunk;
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            uint32_t v = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) v |= (code[8 * w + e] & 15u) << (4 * e);
            o[w] = v;
        }
    } else if (qtype == QT_6bit) {
        // a little-endian stream of 6-bit fields: 4 components per 3 bytes (codecs.h:67-92)
        uint32_t* o = (uint32_t*)chunk;
        unsigned long long lo = 0, hi = 0; // 96 bits
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int off = 6 * e;
            const unsigned long long v = code[e] & 63u;
            if (off < 64) {
                lo |= v << off;
                if (off > 58) hi |= v >> (64 - off);
            } else {
                hi |= v << (off - 64);
            }
        }
        o[0] = (uint32_t)lo;
        o[1] = (uint32_t)(lo >> 32);
        o[2] = (uint32_t)hi;
    } else {
        uint32_t* o = (uint32_t*)chunk;
#pragma unroll
        for (int w = 0; w < 4; ++w)
            o[w] = (code[4 * w] & 255u) | ((code[4 * w + 1] & 255u) << 8) | ((code[4 * w + 2] & 255u) << 16) |
                   ((code[4 * w + 3] & 255u) << 24);
    }
}
void launch_ivfsq_encode_append(int qtype, const float* x, int64_t ldx, int n, int d, const int64_t* labels,
                                const int64_t* dest, const float* centroids, int64_t ldc, bool by_residual,
                                const float* vmin, const float* vdiff, uint8_t* arena, int ld, hipStream_t stream) {
    if (n == 0) return;
    const int nch = (int)div_up(d, 16);
    hipLaunchKernelGGL(ivfsq_encode_kernel, dim3((unsigned)div_up((int64_t)n * nch, 256)), dim3(256), 0, stream, qtype, x,
                       ldx, n, d, labels, dest, centroids, ldc, by_residual ? 1 : 0, vmin, vdiff, arena, ld, nch);
    HIP_CHECK(hipGetLastError());
}

__global__ void ivfsq_pack_lists_kernel(const uint8_t* __restrict__ rows, const int64_t* __restrict__ src_start,
                                        const int64_t* __restrict__ dst_start, const uint32_t* __restrict__ len, int ld,
                                        int chb, uint8_t* __restrict__ arena) {
    const int l = blockIdx.x;
    const int wpr = ld >> 2; // 4-byte words per row
    const int64_t words = (int64_t)len[l] * wpr;
    const int64_t s0 = src_start[l], d0 = dst_start[l];
    for (int64_t w = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; w < words; w += (int64_t)gridDim.y * blockDim.x) {
        const int64_t i = w / wpr;
        const int byte = (int)(w - i * wpr) * 4;
        *(uint32_t*)(arena + sq_code_offset(d0 + i, byte, ld, chb)) = *(const uint32_t*)(rows + (s0 + i) * ld + byte);
    }
}
void launch_ivfsq_pack_lists(const uint8_t* rows, const int64_t* src_start, const int64_t* dst_start, const uint32_t* len,
                             int nlist, int ld, int chb, uint8_t* arena, hipStream_t stream) {
    if (nlist == 0) return;
    FA_THROW_IF_NOT(ld % 4 == 0 && chb % 4 == 0);
    const unsigned gy = nlist >= 1024 ? 1u : nlist >= 64 ? 8u : 64u;
    hipLaunchKernelGGL(ivfsq_pack_lists_kernel, dim3((unsigned)nlist, gy), dim3(256), 0, stream, rows, src_start, dst_start,
                       len, ld, chb, arena);
    HIP_CHECK(hipGetLastError());
}
__global__ void ivfsq_unpack_rows_kernel(const uint8_t* __restrict__ arena, int64_t row0, int64_t n, int ld, int chb,
                                         uint8_t* __restrict__ rows) {
    const int wpr = ld >> 2;
    const int64_t words = n * wpr;
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < words; w += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = w / wpr;
        const int byte = (int)(w - i * wpr) * 4;
        *(uint32_t*)(rows + i * ld + byte) = *(const uint32_t*)(arena + sq_code_offset(row0 + i, byte, ld, chb));
    }
}
void launch_ivfsq_unpack_rows(const uint8_t* arena, int64_t row0, int64_t n, int ld, int chb, uint8_t* rows,
                              hipStream_t stream) {
    if (n == 0) return;
    const unsigned grid = (unsigned)std::min<int64_t>(div_up(n * (ld >> 2), 256), 4096);
    hipLaunchKernelGGL(ivfsq_unpack_rows_kernel, dim3(grid), dim3(256), 0, stream, arena, row0, n, ld, chb, rows);
    HIP_CHECK(hipGetLastError());
}

constexpr int SQ_MM_BLOCKS = 256;
int ivfsq_minmax_blocks(int64_t n) {
    return (int)std::min<int64_t>(SQ_MM_BLOCKS, std::max<int64_t>(n, 1));
}
__global__ void ivfsq_minmax_kernel(const float* __restrict__ x, int64_t ldx, int64_t n, int d, float* __restrict__ out) {
    const int b = blockIdx.x, nb = gridDim.x;
    for (int j = threadIdx.x; j < d; j += blockDim.x) {
        float lo = INFINITY, hi = -INFINITY;
        for (int64_t r = b; r < n; r += nb) {
            const float v = x[r * ldx + j];
            if (v < lo) lo = v; // (train_NonUniform's comparisons, impl/scalar_quantizer/training.cpp:345-356)
            if (v > hi) hi = v;
        }
        out[((int64_t)b * 2 + 0) * d + j] = lo;
        out[((int64_t)b * 2 + 1) * d + j] = hi;
    }
}
void launch_ivfsq_minmax(const float* x, int64_t ldx, int64_t n, int d, float* out, hipStream_t stream) {
    hipLaunchKernelGGL(ivfsq_minmax_kernel, dim3((unsigned)ivfsq_minmax_blocks(n)), dim3(256), 0, stream, x, ldx, n, d, out);
    HIP_CHECK(hipGetLastError());
}

__global__ void ivfflat_rows_by_id_kernel(const float* __restrict__ arena, int64_t ldv, const int64_t* __restrict__ ids,
                                          const int64_t* __restrict__ list_start, const uint32_t* __restrict__ list_len,
                                          int d, int64_t i0, int64_t ni, float* __restrict__ out) {
    const int l = blockIdx.x;
    const int64_t start = list_start[l];
    const unsigned len = list_len[l];
    // one wavefront per candidate row: the 64 lanes copy the row once its id is known to be wanted
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    for (unsigned i = blockIdx.y * nw + wave; i < len; i += gridDim.y * nw) {
        const int64_t id = ids[start + i];
        if (id < i0 || id >= i0 + ni) continue;
        for (int c = lane; c < d; c += 64) out[(id - i0) * d + c] = arena[(start + i) * ldv + c];
    }
}
void launch_ivfflat_rows_by_id(const float* arena_vecs, int64_t ldv, const int64_t* arena_ids, const int64_t* list_start,
                               const uint32_t* list_len, int nlist, int d, int64_t i0, int64_t ni, float* out,
                               hipStream_t stream) {
    if (nlist == 0 || ni == 0) return;
    const unsigned gy = nlist >= 1024 ? 1u : nlist >= 64 ? 8u : 64u;
    hipLaunchKernelGGL(ivfflat_rows_by_id_kernel, dim3((unsigned)nlist, gy), dim3(256), 0, stream, arena_vecs, ldv,
                       arena_ids, list_start, list_len, d, i0, ni, out);
    HIP_CHECK(hipGetLastError());
}

// one thread per (vector, sub-quantizer)
__global__ void ivfpq_encode_append_kernel(const float* __restrict__ x, int64_t ldx, int n, int d,
                                           const int64_t* __restrict__ labels,
                                           const int64_t* __restrict__ dest,
                                           const float* __restrict__ centroids, int64_t ldc, int M,
                                           int dsub, const float* __restrict__ pq,
                                           uint8_t* __restrict__ codes) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)n * M) return;
    const int i = (int)(t / M), m = (int)(t % M);
    const int64_t dst = dest[i];
    if (dst < 0) return;
    const int64_t list = labels[i];
    const float* xr = x + (int64_t)i * ldx + m * dsub;
    const float* cr = centroids + list * ldc + m * dsub;
    const float* pm = pq + (size_t)m * 256 * dsub;
    float best = INFINITY;
    int bestc = 0;
    for (int c = 0; c < 256; ++c) {
        const float* cen = pm + c * dsub;
        float acc = 0.f;
        for (int jd = 0; jd < dsub; ++jd) {
            float r = xr[jd] - cr[jd];
            float tt = r - cen[jd];
            acc = __fmaf_rn(tt, tt, acc);
        }
        if (acc < best) {
            best = acc;
            bestc = c;
        }
    }
    codes[pq_code_offset(M, dst, m)] = (uint8_t)bestc;
}

// Same arithmetic, one workgroup per (256 vectors, sub-quantizer): the 256 x dsub codebook of the sub-quantizer
// sits in LDS (every lane reads the same centroid: a broadcast) and the residual slice in registers, instead of
// 256 dependent global loads per thread (82 ms per million vectors at M = 64).
template <int DSUB_MAX>
__global__ void __launch_bounds__(256) ivfpq_encode_append_lds_kernel(const float* __restrict__ x, int64_t ldx, int n,
                                                                      const int64_t* __restrict__ labels,
                                                                      const int64_t* __restrict__ dest,
                                                                      const float* __restrict__ centroids, int64_t ldc,
                                                                      int M, int dsub, const float* __restrict__ pq,
                                                                      uint8_t* __restrict__ codes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* cb = (float*)smem; // [256][dsub]
    const int m = blockIdx.y;
    const float* pm = pq + (size_t)m * 256 * dsub;
    for (int e = threadIdx.x; e < 256 * dsub; e += 256) cb[e] = pm[e];
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t dst = dest[i];
    if (dst < 0) return;
    const int64_t list = labels[i];
    const float* xr = x + (int64_t)i * ldx + m * dsub;
    const float* cr = centroids + list * ldc + m * dsub;
    float r[DSUB_MAX];
#pragma unroll
    for (int jd = 0; jd < DSUB_MAX; ++jd) r[jd] = jd < dsub ? xr[jd] - cr[jd] : 0.f;
    float best = INFINITY;
    int bestc = 0;
    for (int c = 0; c < 256; ++c) {
        const float* cen = cb + c * dsub;
        float acc = 0.f;
#pragma unroll
        for (int jd = 0; jd < DSUB_MAX; ++jd) {
            if (jd < dsub) {
                const float tt = r[jd] - cen[jd];
                acc = __fmaf_rn(tt, tt, acc);
            }
        }
        if (acc < best) {
            best = acc;
            bestc = c;
        }
    }
    codes[pq_code_offset(M, dst, m)] = (uint8_t)bestc;
}

void launch_ivfpq_encode_append(const float* x, int64_t ldx, int n, int d, const int64_t* labels,
                                const int64_t* dest, const float* centroids, int64_t ldc, int M,
                                int dsub, const float* pq_centroids, uint8_t* arena_codes,
                                hipStream_t stream) {
    if (n == 0) return;
    if (dsub <= 16) {
        const dim3 grid((unsigned)div_up(n, 256), (unsigned)M);
        const size_t lds = (size_t)256 * dsub * 4;
        if (dsub <= 4)
            hipLaunchKernelGGL((ivfpq_encode_append_lds_kernel<4>), grid, dim3(256), lds, stream, x, ldx, n, labels, dest,
                               centroids, ldc, M, dsub, pq_centroids, arena_codes);
        else
            hipLaunchKernelGGL((ivfpq_encode_append_lds_kernel<16>), grid, dim3(256), lds, stream, x, ldx, n, labels, dest,
                               centroids, ldc, M, dsub, pq_centroids, arena_codes);
    } else {
        int64_t total = (int64_t)n * M;
        hipLaunchKernelGGL(ivfpq_encode_append_kernel, dim3((unsigned)div_up(total, 256)), dim3(256), 0,
                           stream, x, ldx, n, d, labels, dest, centroids, ldc, M, dsub, pq_centroids,
                           arena_codes);
    }
    HIP_CHECK(hipGetLastError());
}

__device__ __forceinline__ float ivfpq_t2_of_row(const uint8_t* __restrict__ codes, int64_t row, const float* cen, int M,
                                                 int dsub, const float* __restrict__ pq) {
    float acc = 0.f;
    for (int m = 0; m < M; ++m) {
        const float* pc = pq + ((size_t)m * 256 + codes[pq_code_offset(M, row, m)]) * dsub;
        for (int jd = 0; jd < dsub; ++jd) {
            const float rv = pc[jd];
            acc = __fmaf_rn(rv, __fmaf_rn(2.f, cen[m * dsub + jd], rv), acc);
        }
    }
    return acc;
}
__global__ void ivfpq_t2_lists_kernel(const uint8_t* __restrict__ codes, const int64_t* __restrict__ list_start,
                                      const uint32_t* __restrict__ list_len, const float* __restrict__ centroids,
                                      int64_t ldc, int M, int dsub, const float* __restrict__ pq,
                                      float* __restrict__ t2) {
    const int l = blockIdx.x;
    const int64_t start = list_start[l];
    const unsigned len = list_len[l];
    const float* cen = centroids + (int64_t)l * ldc;
    for (unsigned i = blockIdx.y * blockDim.x + threadIdx.x; i < len; i += gridDim.y * blockDim.x)
        t2[start + i] = ivfpq_t2_of_row(codes, start + i, cen, M, dsub, pq);
}
void launch_ivfpq_t2_lists(const uint8_t* arena_codes, const int64_t* list_start, const uint32_t* list_len, int nlist,
                           const float* centroids, int64_t ldc, int M, int dsub, const float* pq_centroids, float* t2,
                           hipStream_t stream) {
    if (nlist == 0) return;
    const unsigned gy = nlist >= 1024 ? 1u : nlist >= 64 ? 8u : 64u;
    hipLaunchKernelGGL(ivfpq_t2_lists_kernel, dim3((unsigned)nlist, gy), dim3(256), 0, stream, arena_codes, list_start,
                       list_len, centroids, ldc, M, dsub, pq_centroids, t2);
    HIP_CHECK(hipGetLastError());
}
__global__ void ivfpq_t2_rows_kernel(const uint8_t* __restrict__ codes, const int64_t* __restrict__ labels,
                                     const int64_t* __restrict__ dest, int n, const float* __restrict__ centroids,
                                     int64_t ldc, int M, int dsub, const float* __restrict__ pq, float* __restrict__ t2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t row = dest[i];
    if (row < 0) return;
    t2[row] = ivfpq_t2_of_row(codes, row, centroids + labels[i] * ldc, M, dsub, pq);
}
void launch_ivfpq_t2_rows(const uint8_t* arena_codes, const int64_t* labels, const int64_t* dest, int n,
                          const float* centroids, int64_t ldc, int M, int dsub, const float* pq_centroids, float* t2,
                          hipStream_t stream) {
    if (n == 0) return;
    hipLaunchKernelGGL(ivfpq_t2_rows_kernel, dim3((unsigned)div_up(n, 256)), dim3(256), 0, stream, arena_codes, labels,
                       dest, n, centroids, ldc, M, dsub, pq_centroids, t2);
    HIP_CHECK(hipGetLastError());
}

// plain [row][M] <-> rotated block layout
__global__ void ivfpq_pack_lists_kernel(const uint8_t* __restrict__ plain, const int64_t* __restrict__ src_start,
                                        const int64_t* __restrict__ list_start, const uint32_t* __restrict__ list_len,
                                        int M, uint8_t* __restrict__ codes) {
    const int l = blockIdx.x;
    const int64_t s0 = src_start[l], d0 = list_start[l];
    const int64_t total = (int64_t)list_len[l] * M;
    for (int64_t t = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.y * blockDim.x) {
        const int64_t i = t / M;
        const int m = (int)(t - i * M);
        codes[pq_code_offset(M, d0 + i, m)] = plain[(s0 + i) * M + m];
    }
}
void launch_ivfpq_pack_lists(const uint8_t* plain, const int64_t* src_start, const int64_t* list_start,
                             const uint32_t* list_len, int nlist, int M, uint8_t* arena_codes, hipStream_t stream) {
    if (nlist == 0) return;
    const unsigned gy = nlist >= 1024 ? 1u : nlist >= 64 ? 8u : 64u;
    hipLaunchKernelGGL(ivfpq_pack_lists_kernel, dim3((unsigned)nlist, gy), dim3(256), 0, stream, plain, src_start,
                       list_start, list_len, M, arena_codes);
    HIP_CHECK(hipGetLastError());
}
__global__ void ivfpq_unpack_list_kernel(const uint8_t* __restrict__ codes, int64_t first_row, uint32_t len, int M,
                                         uint8_t* __restrict__ plain) {
    const int64_t total = (int64_t)len * M;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = t / M;
        const int m = (int)(t - i * M);
        plain[t] = codes[pq_code_offset(M, first_row + i, m)];
    }
}
void launch_ivfpq_unpack_list(const uint8_t* arena_codes, int64_t first_row, uint32_t len, int M, uint8_t* plain,
                              hipStream_t stream) {
    if (len == 0) return;
    const unsigned grid = (unsigned)std::min<int64_t>(div_up((int64_t)len * M, 256), 4096);
    hipLaunchKernelGGL(ivfpq_unpack_list_kernel, dim3(grid), dim3(256), 0, stream, arena_codes, first_row, len, M, plain);
    HIP_CHECK(hipGetLastError());
}
__global__ void pq_transpose_kernel(const float* __restrict__ pq, int M, int dsub, float* __restrict__ pq_t) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x; // over [256][M][dsub]
    if (t >= 256 * M * dsub) return;
    const int jd = t % dsub, m = (t / dsub) % M, c = t / (dsub * M);
    pq_t[t] = pq[((size_t)m * 256 + c) * dsub + jd];
}
void launch_pq_transpose(const float* pq, int M, int dsub, float* pq_t, hipStream_t stream) {
    hipLaunchKernelGGL(pq_transpose_kernel, dim3((unsigned)div_up((size_t)256 * M * dsub, 256)), dim3(256), 0, stream, pq,
                       M, dsub, pq_t);
    HIP_CHECK(hipGetLastError());
}

__global__ void scatter_i64_kernel(const int64_t* __restrict__ src, const int64_t* __restrict__ dest,
                                   int64_t n, int64_t* __restrict__ dst) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t t = dest[i];
    if (t >= 0) dst[t] = src[i];
}

void launch_scatter_i64(const int64_t* src, const int64_t* dest, int64_t n, int64_t* dst,
                        hipStream_t stream) {
    if (n == 0) return;
    hipLaunchKernelGGL(scatter_i64_kernel, dim3((unsigned)div_up(n, 256)), dim3(256), 0, stream, src, dest,
                       n, dst);
    HIP_CHECK(hipGetLastError());
}

// INDICES_IVF (faiss/gpu/GpuIndicesOptions.h:20-23, impl/IVFUtilsSelect2.cu:148): the label of an entry is
// (inverted list << 32 | offset inside the list) instead of a user id
__global__ void pair_ids_kernel(const int64_t* __restrict__ labels, const int64_t* __restrict__ dest, int64_t n,
                                const int64_t* __restrict__ list_start, int64_t* __restrict__ dst) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t t = dest[i], l = labels[i];
    if (t >= 0 && l >= 0) dst[t] = (l << 32) | (t - list_start[l]);
}
void launch_ivf_pair_ids(const int64_t* labels, const int64_t* dest, int64_t n, const int64_t* list_start, int64_t* dst,
                         hipStream_t stream) {
    if (n == 0) return;
    hipLaunchKernelGGL(pair_ids_kernel, dim3((unsigned)div_up(n, 256)), dim3(256), 0, stream, labels, dest, n, list_start, dst);
    HIP_CHECK(hipGetLastError());
}

__global__ void residual_kernel(const float* __restrict__ x, int64_t ldx, int64_t n, int d,
                                const int64_t* __restrict__ labels, const float* __restrict__ centroids,
                                int64_t ldc, float* __restrict__ out, int64_t ldo) {
    const int64_t i = blockIdx.x;
    const int64_t l = labels[i];
    for (int c = threadIdx.x; c < d; c += blockDim.x) {
        float v = x[i * ldx + c];
        out[i * ldo + c] = l >= 0 ? v - centroids[l * ldc + c] : v;
    }
}

void launch_residual(const float* x, int64_t ldx, int64_t n, int d, const int64_t* labels,
                     const float* centroids, int64_t ldc, float* out, int64_t ldo, hipStream_t stream) {
    if (n == 0) return;
    hipLaunchKernelGGL(residual_kernel, dim3((unsigned)n), dim3(64), 0, stream, x, ldx, n, d, labels,
                       centroids, ldc, out, ldo);
    HIP_CHECK(hipGetLastError());
}

} // namespace faiss_amd
