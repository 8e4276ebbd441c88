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
//   IVFPQ L2: r = q - centroid; lut[m][c] = chain_j fmaf(r_mj - pq[m][c][j], same, acc);
//             dis = ((0 + lut[0][c0]) + lut[1][c1]) + ...  (m ascending)
//   IVFPQ IP: lut[m][c] = chain_j fmaf(q_mj, pq[m][c][j], acc); dis = coarse_ip + sum_m lut
//   IVFFlat : dis = chain_k fmaf(q[k]-y[k], q[k]-y[k], acc) (L2) / fmaf(q[k], y[k], acc) (IP)
//   selection: k smallest keys (ordkey(dis) << 32 | position in probe order) = "first scanned wins"
//             among equal distances (faiss/IndexIVF.cpp:642-655 + strict heap admission); the k winners
//             are then ordered by (distance, id) (faiss/impl/ResultHandler.h:439-453).
#include "kernels.h"
#include "wg_select.h"

namespace faiss_amd {

constexpr int FB = 1024;  // threads per workgroup (16 wavefronts, one workgroup per CU)
constexpr int FPART = FB / 256; // codebook split over the threads: thread t owns centroid t & 255 of part t >> 8
constexpr int FMAXR = 8;  // reservoir capacity <= FMAXR * FB keys

struct FusedLds {
    char* lut;        // [M][256] fp32 | at the end: winners
    float* rs;        // [dpad] residual (L2) or query (IP)
    u64* res;         // [cap]
    uint32_t* pre;    // [nprobe + 1]
    int* lst;         // [nprobe]
    int64_t* lstart;  // [nprobe] first arena row of each probed list
    unsigned* hist;   // [256]
    WgSelCtl* ctl;
};

static size_t fused_region0_bytes(int kind, int M, int kp, int nlut) {
    size_t lut = kind == 1 ? (size_t)M * 1024 * nlut : 0;
    return round_up(std::max<size_t>(std::max<size_t>(lut, (size_t)kp * 12), 16), 16);
}
size_t ivf_fused_lds_bytes(int kind, int M, int dpad, int kp, int cap, int nprobe, int nlut) {
    return fused_region0_bytes(kind, M, kp, nlut) + 2 * round_up((size_t)dpad * 4, 16) + (size_t)cap * 8 +
           round_up((size_t)(nprobe + 1) * 4, 16) + round_up((size_t)nprobe * 4, 16) + (size_t)nprobe * 8 + 1024 + 64;
}

__device__ __forceinline__ FusedLds fused_carve(char* smem, const IvfFusedParams& p) {
    FusedLds L;
    size_t o = 0;
    L.lut = smem;
    {
        size_t lut = p.kind == 1 ? (size_t)p.M * 1024 * p.nlut : 0;
        size_t r0 = lut > (size_t)p.kp * 12 ? lut : (size_t)p.kp * 12;
        if (r0 < 16) r0 = 16;
        o = (r0 + 15) & ~(size_t)15;
    }
    L.rs = (float*)(smem + o); // two buffers of rs_stride floats
    o += 2 * (((size_t)p.dpad * 4 + 15) & ~(size_t)15);
    L.res = (u64*)(smem + o);
    o += (size_t)p.cap * 8;
    L.pre = (uint32_t*)(smem + o);
    o += ((size_t)(p.nprobe + 1) * 4 + 15) & ~(size_t)15;
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
__device__ __forceinline__ void fused_load_probes(const IvfFusedParams& p, int q, const FusedLds& L) {
    const int tid = threadIdx.x;
    for (int t = tid; t < p.nprobe; t += FB) {
        const int64_t l = p.coarse_ids[(int64_t)q * p.nprobe + t];
        L.lst[t] = (int)l;
        L.pre[t + 1] = l >= 0 ? p.list_len[l] : 0u;
        L.lstart[t] = l >= 0 ? p.list_start[l] : 0;
    }
    if (tid == 0) {
        L.pre[0] = 0;
        L.ctl->cnt = 0;
    }
    __syncthreads();
    if (tid < 64) {
        unsigned carry = 0;
        for (int base = 0; base < p.nprobe; base += 64) {
            const int t = base + tid;
            unsigned v = t < p.nprobe ? L.pre[t + 1] : 0u;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned o = __shfl_up(v, off, 64);
                if (tid >= off) v += o;
            }
            v += carry;
            if (t < p.nprobe) L.pre[t + 1] = v;
            carry = __shfl(v, 63, 64);
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

// final k-selection, position -> user id, ordering, write-out (or partial keys when G > 1)
__device__ __forceinline__ void fused_finish(const IvfFusedParams& p, int q, int g, const FusedLds& L) {
    const int tid = threadIdx.x;
    __syncthreads();
    int n = (int)L.ctl->cnt;
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
    wg_bitonic_sort<FB>(w_key, w_id, p.kp);
    const float pad = neutral_distance(p.metric);
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
// IVFPQ.  DSUB > 0: the PQ codebook lives in registers (thread t owns centroid t & 255 of the
// sub-quantizers of its part (t >> 8) of M; d <= 128) and every table is built without
// touching global memory; DSUB == 0: generic fallback that re-reads the codebook through L2.
// ---------------------------------------------------------------------------------
// M64: the sub-quantizer count is the compile-time constant 64 (the headline PQ64 shape): every lane's
// quarter of a code is exactly one 16-byte load and the byte-wise fallback paths (and their address
// registers) disappear from the kernel.
template <int METRIC, int DSUB, bool M64, bool TIMING = false>
__global__ void __launch_bounds__(FB) ivfpq_fused_kernel(IvfFusedParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const FusedLds L = fused_carve(smem, p);
    const int tid = threadIdx.x;
    const int q = blockIdx.x / p.G, g = blockIdx.x - q * p.G;
    const int p0 = g * p.npc, p1 = min(p.nprobe, p0 + p.npc);
    float* lut = (float*)L.lut;
    const int M = M64 ? 64 : p.M, d = p.d, dsub = p.dsub;

    unsigned long long t_prev = TIMING ? clock64() : 0ull;
    unsigned long long t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define FUSED_TICK(slot)                               \
    do {                                               \
        if (TIMING) {                                  \
            const unsigned long long t_now = clock64(); \
            t_acc[slot] += t_now - t_prev;             \
            t_prev = t_now;                            \
        }                                              \
    } while (0)
    fused_load_probes(p, q, L);

    // codebook slice of this thread: centroid c of the sub-quantizers of its quarter of M
    constexpr int NREG = DSUB > 0 ? 128 / FPART : 1;
    float pqr[NREG];
    const int c = tid & 255;
    const int quarter = __builtin_amdgcn_readfirstlane(tid >> 8);
    const int mq = M / FPART; // sub-quantizers per part
    if (DSUB > 0) {
#pragma unroll
        for (int mm = 0; mm < NREG / DSUB; ++mm) {
#pragma unroll
            for (int jd = 0; jd < DSUB; ++jd) {
                float v = 0.f;
                if (mm < mq) v = p.pq_centroids[((size_t)(quarter * mq + mm) * 256 + c) * DSUB + jd];
                pqr[mm * DSUB + jd] = v;
            }
        }
    }

    const int rs_stride = (p.dpad + 3) & ~3;
    auto build_lut = [&](float* lut, const float* rsb) {
        if (DSUB > 0) {
#pragma unroll
            for (int mm = 0; mm < NREG / DSUB; ++mm) {
                if (mm < mq) {
                    const int m = quarter * mq + mm;
                    float acc = 0.f;
#pragma unroll
                    for (int jd = 0; jd < DSUB; ++jd) {
                        const float r = rsb[m * DSUB + jd];
                        if (METRIC == METRIC_L2) {
                            const float t = r - pqr[mm * DSUB + jd];
                            acc = __fmaf_rn(t, t, acc);
                        } else {
                            acc = __fmaf_rn(r, pqr[mm * DSUB + jd], acc);
                        }
                    }
                    lut[m * 256 + c] = acc;
                }
                // keep the LDS reads of the residual from being hoisted 32 deep (VGPR pressure)
                if ((mm & 7) == 7) __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            for (int e = tid; e < M * 256; e += FB) {
                const int m = e >> 8;
                const float* cen = p.pq_centroids + (size_t)e * dsub;
                const float* r = rsb + m * dsub;
                float acc = 0.f;
                for (int jd = 0; jd < dsub; ++jd) {
                    if (METRIC == METRIC_L2) {
                        const float t = r[jd] - cen[jd];
                        acc = __fmaf_rn(t, t, acc);
                    } else {
                        acc = __fmaf_rn(r[jd], cen[jd], acc);
                    }
                }
                lut[e] = acc;
            }
        }
    };

    // ---- scan geometry: four adjacent lanes share one code, each owns M/4 consecutive
    // sub-quantizers (a 16-byte load per lane at M = 64, the wave reads 1 KB of contiguous codes per
    // instruction); the four partial sums meet in a 2-step lane butterfly.  All 16 wavefronts
    // gather from the table instead of the 4 that one-code-per-thread leaves busy at ~250-entry lists.
    // ---- software pipeline across probes: the residual of probe pr+1 (one value per thread) and
    // the first code chunk of probe pr are fetched from L2/HBM while the table of probe pr is
    // being built, so that no global latency sits between two barriers.
    constexpr int CPC = FB / 4;                 // codes per chunk
    const int ml = M >> 2;                      // sub-quantizers per lane
    const int jq = tid & 3;                     // this lane's quarter of the code
    const bool wide = M64 || ((ml & 3) == 0 && ml <= 16); // quarter fetched as up to 4 dwords kept in registers
    unsigned cw[4];
    auto fetch_codes = [&](int64_t start, unsigned i, unsigned len) {
        if (wide && i < len) {
            const uint8_t* code = p.arena_codes + (start + i) * M + jq * ml;
            if (M64 || ml == 16) {
                const uint4 v = *(const uint4*)code;
                cw[0] = v.x; cw[1] = v.y; cw[2] = v.z; cw[3] = v.w;
            } else {
#pragma unroll
                for (int w = 0; w < 4; ++w)
                    if (w * 4 < ml) cw[w] = *(const unsigned*)(code + w * 4);
            }
        }
    };
    auto residual_of = [&](int pr) -> float {
        // threads tid < d only; list < 0 (fewer than nprobe lists exist) reads list 0, result unused
        const int l = max(L.lst[pr], 0);
        return p.xq[(int64_t)q * p.ldq + tid] - p.centroids[(int64_t)l * p.ldc + tid];
    };

    if (METRIC != METRIC_L2) {
        // inner product: the table depends on the query only
        for (int cc = tid; cc < d; cc += FB) L.rs[cc] = p.xq[(int64_t)q * p.ldq + cc];
        __syncthreads();
        build_lut(lut, L.rs);
        __syncthreads();
    }
    const bool res_in_reg = METRIC == METRIC_L2 && d <= FB; // one residual coordinate per thread
    float rres = 0.f;
    if (res_in_reg && tid < d && p0 < p1) rres = residual_of(p0);

    u64 tau = ~0ull;
    int bound = 0;
    FUSED_TICK(0); // prologue: probe table, codebook registers
    // scan of one list against table `lt` (first chunk's codes already in cw)
    auto scan_list = [&](const float* lt, int64_t start, unsigned pos0, unsigned len, float dis0) {
        for (unsigned base = 0; base < len; base += CPC) {
            FUSED_MAKE_ROOM(min((unsigned)CPC, len - base));
            FUSED_TICK(4); // make room (reservoir compaction when needed)
            const unsigned i = base + (tid >> 2);
            float part = 0.f;
            if (i < len) {
                const float* lq = lt + (size_t)jq * ml * 256;
                if (wide) {
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        if (w * 4 < ml) {
#pragma unroll
                            for (int b = 0; b < 4; ++b)
                                part = part + lq[(w * 4 + b) * 256 + ((cw[w] >> (8 * b)) & 255u)];
                        }
                    }
                } else {
                    const uint8_t* code = p.arena_codes + (start + i) * M + jq * ml;
                    for (int m = 0; m < ml; ++m) part = part + lq[m * 256 + code[m]];
                }
            }
            // (p0 + p1) + (p2 + p3): xor-butterfly over the quad, identical bits in its four lanes
            part = part + __shfl_xor(part, 1, 64);
            part = part + __shfl_xor(part, 2, 64);
            const float acc = dis0 + part;
            const u64 key = ((u64)ordkey<METRIC>(acc) << 32) | (u64)(pos0 + i);
            const bool pass = i < len && jq == 0 && key < tau;
            // next chunk of this list (long lists): in flight behind the append and the barrier
            if (base + CPC < len) fetch_codes(start, i + CPC, len);
            FUSED_TICK(5); // gathers + butterfly (includes the wait for the codes)
            wg_append(L.res, L.ctl, pass, key);
            if (base + CPC < len) __syncthreads(); // (the caller ends the list with its own barrier)
            FUSED_TICK(6); // append (+ barrier)
        }
    };

    if (METRIC == METRIC_L2 && p.nlut == 2 && res_in_reg) {
        // ---- double-buffered tables: while the gathers of probe pr run against table b, the
        // table of probe pr+1 is written into table b^1 by the same threads (LDS writes and reads
        // of different buffers interleave in the LDS pipeline); ONE barrier per probe.
        float* lutb[2] = {lut, lut + (size_t)M * 256};
        float* rsb[2] = {L.rs, L.rs + rs_stride};
        if (p0 < p1) {
            if (tid < d) rsb[0][tid] = rres;
            if (tid < d && p0 + 1 < p1) rres = residual_of(p0 + 1);
            __syncthreads();
            build_lut(lutb[0], rsb[0]);
            if (tid < d) rsb[1][tid] = rres; // residual of probe p0+1 (garbage when there is none: unused)
            {
                const int l0 = L.lst[p0];
                fetch_codes(L.lstart[p0], tid >> 2, l0 < 0 ? 0u : L.pre[p0 + 1] - L.pre[p0]);
            }
            __syncthreads();
        }
        FUSED_TICK(1);
        for (int pr = p0; pr < p1; ++pr) {
            const int b = (pr - p0) & 1;
            const int list = L.lst[pr];
            const unsigned pos0 = L.pre[pr];
            const unsigned len = (list < 0) ? 0u : L.pre[pr + 1] - pos0;
            const int64_t start = L.lstart[pr];
            // in flight during this iteration: residual of probe pr+2, first codes of probe pr+1
            if (tid < d && pr + 2 < p1) rres = residual_of(pr + 2);
            unsigned cwn[4] = {0, 0, 0, 0};
            if (pr + 1 < p1) {
                const int l1 = L.lst[pr + 1];
                const unsigned len1 = l1 < 0 ? 0u : L.pre[pr + 2] - L.pre[pr + 1];
                const unsigned i1 = tid >> 2;
                if (wide && i1 < len1) {
                    const uint8_t* code = p.arena_codes + (L.lstart[pr + 1] + i1) * M + jq * ml;
                    if (M64 || ml == 16) {
                        const uint4 v = *(const uint4*)code;
                        cwn[0] = v.x; cwn[1] = v.y; cwn[2] = v.z; cwn[3] = v.w;
                    } else {
#pragma unroll
                        for (int w = 0; w < 4; ++w)
                            if (w * 4 < ml) cwn[w] = *(const unsigned*)(code + w * 4);
                    }
                }
                build_lut(lutb[b ^ 1], rsb[b ^ 1]);
            }
            FUSED_TICK(2);
            scan_list(lutb[b], start, pos0, len, 0.f);
            // rs[b] was consumed by the build of probe pr (previous iteration): refill for pr+2
            if (tid < d && pr + 2 < p1) rsb[b][tid] = rres;
#pragma unroll
            for (int w = 0; w < 4; ++w) cw[w] = cwn[w];
            __syncthreads();
            FUSED_TICK(3);
        }
    } else {
        for (int pr = p0; pr < p1; ++pr) {
            const int list = L.lst[pr];
            const unsigned pos0 = L.pre[pr];
            const unsigned len = (list < 0) ? 0u : L.pre[pr + 1] - pos0;
            const int64_t start = L.lstart[pr];
            float dis0 = 0.f;
            if (METRIC == METRIC_L2) {
                if (res_in_reg) {
                    if (tid < d) L.rs[tid] = rres;
                } else {
                    const int l = max(list, 0);
                    for (int cc = tid; cc < d; cc += FB)
                        L.rs[cc] = p.xq[(int64_t)q * p.ldq + cc] - p.centroids[(int64_t)l * p.ldc + cc];
                }
                __syncthreads();
                FUSED_TICK(1); // residual -> LDS + barrier
                // in flight during the table build: next probe's residual, this probe's first codes
                if (res_in_reg && tid < d && pr + 1 < p1) rres = residual_of(pr + 1);
                fetch_codes(start, tid >> 2, len);
                build_lut(lut, L.rs);
                FUSED_TICK(2); // table build (this wave)
                __syncthreads();
                FUSED_TICK(3); // barrier after the build
            } else {
                dis0 = p.coarse_dis[(int64_t)q * p.nprobe + pr];
                fetch_codes(start, tid >> 2, len);
            }
            scan_list(lut, start, pos0, len, dis0);
            __syncthreads();
        }
    }
    fused_finish(p, q, g, L);
    FUSED_TICK(7); // final selection, id translation, sort, write-out
    if (TIMING && p.dbg && tid == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) atomicAdd(p.dbg + i, t_acc[i]);
    }
#undef FUSED_TICK
}

// ---------------------------------------------------------------------------------
// IVFFlat: same reservoir machinery, distances straight from the fp32 rows of the list
// ---------------------------------------------------------------------------------
template <int METRIC>
__global__ void __launch_bounds__(FB) ivfflat_fused_kernel(IvfFusedParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const FusedLds L = fused_carve(smem, p);
    const int tid = threadIdx.x;
    const int q = blockIdx.x / p.G, g = blockIdx.x - q * p.G;
    const int p0 = g * p.npc, p1 = min(p.nprobe, p0 + p.npc);

    fused_load_probes(p, q, L);
    for (int cc = tid; cc < p.dpad; cc += FB) L.rs[cc] = p.xq[(int64_t)q * p.ldq + cc];
    __syncthreads();

    u64 tau = ~0ull;
    int bound = 0;
    for (int pr = p0; pr < p1; ++pr) {
        const int list = L.lst[pr];
        const unsigned pos0 = L.pre[pr];
        const unsigned len = L.pre[pr + 1] - pos0;
        if (list < 0 || len == 0) continue;
        const int64_t start = L.lstart[pr];
        for (unsigned base = 0; base < len; base += FB) {
            FUSED_MAKE_ROOM(min((unsigned)FB, len - base));
            const unsigned i = base + tid;
            bool pass = false;
            u64 key = 0;
            if (i < len) {
                const float* y = p.arena_vecs + (start + i) * p.ldv;
                float acc = 0.f;
                for (int k = 0; k < p.dpad; k += 4) {
                    const float4 yv = *(const float4*)(y + k);
                    const float4 qv = *(const float4*)(L.rs + k);
                    if (METRIC == METRIC_L2) {
                        float t;
                        t = qv.x - yv.x; acc = __fmaf_rn(t, t, acc);
                        t = qv.y - yv.y; acc = __fmaf_rn(t, t, acc);
                        t = qv.z - yv.z; acc = __fmaf_rn(t, t, acc);
                        t = qv.w - yv.w; acc = __fmaf_rn(t, t, acc);
                    } else {
                        acc = __fmaf_rn(qv.x, yv.x, acc);
                        acc = __fmaf_rn(qv.y, yv.y, acc);
                        acc = __fmaf_rn(qv.z, yv.z, acc);
                        acc = __fmaf_rn(qv.w, yv.w, acc);
                    }
                }
                key = ((u64)ordkey<METRIC>(acc) << 32) | (u64)(pos0 + i);
                pass = key < tau;
            }
            wg_append(L.res, L.ctl, pass, key);
            __syncthreads();
        }
    }
    fused_finish(p, q, g, L);
}

// ---------------------------------------------------------------------------------
bool ivf_fused_supported(int kind, int M, int dpad, int k, int nprobe, int* cap_out, int* kp_out, int* nlut_out) {
    int kp = 1;
    while (kp < k) kp <<= 1;
    int cap = 1024;
    while (cap < k + FB) cap <<= 1;
    if (cap > FMAXR * FB) return false;
    if (cap_out) *cap_out = cap;
    if (kp_out) *kp_out = kp;
    // two lookup tables (build of probe p+1 overlapped with the scan of probe p) when they fit
    int nlut = (kind == 1 && ivf_fused_lds_bytes(kind, M, dpad, kp, cap, nprobe, 2) <= 160 * 1024) ? 2 : 1;
    if (nlut_out) *nlut_out = nlut;
    return ivf_fused_lds_bytes(kind, M, dpad, kp, cap, nprobe, nlut) <= 160 * 1024;
}

template <typename K>
static void launch_one(K kern, const IvfFusedParams& p, size_t lds, hipStream_t stream) {
    HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)(p.nq * p.G)), dim3(FB), lds, stream, p);
}

void launch_ivf_fused(const IvfFusedParams& p, hipStream_t stream) {
    if (p.nq == 0) return;
    FA_THROW_IF_NOT(p.G >= 1 && p.npc >= 1 && p.G * p.npc >= p.nprobe);
    FA_THROW_IF_NOT(p.cap >= p.k + FB && p.cap <= FMAXR * FB);
    FA_THROW_IF_NOT(p.nlut == 1 || p.nlut == 2);
    const size_t lds = ivf_fused_lds_bytes(p.kind, p.M, p.dpad, p.kp, p.cap, p.nprobe, p.nlut);
    FA_THROW_IF_NOT_MSG(lds <= 160 * 1024, "fused IVF scan does not fit the LDS");
    const bool l2 = p.metric == METRIC_L2;
    if (p.kind == 0) {
        if (l2) launch_one(ivfflat_fused_kernel<METRIC_L2>, p, lds, stream);
        else launch_one(ivfflat_fused_kernel<METRIC_INNER_PRODUCT>, p, lds, stream);
    } else {
        const bool regs = p.d <= 128 && (p.M % FPART) == 0 && p.dsub * p.M == p.d;
        int ds = regs ? p.dsub : 0;
        if (!(ds == 1 || ds == 2 || ds == 4 || ds == 8 || ds == 16 || ds == 32)) ds = 0;
#define FA_PQ(DS)                                                                               \
    do {                                                                                        \
        if (l2) launch_one(ivfpq_fused_kernel<METRIC_L2, DS, false>, p, lds, stream);            \
        else launch_one(ivfpq_fused_kernel<METRIC_INNER_PRODUCT, DS, false>, p, lds, stream);    \
    } while (0)
        if (p.M == 64 && ds == 2) {
            // PQ64 on d = 128: compile-time M
            if (p.dbg && l2) launch_one(ivfpq_fused_kernel<METRIC_L2, 2, true, true>, p, lds, stream); // phase timing
            else if (l2) launch_one(ivfpq_fused_kernel<METRIC_L2, 2, true>, p, lds, stream);
            else launch_one(ivfpq_fused_kernel<METRIC_INNER_PRODUCT, 2, true>, p, lds, stream);
            HIP_CHECK(hipGetLastError());
            return;
        }
        switch (ds) {
            case 1: FA_PQ(1); break;
            case 2: FA_PQ(2); break;
            case 4: FA_PQ(4); break;
            case 8: FA_PQ(8); break;
            case 16: FA_PQ(16); break;
            case 32: FA_PQ(32); break;
            default: FA_PQ(0); break;
        }
#undef FA_PQ
    }
    HIP_CHECK(hipGetLastError());
}

} // namespace faiss_amd
