// tools/mfma_probe.hip -- where does the time of the flat filter loop go?  A stand-alone copy of its
// SKELETON (8 waves x 128 queries, 64-row fp16 tiles through a 3-slot LDS ring, 64 MFMAs per wave and
// tile) whose ingredients can be switched on one at a time:
//   bit 0  A operands read from LDS (ds_read_b128, swizzled) instead of constant registers
//   bit 1  s_barrier per tile
//   bit 2  LDS-DMA prefetch of the next-but-one tile (global_load_lds_dwordx4) + counted vmcnt wait
//   bit 3  epilogue: max of the 16 scores of every accumulator + compare + (never taken) branch
//   bit 4  accumulators start from an LDS bias quad instead of 0
//   bit 5  the whole tile is staged by wave 0
//   bit 6  software-pipelined epilogue: the max3/compare of block n sits between the first-k-step MFMAs of block n+1
//          (per 32-query column block: epilogue(acc[qb]) then the MFMA that overwrites acc[qb]), across the barrier too
// build:  hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o tools/mfma_probe.bin   (run on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                 \
    do {                                                                      \
        hipError_t e_ = (x);                                                  \
        if (e_ != hipSuccess) {                                               \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));           \
            exit(1);                                                          \
        }                                                                     \
    } while (0)

__device__ __forceinline__ float max3(float a, float b, float c) {
    float r;
    asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ void glds16_s(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
            "s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(voff), "s"(sbase), "s"(lds_dst)
            : "memory");
}
__device__ __forceinline__ const char* uniform_ptr(const char* ptr) {
    const unsigned long long v = (unsigned long long)ptr;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const char*)(((unsigned long long)hi << 32) | lo);
}

constexpr int TILE = 16384;

template <int F, int WAVES, int RING>
__global__ void __launch_bounds__(WAVES * 64, 1)
probe(const _Float16* __restrict__ xb, const _Float16* __restrict__ xq, float* out, int nsteps, int nsplit, float th) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31;
    const int split = blockIdx.x % nsplit;
    half8 bq[4][8];
#pragma unroll
    for (int qb = 0; qb < 4; ++qb)
#pragma unroll
        for (int s = 0; s < 8; ++s)
            bq[qb][s] = *(const half8*)(xq + ((size_t)(blockIdx.x * 64 + wave * 4 + qb) * 32 + j) % 4096 * 128 + s * 16 + h * 8);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (int i = tid; i < (RING * TILE + 1024) / 4; i += WAVES * 64) ((float*)smem)[i] = 0.001f * (i & 255);
    __syncthreads();
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) char*)smem);
    constexpr bool ONE = (F & 32) != 0;            // wave 0 issues the whole tile
    constexpr int DMA_ROWS = ONE ? 16 : 16 / WAVES;
    unsigned voff[DMA_ROWS];
#pragma unroll
    for (int i = 0; i < DMA_ROWS; ++i) {
        const int g = ((ONE ? 0 : wave) * DMA_ROWS + i) * 64 + lane;
        const int row = g >> 4, cpos = g & 15, c = cpos ^ (row & 15);
        voff[i] = (unsigned)(row * 256 + c * 16);
    }
    auto stage = [&](int u, int slot) __attribute__((always_inline)) {
        const int row0 = __builtin_amdgcn_readfirstlane((split + u * nsplit) * 64);
        const char* sb = uniform_ptr((const char*)xb + (size_t)row0 * 256);
        if (ONE && wave != 0) return;
#pragma unroll
        for (int i = 0; i < DMA_ROWS; ++i) glds16_s(sb, voff[i], lds_base + slot * TILE + ((ONE ? 0 : wave) * DMA_ROWS + i) * 1024);
    };
    float keep = 0.f;
    int hits = 0;
    if (F & 4) {
#pragma unroll
        for (int t = 0; t < RING - 1; ++t) stage(t, t);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 2) * DMA_ROWS) : "memory");
        __syncthreads();
    }
    half8 areg;
#pragma unroll
    for (int i = 0; i < 8; ++i) areg[i] = (_Float16)(0.01f * (lane + i));
    int slot = 0;
    f32x16 acc[4];
#pragma unroll
    for (int qb = 0; qb < 4; ++qb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[qb][r] = 0.f;
    for (int u = 0; u < nsteps; ++u) {
        const int slot2 = slot >= 1 ? slot - 1 : RING - 1;
        if (F & 4) {
            if (u + RING - 1 < nsteps) stage(u + RING - 1, slot2);
        }
        const char* tile = smem + slot * TILE;
        const float* bias = (const float*)(smem + RING * TILE);
        const int sw = j & 15;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            f32x16 c0;
            if (F & 16) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 b4 = *(const f32x4*)(bias + rb * 32 + 8 * g + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) c0[4 * g + e] = b4[e];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) c0[r] = 0.f;
            }
            const char* rowp = tile + (rb * 32 + j) * 256;
            auto epi = [&](const f32x16& a) __attribute__((always_inline)) {
                const float m = max3(max3(max3(a[0], a[1], a[2]), max3(a[3], a[4], a[5]), a[15]), max3(a[6], a[7], a[8]),
                                     max3(max3(a[9], a[10], a[11]), max3(a[12], a[13], a[14]), a[12]));
                if (__builtin_expect(__ballot(m > th) != 0ull, 0)) {
                    hits++;
                    keep += m;
                }
            };
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                half8 a0;
                if (F & 1) a0 = *(const half8*)(rowp + (((2 * s + h) ^ sw) << 4));
                else {
                    a0 = areg;
                    asm volatile("" : "+v"(a0));
                }
#pragma unroll
                for (int qb = 0; qb < 4; ++qb) {
                    if ((F & 64) && s == 0) {
                        // the previous block's scores of this column block, then the MFMA that overwrites them
                        if (u > 0 || rb > 0) {
                            asm volatile("s_nop 7" ::"v"(acc[qb]));
                            epi(acc[qb]);
                        }
                        acc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bq[qb][s], c0, 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    } else {
                        acc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bq[qb][s], s == 0 ? c0 : acc[qb], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!(F & 64)) {
                if (F & 8) {
                    asm volatile("s_nop 15\n\ts_nop 3" ::"v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]));
#pragma unroll
                    for (int qb = 0; qb < 4; ++qb) epi(acc[qb]);
                } else {
#pragma unroll
                    for (int qb = 0; qb < 4; ++qb) asm volatile("" ::"v"(acc[qb]));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (F & 4) {
            if (u + RING <= nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 2) * DMA_ROWS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (F & 2) __syncthreads();
        slot = slot == RING - 1 ? 0 : slot + 1;
    }
    if (F & 64) {
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) keep += acc[qb][0] > th ? 1.f : 0.f;
    }
    if (hits == 12345) out[blockIdx.x * WAVES * 64 + tid] = keep;
}

template <int F, int WAVES, int RING = 3>
static void run(const char* name, const _Float16* xb, const _Float16* xq, float* out, int nwg, int nsteps, int nsplit) {
    const size_t lds = WAVES == 4 ? 100 * 1024 : RING * TILE + 1024; // (4 waves: one workgroup per CU all the same)
    CK(hipFuncSetAttribute((const void*)probe<F, WAVES, RING>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((probe<F, WAVES, RING>), dim3(nwg), dim3(WAVES * 64), lds, 0, xb, xq, out, nsteps, nsplit, 1e30f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((probe<F, WAVES, RING>), dim3(nwg), dim3(WAVES * 64), lds, 0, xb, xq, out, nsteps, nsplit, 1e30f);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    // flops: per WG and step WAVES waves x 64 MFMAs x 32768
    const double fl = (double)nwg * nsteps * WAVES * 64 * 32768.0;
    const double rounds = (double)((nwg + 255) / 256);
    printf("%-44s waves=%d wgs=%d steps=%d  %.3f ms  %.0f TFLOP/s  %.0f ns/step/WG-round\n", name, WAVES, nwg, nsteps, ms,
           fl / ms / 1e9, ms * 1e6 / (rounds * nsteps));
}

int main(int argc, char** argv) {
    const int nwg = argc > 1 ? atoi(argv[1]) : 512, nsteps = argc > 2 ? atoi(argv[2]) : 320, nsplit = 48;
    const size_t nrows = (size_t)nsplit * nsteps * 64 + 64;
    _Float16 *xb, *xq;
    float* out;
    CK(hipMalloc(&xb, nrows * 256));
    CK(hipMalloc(&xq, 4096 * 256));
    CK(hipMalloc(&out, (size_t)nwg * 512 * 4));
    std::vector<_Float16> hb(nrows * 128), hq(4096 * 128);
    for (size_t i = 0; i < hb.size(); ++i) hb[i] = (_Float16)(0.001f * (float)((i * 2654435761u) % 2000) - 1.f);
    for (size_t i = 0; i < hq.size(); ++i) hq[i] = (_Float16)(0.001f * (float)((i * 40503u) % 2000) - 1.f);
    CK(hipMemcpy(xb, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(xq, hq.data(), hq.size() * 2, hipMemcpyHostToDevice));
    run<0, 8>("mfma only (A in registers)", xb, xq, out, nwg, nsteps, nsplit);
    run<0, 4>("mfma only, 1 wave per SIMD", xb, xq, out, nwg, nsteps, nsplit);
    run<1, 8>("+ A from LDS", xb, xq, out, nwg, nsteps, nsplit);
    run<3, 8>("+ A from LDS + barrier", xb, xq, out, nwg, nsteps, nsplit);
    run<7, 8>("+ A from LDS + barrier + DMA", xb, xq, out, nwg, nsteps, nsplit);
    run<15, 8>("+ A from LDS + barrier + DMA + epilogue", xb, xq, out, nwg, nsteps, nsplit);
    run<31, 8>("+ ... + bias as C operand (= the kernel)", xb, xq, out, nwg, nsteps, nsplit);
    run<9, 8>("A from LDS + epilogue, no barrier, no DMA", xb, xq, out, nwg, nsteps, nsplit);
    run<8, 8>("mfma (A regs) + epilogue", xb, xq, out, nwg, nsteps, nsplit);
    run<5, 8>("A from LDS + DMA, no barrier (racy, timing only)", xb, xq, out, nwg, nsteps, nsplit);
    run<7, 8, 4>("LDS + barrier + DMA, 4-slot ring", xb, xq, out, nwg, nsteps, nsplit);
    run<7, 8, 6>("LDS + barrier + DMA, 6-slot ring", xb, xq, out, nwg, nsteps, nsplit);
    run<7 | 32, 8, 3>("LDS + barrier + DMA issued by wave 0 only", xb, xq, out, nwg, nsteps, nsplit);
    run<7 | 32, 8, 4>("LDS + barrier + DMA by wave 0, 4-slot ring", xb, xq, out, nwg, nsteps, nsplit);
    run<31, 8, 4>("the kernel, 4-slot ring", xb, xq, out, nwg, nsteps, nsplit);
    run<31 | 32, 8, 4>("the kernel, 4-slot ring, DMA by wave 0", xb, xq, out, nwg, nsteps, nsplit);
    run<31, 8, 3>("the kernel (again)", xb, xq, out, nwg, nsteps, nsplit);
    run<31 | 64, 8, 3>("the kernel, software-pipelined epilogue", xb, xq, out, nwg, nsteps, nsplit);
    run<15 | 64, 8, 3>("same without the bias operand", xb, xq, out, nwg, nsteps, nsplit);
    return 0;
}
