// tools/mfma_f32_probe.hip -- what does v_mfma_f32_32x32x2_f32 sustain on this chip?  Dependent chains of MFMAs (the shape
// of the list-major / exact flat scans: one accumulator per 32-row block), NACC independent accumulators per wave, WAVES
// waves per workgroup, one workgroup per CU.   build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f32_probe.hip -o tools/mfma_f32_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void __launch_bounds__(1024) probe(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n)
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-9f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[n], 0, 0, 0);
    }
    float s = 0.f;
    for (int n = 0; n < NACC; ++n)
        for (int r = 0; r < 16; ++r) s += acc[n][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int waves, int iters, float* d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 256;
    hipLaunchKernelGGL(probe<NACC>, dim3(grid), dim3(waves * 64), 0, 0, d, 10, 1.f, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<NACC>, dim3(grid), dim3(waves * 64), 0, 0, d, iters, 1.0001f, 0.9999f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)grid * waves * iters * 16.0 * NACC * 4096.0;
    printf("waves/CU %2d, accumulators/wave %d: %.3f ms, %.1f TFLOP/s (%.1f%% of 157.3)\n", waves, NACC, ms, flop / ms / 1e9,
           flop / ms / 1e9 / 157.3 * 100);
}
// the same chain fed like the scans feed it: A operands re-read from LDS (random data, ds_read_b128 per 4 MFMAs), B operands
// 64 random registers per lane
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void __launch_bounds__(1024) probe_lds(float* out, const float* in, int iters) {
    __shared__ __attribute__((aligned(16))) float tile[64 * 128];
    for (int i = threadIdx.x; i < 64 * 128; i += blockDim.x) tile[i] = in[(blockIdx.x * 8192 + i) & 0xfffff];
    f32x4 bq[16];
    for (int s = 0; s < 16; ++s)
        for (int e = 0; e < 4; ++e) bq[s][e] = in[(threadIdx.x * 64 + s * 4 + e + blockIdx.x * 977) & 0xfffff];
    __syncthreads();
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n)
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    const int lane = threadIdx.x & 63, h = lane >> 5, j = lane & 31, sw = j & 15;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NACC; ++n) {
            const char* rowp = (const char*)tile + (((n & 1) * 32 + j) * 512);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const f32x4 a = *(const f32x4*)(rowp + (((2 * s + h) ^ sw) << 4));
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], bq[s][e], acc[n], 0, 0, 0);
            }
        }
        // keep the accumulators bounded and the loop honest
        if ((it & 15) == 15)
            for (int n = 0; n < NACC; ++n)
                for (int r = 0; r < 16; ++r) acc[n][r] *= 1e-3f;
    }
    float sum = 0.f;
    for (int n = 0; n < NACC; ++n)
        for (int r = 0; r < 16; ++r) sum += acc[n][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}
template <int NACC>
void run_lds(int waves, int wgs_per_cu, int iters, float* d, const float* in) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 256 * wgs_per_cu;
    hipLaunchKernelGGL(probe_lds<NACC>, dim3(grid), dim3(waves * 64), 0, 0, d, in, 4);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe_lds<NACC>, dim3(grid), dim3(waves * 64), 0, 0, d, in, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)grid * waves * iters * 64.0 * NACC * 4096.0;
    printf("LDS-fed, random data: %d x %d waves/CU, %d chains of 64 MFMAs per wave and round: %.3f ms, %.1f TFLOP/s (%.1f%% of 157.3)\n", wgs_per_cu,
           waves, NACC, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100);
}
int main() {
    float* d;
    hipMalloc(&d, 1024 * 1024 * 4);
    {
        float* in;
        hipMalloc(&in, (1 << 20) * 4);
        float* hin = new float[1 << 20];
        unsigned x = 12345u;
        for (int i = 0; i < (1 << 20); ++i) {
            x = x * 1664525u + 1013904223u;
            hin[i] = ((x >> 8) & 0xffff) / 65536.0f - 0.5f;
        }
        hipMemcpy(in, hin, (1 << 20) * 4, hipMemcpyHostToDevice);
        run_lds<1>(4, 1, 600, d, in);
        run_lds<1>(4, 2, 600, d, in);
        run_lds<1>(8, 1, 600, d, in);
        run_lds<2>(4, 2, 300, d, in);
        run_lds<2>(8, 1, 300, d, in);
        run_lds<1>(4, 3, 600, d, in);
    }
    for (int waves : {4, 8, 12, 16}) {
        run<1>(waves, 4000, d);
        run<2>(waves, 2000, d);
        run<4>(waves, 1000, d);
    }
    return 0;
}
