// tools/dpp_scan_probe.hip -- checks common.h wave_incl_scan (DPP row_shr / row_bcast adds) against the __shfl_up ladder on
// the device: random counts, 4096 wavefronts.
//   hipcc --offload-arch=gfx950 -O3 -I faiss_amd/csrc tools/dpp_scan_probe.hip -o gpurun_out/dpp_scan_probe && gpurun_out/dpp_scan_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "common.h"
using namespace faiss_amd;
__global__ void probe(const unsigned* in, unsigned* out_dpp, unsigned* out_shfl) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    const unsigned v = in[i];
    out_dpp[i] = wave_incl_scan(v);
    unsigned inc = v;
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned o = __shfl_up(inc, off, 64);
        if ((int)threadIdx.x >= off) inc += o;
    }
    out_shfl[i] = inc;
}
int main() {
    const int W = 4096, N = W * 64;
    std::vector<unsigned> h(N), a(N), b(N);
    unsigned s = 12345u;
    for (int i = 0; i < N; ++i) {
        s = s * 1664525u + 1013904223u;
        h[i] = (i / 64) % 3 == 0 ? (s >> 28) : (i / 64) % 3 == 1 ? (s >> 12) : ((s >> 31) ? 1u : 0u);
    }
    unsigned *d, *o1, *o2;
    hipMalloc(&d, N * 4); hipMalloc(&o1, N * 4); hipMalloc(&o2, N * 4);
    hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(W), dim3(64), 0, 0, d, o1, o2);
    hipMemcpy(a.data(), o1, N * 4, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), o2, N * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < N; ++i) {
        unsigned ref = h[i] + (i % 64 ? 0u : 0u);
        (void)ref;
        if (a[i] != b[i]) { if (bad < 5) printf("mismatch at wave %d lane %d: dpp %u shfl %u\n", i / 64, i % 64, a[i], b[i]); ++bad; }
    }
    // and against the host
    for (int w = 0; w < W; ++w) { unsigned run = 0; for (int l = 0; l < 64; ++l) { run += h[w * 64 + l]; if (a[w * 64 + l] != run) ++bad; } }
    printf("dpp scan probe: %d mismatches in %d lanes\n", bad, N);
    return bad ? 1 : 0;
}
