// faiss_amd/csrc/selector_kernels.hip -- faiss::IDSelector (faiss/impl/IDSelector.h:21-215) on the device.
//
// The reference evaluates `sel->is_member(id)` per scanned entry on the CPU (IndexIVF.cpp scan_codes, the
// exhaustive_*_seq loops of utils/distances.cpp); its GPU indexes accept the parameter and ignore it outside the cuVS
// back-end.  Here a search with a selector first turns it into ONE BIT PER STORED ROW (a pass over the ids: 8 bytes per
// row, once per search call), and the scan kernels test that bit only for the rows that would otherwise become
// candidates -- the code / vector stream, which bounds the scans, is untouched.
#include "kernels.h"

namespace faiss_amd {

__device__ __forceinline__ bool sel_eval(const SelProgram& P, int64_t id) {
    unsigned stack = 0; // bit 0 = top of the stack
    for (int i = 0; i < P.n; ++i) {
        const SelInstr& in = P.ins[i];
        if (in.op <= SEL_BITMAP) {
            bool m;
            if (in.op == SEL_ALL) {
                m = true;
            } else if (in.op == SEL_RANGE) {
                m = id >= in.a && id < in.b;
            } else if (in.op == SEL_SET) {
                const int64_t* ids = (const int64_t*)in.ptr;
                int64_t lo = 0, hi = in.a; // first position with ids[pos] >= id
                while (lo < hi) {
                    const int64_t mid = (lo + hi) >> 1;
                    if (ids[mid] < id) lo = mid + 1;
                    else hi = mid;
                }
                m = lo < in.a && ids[lo] == id;
            } else {
                // IDSelectorBitmap::is_member (IDSelector.cpp:123-129): the id as an unsigned number
                const uint64_t u = (uint64_t)id;
                m = (u >> 3) < (uint64_t)in.a && ((((const uint8_t*)in.ptr)[u >> 3] >> (u & 7)) & 1);
            }
            stack = (stack << 1) | (m ? 1u : 0u);
        } else if (in.op == SEL_NOT) {
            stack ^= 1u;
        } else {
            const unsigned b = stack & 1u;
            stack >>= 1;
            const unsigned a = stack & 1u;
            const unsigned r = in.op == SEL_AND ? (a & b) : in.op == SEL_OR ? (a | b) : (a ^ b);
            stack = (stack & ~1u) | r;
        }
    }
    return (stack & 1u) != 0u;
}

__global__ void __launch_bounds__(256) selector_mask_kernel(const int64_t* __restrict__ ids, int64_t n, int64_t id_base,
                                                             SelProgram P, uint64_t* __restrict__ mask,
                                                             unsigned long long* __restrict__ count) {
    // a wavefront owns 64 consecutive rows = one mask word
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool m = i < n && sel_eval(P, ids ? ids[i] : id_base + i);
    const uint64_t b = __ballot(m);
    if ((threadIdx.x & 63) == 0 && (i >> 6) < (n + 63) / 64) {
        mask[i >> 6] = b;
        if (count && b) atomicAdd(count, (unsigned long long)__popcll(b));
    }
}

void launch_selector_mask(const int64_t* ids, int64_t n, int64_t id_base, const SelProgram& prog, uint64_t* mask,
                          unsigned long long* count, hipStream_t stream) {
    if (n == 0) return;
    FA_THROW_IF_NOT(prog.n >= 1 && prog.n <= kSelMaxInstr);
    hipLaunchKernelGGL(selector_mask_kernel, dim3((unsigned)div_up((size_t)n, 256)), dim3(256), 0, stream, ids, n, id_base,
                       prog, mask, count);
    HIP_CHECK(hipGetLastError());
}

__global__ void mask_bias_kernel(const float* __restrict__ src, const uint32_t* __restrict__ mask, int64_t n, int npad,
                                 float excluded, float pad, float* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = ((mask[i >> 5] >> (i & 31)) & 1u) ? src[i] : excluded;
    else if (i < n + npad) dst[i] = pad;
}

void launch_mask_bias(const float* src, const uint32_t* mask, int64_t n, int npad, float excluded, float pad, float* dst,
                      hipStream_t stream) {
    if (n + npad == 0) return;
    hipLaunchKernelGGL(mask_bias_kernel, dim3((unsigned)div_up((size_t)(n + npad), 256)), dim3(256), 0, stream, src, mask, n,
                       npad, excluded, pad, dst);
    HIP_CHECK(hipGetLastError());
}

} // namespace faiss_amd
