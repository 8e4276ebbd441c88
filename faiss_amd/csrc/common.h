// faiss_amd/csrc/common.h -- shared host/device definitions for the MI355X (gfx950) backend.
//
// This backend re-implements the search path of the reference's GpuIndexFlat /
// GpuIndexIVFFlat / GpuIndexIVFPQ (reference: faiss/gpu/GpuIndex.h:49-176,
// faiss/Index.h:101-431) from scratch for CDNA4.  Nothing here is derived from
// faiss/gpu sources; only the public contract (argument meaning, result layout,
// error behaviour) is mirrored.
#pragma once

#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>

namespace faiss_amd {

using idx_t = int64_t; // reference: faiss/MetricType.h:52

// Same numeric values as the reference's faiss::MetricType (faiss/MetricType.h:23-28).
enum MetricType : int {
    METRIC_INNER_PRODUCT = 0,
    METRIC_L2 = 1,
    // the "extra" metrics of the flat index / bfKnn (faiss/gpu/impl/GeneralDistance.cuh over the functors of
    // faiss/gpu/impl/DistanceUtils.cuh:47-281): no GEMM form, one pass over the dimensions per (query, vector) pair
    METRIC_L1 = 2,
    METRIC_Linf = 3,
    METRIC_Lp = 4, // p = Index::metric_arg
    METRIC_Canberra = 20,
    METRIC_BrayCurtis = 21,
    METRIC_JensenShannon = 22,
    METRIC_Jaccard = 23, // a similarity: larger is better (faiss::is_similarity_metric, faiss/MetricType.h:59-62)
};
static inline bool is_general_metric(int m) {
    return m == METRIC_L1 || m == METRIC_Linf || m == METRIC_Lp || m == METRIC_Canberra || m == METRIC_BrayCurtis ||
           m == METRIC_JensenShannon || m == METRIC_Jaccard;
}
// The one predicate every constructor / bfKnn validates its metric with (also exported as faiss_amd_metric_supported, so
// that the argument-validation expectations of the GPU tests can be checked without a device: tests/test_abi_cpu.py).
// index_kind: 0 = GpuIndexFlat / bfKnn (faiss/gpu/GpuIndexFlat.cu, GpuDistance.cu), 1 = the IVF indexes
// (faiss/gpu/GpuIndexIVF.cu:35-37: L2 and inner product only).
static inline bool metric_supported(int index_kind, int m) {
    if (m == METRIC_L2 || m == METRIC_INNER_PRODUCT) return true;
    return index_kind == 0 && is_general_metric(m);
}
// the order results are kept in: a similarity is searched like the inner product (descending, padded with -FLT_MAX),
// a distance like L2
static inline int order_metric(int m) {
    return (m == METRIC_INNER_PRODUCT || m == METRIC_Jaccard) ? METRIC_INNER_PRODUCT : METRIC_L2;
}

// Mirrors faiss::FaissException (faiss/impl/FaissException.h:20-43): thrown by host code,
// translated to error code -2 at the C ABI (reference convention c_api/macros_impl.h:22-56).
struct FaissAmdException : public std::runtime_error {
    explicit FaissAmdException(const std::string& m) : std::runtime_error(m) {}
};

// Experiment / diagnostic knobs (FAISS_AMD_* environment variables) change timings and, some of them, RESULTS: a stray
// variable in a production environment must not.  EVERY knob of the library is read through this gate: it answers only
// when FAISS_AMD_EXPERIMENTS=1 is set too (tools/ and the two tests that drive a knob set it).
static inline const char* experiment_env(const char* name) {
    const char* e = getenv("FAISS_AMD_EXPERIMENTS");
    return (e && e[0] == '1' && e[1] == 0) ? getenv(name) : nullptr;
}

// device allocation failure: lets a caller with an alternative (the query-major scan instead of the filter sweeps' shadow
// copies) take it; everything else sees a FaissAmdException
struct DeviceOutOfMemory : public FaissAmdException {
    explicit DeviceOutOfMemory(const std::string& m) : FaissAmdException(m) {}
};

// This code base is written for gfx950 (CDNA4) and nothing else: inline assembly with gfx942+ cache-policy bits (`sc0`),
// DPP row broadcasts, the 160 KB LDS, v_mfma_f32_32x32x16_f16.  Refuse any other device target at compile time.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "faiss_amd is written for gfx950 (MI355X); build with --offload-arch=gfx950"
#endif

#define FA_STR2(x) #x
#define FA_STR(x) FA_STR2(x)
#define FA_THROW_MSG(msg)                                                          \
    throw ::faiss_amd::FaissAmdException(                                          \
            std::string("Error in ") + __func__ + " at " __FILE__ ":" FA_STR(      \
                    __LINE__) ": " + (msg))
#define FA_THROW_IF_NOT_MSG(cond, msg) \
    do {                               \
        if (!(cond)) {                 \
            FA_THROW_MSG(std::string("'" #cond "' failed: ") + (msg)); \
        }                              \
    } while (0)
#define FA_THROW_IF_NOT(cond) FA_THROW_IF_NOT_MSG(cond, "")

#define HIP_CHECK(expr)                                                              \
    do {                                                                             \
        hipError_t _e = (expr);                                                      \
        if (_e != hipSuccess) {                                                      \
            FA_THROW_MSG(std::string("HIP error: ") + hipGetErrorString(_e) + " in " \
                         #expr);                                                     \
        }                                                                            \
    } while (0)

// Limits mirrored from the reference GPU path: k and nprobe are capped at 2048
// (faiss/gpu/utils/DeviceDefs.cuh:39, faiss/gpu/impl/IndexUtils.cu:28-42).
constexpr int kMaxSelectionK = 2048;

static inline size_t round_up(size_t x, size_t m) {
    return (x + m - 1) / m * m;
}
static inline size_t div_up(size_t x, size_t m) {
    return (x + m - 1) / m;
}

// ---------------------------------------------------------------------------------
// Sortable keys.  Every candidate that survives a scan is represented by one 64-bit
// key = (ordkey(distance) << 32) | payload, compared as an unsigned integer.
// ordkey() maps a float to a uint32 whose unsigned order is the search order:
//   L2 (smaller is better): ascending float order
//   IP (larger is better):  descending float order
// so "k smallest keys" is always the answer, and ties on distance are resolved by the
// payload (vector id for Flat => the total order (distance, id) of the reference CPU
// heap, faiss/utils/ordered_key_value.h:74-76 + faiss/impl/ResultHandler.h:276-281).
// -0.0f is canonicalised to +0.0f.
// ---------------------------------------------------------------------------------
#if defined(__HIPCC__)
// Inclusive prefix sum over the 64 lanes of a wavefront -- ALL lanes active (call it from wave-uniform control flow).  Six
// DPP adds: row_shr 1 / 2 / 4 / 8 inside the 16-lane rows (lanes the shift leaves without a source read 0), then row_bcast 15
// into rows 1 and 3 and row_bcast 31 into rows 2 and 3.  The __shfl_up ladder it replaces is six ds_bpermute round trips
// through the LDS crossbar (~100 cycles each, serialised by lgkmcnt(0)): the candidate parking of the list-major sweeps
// runs one scan per (32-row block, 32-query block) that holds a candidate -- nearly all of them at nb <= 10M.
// tools/dpp_scan_probe.hip checks it against the shuffle ladder on the device.
// A `volatile` access through a plain (generic) pointer to LDS stays a FLAT instruction -- hipcc's address-space inference does not
// rewrite volatile accesses -- i.e. it goes through the vector-memory path with an aperture check, is waited for with vmcnt(0)
// (which also drains every LDS-DMA / global load in flight) and, for stores and atomics, is not covered by the lgkmcnt(0) in
// front of an s_barrier (wg_select.h).  Volatile LDS accesses therefore go through a pointer typed to the LDS
// address space: a ds_read / ds_write that the compiler still may not cache or drop.
typedef volatile __attribute__((address_space(3))) unsigned lds_volatile_u32;
__device__ __forceinline__ lds_volatile_u32* lds_volatile(const void* p) {
    return (lds_volatile_u32*)(__attribute__((address_space(3))) void*)p;
}
__device__ __forceinline__ unsigned wave_incl_scan(unsigned v) {
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true); // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true); // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true); // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true); // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false); // row_bcast:15 -> rows 1, 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false); // row_bcast:31 -> rows 2, 3
    return (unsigned)x;
}
#endif

__host__ __device__ static inline uint32_t float_flip(float f) {
    union {
        float f;
        uint32_t u;
    } v;
    v.f = (f == 0.0f) ? 0.0f : f;
    return (v.u & 0x80000000u) ? ~v.u : (v.u | 0x80000000u);
}
// Keys at or above this value are "not better than the neutral element" for both metrics
// (+/-FLT_MAX, infinities, NaN): the reference never admits them (strict compare against the
// heap's neutral value), so they are reported as missing results.
constexpr uint32_t kInvalidOrdKey = 0xff7fffffu;
__host__ __device__ static inline bool float_isnan(float f) {
    return f != f;
}
__host__ __device__ static inline float float_unflip(uint32_t u) {
    union {
        float f;
        uint32_t u;
    } v;
    v.u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return v.f;
}
template <int METRIC>
__host__ __device__ static inline uint32_t ordkey(float dis) {
    if (float_isnan(dis)) return 0xffffffffu;
    return METRIC == METRIC_L2 ? float_flip(dis) : ~float_flip(dis);
}
template <int METRIC>
__host__ __device__ static inline float unordkey(uint32_t k) {
    return METRIC == METRIC_L2 ? float_unflip(k) : float_unflip(~k);
}
__host__ __device__ static inline uint32_t ordkey_rt(int metric, float dis) {
    if (float_isnan(dis)) return 0xffffffffu;
    return metric == METRIC_L2 ? float_flip(dis) : ~float_flip(dis);
}
__host__ __device__ static inline float unordkey_rt(int metric, uint32_t k) {
    return metric == METRIC_L2 ? float_unflip(k) : float_unflip(~k);
}

// Value the reference pads missing results with: heap "neutral" element
// (faiss/utils/ordered_key_value.h:57-59, 80-82; faiss/utils/Heap.h:427-457).
__host__ __device__ static inline float neutral_distance(int metric) {
    return metric == METRIC_L2 ? FLT_MAX : -FLT_MAX;
}

} // namespace faiss_amd
