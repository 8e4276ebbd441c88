/* faiss_amd/csrc/faiss_amd_internal.h -- test and tuning hooks of libfaiss_amd.so that have NO counterpart in the
 * reference's interface.  They are exported (the Python test-suite and tools/ bind them through ctypes) but they are not
 * part of the drop-in boundary: include/faiss_amd_c.h declares only what a reference-side binding would use.  Same error
 * convention as the public header (0 ok, -2 FaissAmdException, -4 std::exception, -1 unknown; faiss_amd_get_last_error). */
#ifndef FAISS_AMD_INTERNAL_H
#define FAISS_AMD_INTERNAL_H
#include "../../include/faiss_amd_c.h"
#ifdef __cplusplus
extern "C" {
#endif

/* approximate scores [n][ntotal] of the filter kernel (L2: <q,y> - |y|^2/2 on fp16 inputs; IP: <q,y>) and
 * the per-query bound err_bound[n] on their deviation from the exact fp32 scores */
int faiss_amd_GpuIndexFlat_filter_scores(const FaissAmdIndex* index, faiss_amd_idx_t n, const float* x, float* scores,
                                         float* err_bound);

/* the k best entries (smallest for L2, largest for inner product; ties to the lower column) of every row of the host
 * matrix vals [rows][cols] through one selection primitive in isolation -- the stand-alone select test of the reference
 * (faiss/gpu/test/TestGpuSelect.cu:23-198, runBlockSelect / runWarpSelect).  which: 0 = select_k_kernel (BlockSelect's
 * role), 1 = workgroup LDS reservoir (fused IVF scans), 2 = wavefront select (flat scan reservoirs; winners unordered) */
int faiss_amd_test_select(FaissAmdGpuResources* res, int which, FaissAmdMetricType metric, int rows, int cols, int k,
                          const float* vals, float* out_distances, faiss_amd_idx_t* out_indices);

/* Tuning experiments of the filter path (tools/lmf_sweep.py; results never change, only timings): rows of a list per
 * work item, 32-row blocks per granule (1, 2, 4, 8), candidate room per query, and the block sampling stride of the
 * first sweep (it may bound the k-th best estimate from every min_stride-th 32-row block).  0 = the built-in rule. */
int faiss_amd_GpuIndexIVF_set_lmf_tuning(FaissAmdIndex* index, int rows_per_item, int gran_blocks, int cand_cap, int min_stride);

/* Sweep 1 of the filter path on a sample of the rows: the first rows_per_item >> sample_shift rows of every work item
 * (1 ... 4), 0 = the built-in rule (a quarter for lists of >= 1024 rows on average), -1 = every row.  Results never change. */
int faiss_amd_GpuIndexIVF_set_lmf_sampling(FaissAmdIndex* index, int sample_shift);
/* A/B knob of the IVFPQ sweeps at PQ64 over d = 128: the fp16 codebook twice in LDS with different code -> bank maps and a copy
 * choice stored with the sweeps' copy of the codes (on) against the one-copy sweeps (off, the DEFAULT: measured slower).  Results never change. */
int faiss_amd_GpuIndexIVFPQ_set_lmf_two_copies(FaissAmdIndex* index, int on);
/* A/B knob of the IVFFlat / scalar-quantizer filter sweeps (rows of <= 128 coordinates): two-wave workgroups in lock-step over the
 * query groups of a (list, row chunk): 0 = a free-running wavefront per item, 1 = sweep 1 only (the default), 2 = both sweeps. */
int faiss_amd_GpuIndexIVF_set_lmf_pair(FaissAmdIndex* index, int on);
/* A/B knob of flat searches over <= 4096 rows with k <= 64 (the coarse quantizer of an IVF index; `index` = a GpuIndexFlat or a
 * GpuIndexIVF, whose quantizer is meant): the one-launch kernel of round 6 (on, the default) against the general launches. */
int faiss_amd_Index_set_small_fused(FaissAmdIndex* index, int on);
/* A/B knob of the IVFPQ sweeps at PQ64 over d = 128 (one copy): one-instruction codebook gathers (on, the default since round 6)
 * against the gathers as hipcc compiles the C++ (off).  Results never change. */
int faiss_amd_GpuIndexIVFPQ_set_lmf_fast_gather(FaissAmdIndex* index, int on);
/* Test hook of the f16 filter (no reference counterpart): for n host queries, the ESTIMATED distance of every row they
 * probe as a key (ordkey(estimate) << 32 | scan position) at keys_out[q * stride + scan position] (slots nobody owns
 * hold ~0), and band_out[q] = the error band the filter grants query q (|estimate - exact| <= band is what makes the
 * collected rows a superset of the answer; tests/test_gpu_listmajor.py::test_list_filter_error_bound_holds).  band_out
 * is read first: queries whose probed lists hold fewer than k granules keep the caller's value. */
int faiss_amd_GpuIndexIVF_test_filter_dump(const FaissAmdIndex* index, int64_t n, const float* x, int nprobe, int64_t k, int64_t stride,
                                           uint64_t* keys_out, float* band_out);

/* Test hook of the scalar quantizer's host-side range statistics (RS_meanstd 1 / RS_quantiles 2 / RS_optim 3,
 * faiss/impl/scalar_quantizer/training.cpp:235-332): `trained` of faiss::ScalarQuantizer(d, qtype)::train(n, rows) -- 2 floats for
 * the uniform types, 2 d otherwise.  Pure host code: callable without a GPU (tests/test_oracle_cpu.py pins it on the reference). */
int faiss_amd_sq_train_rangestat(int qtype, int rangestat, float rangestat_arg, int64_t n, int d, const float* rows, float* trained_out);

#ifdef __cplusplus
}
#endif
#endif
