/* include/faiss_amd_c.h -- C ABI of the MI355X (gfx950) similarity-search backend.
 *
 * Drop-in boundary for the search path of the reference's GpuIndexFlat / GpuIndexIVFFlat /
 * GpuIndexIVFPQ behind faiss::Index's add()/search() surface.  Every entry point below names
 * the reference interface it stands in for (paths relative to the faiss v1.15.0 tree).  The
 * naming and the error convention follow the reference's own C API (c_api/Index_c.h,
 * c_api/gpu/ *.h): functions return 0 on success, -2 for a library exception
 * (faiss::FaissException there, FaissAmdException here), -4 for std::exception, -1 otherwise
 * (c_api/macros_impl.h:22-56); the message is read with faiss_amd_get_last_error()
 * (c_api/error_c.h:30 faiss_get_last_error).
 *
 * Pointers: `x`, `distances`, `labels` may be host OR device pointers, as in the reference
 * GPU indexes (faiss/gpu/GpuIndex.h:76-105); plain pointers and sizes only, no C++ or torch
 * types.  idx_t is int64 (faiss/MetricType.h:52).  L2 distances are squared; inner-product
 * results are largest-first; missing results are label -1 and distance +/-FLT_MAX
 * (faiss/utils/Heap.h:427-457).
 */
#ifndef FAISS_AMD_C_H
#define FAISS_AMD_C_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int64_t faiss_amd_idx_t;

/* same values as FaissMetricType (c_api/Index_c.h:32-44) */
/* values of faiss::MetricType (faiss/MetricType.h:31-52).  L2 and inner product: every index type.  The "extra" metrics:
 * GpuIndexFlat and bfKnn only, like the reference (faiss/gpu/impl/GeneralDistance.cuh; GpuIndexIVF refuses them,
 * faiss/gpu/GpuIndexIVF.cu:35-37); Jaccard is a similarity (results best = largest first, padded with -FLT_MAX). */
typedef enum FaissAmdMetricType {
    FAISS_AMD_METRIC_INNER_PRODUCT = 0,
    FAISS_AMD_METRIC_L2 = 1,
    FAISS_AMD_METRIC_L1 = 2,
    FAISS_AMD_METRIC_Linf = 3,
    FAISS_AMD_METRIC_Lp = 4, /* p = metric_arg */
    FAISS_AMD_METRIC_Canberra = 20,
    FAISS_AMD_METRIC_BrayCurtis = 21,
    FAISS_AMD_METRIC_JensenShannon = 22,
    FAISS_AMD_METRIC_Jaccard = 23
} FaissAmdMetricType;

typedef struct FaissAmdIndex_H FaissAmdIndex;                 /* FaissIndex, c_api/Index_c.h:55 */
typedef struct FaissAmdGpuResources_H FaissAmdGpuResources;   /* FaissStandardGpuResources */

/* c_api/error_c.h:30 faiss_get_last_error */
const char* faiss_amd_get_last_error(void);
/* c_api/gpu/DeviceUtils_c.h:22 faiss_get_num_gpus */
int faiss_amd_get_num_gpus(int* p_output);
/* *p_output = 1 when an index of the kind (0 = GpuIndexFlat / bfKnn, 1 = the IVF indexes) accepts the metric, else 0 --
 * the predicate the constructors apply (faiss/gpu/GpuIndexFlat.cu, faiss/gpu/GpuIndexIVF.cu:35-37 "unsupported metric
 * type"; values of faiss/MetricType.h:31-52).  Needs no device. */
int faiss_amd_metric_supported(int index_kind, int metric, int* p_output);

/* ---- resources: c_api/gpu/StandardGpuResources_c.h:24-33 (one stream + scratch per device) */
int faiss_amd_StandardGpuResources_new(FaissAmdGpuResources** p_res, int device);
void faiss_amd_StandardGpuResources_free(FaissAmdGpuResources* res);
/* c_api/gpu/GpuResources_c.h:48 faiss_GpuResources_syncDefaultStreamCurrentDevice */
int faiss_amd_StandardGpuResources_sync(FaissAmdGpuResources* res);
/* c_api/gpu/GpuResources_c.h:33 faiss_GpuResources_getDefaultStream: *p_stream is a hipStream_t */
int faiss_amd_StandardGpuResources_getDefaultStream(FaissAmdGpuResources* res, void** p_stream);
/* c_api/gpu/StandardGpuResources_c.h:41 faiss_StandardGpuResources_setTempMemory */
int faiss_amd_StandardGpuResources_setTempMemory(FaissAmdGpuResources* res, size_t bytes);

/* paged search of host-resident query batches (GpuIndex::searchFromCpuPaged_, faiss/gpu/GpuIndex.cu:554-774): batches of
 * at least min_bytes of queries (reference: 256 MiB; default here 64 MiB) go through pinned double buffers and a copy
 * stream, page_queries queries at a time (0 = automatic).  *p_count (nullable) receives the number of searches that
 * took this path so far. */
int faiss_amd_StandardGpuResources_setPagedSearch(FaissAmdGpuResources* res, size_t min_bytes, int64_t page_queries);
int faiss_amd_StandardGpuResources_getPagedSearchCount(FaissAmdGpuResources* res, int64_t* p_count);
/* faiss::gpu::StandardGpuResources::setDefaultStream (faiss/gpu/StandardGpuResources.h): all work of the indexes of `res`
 * is ordered on `stream` (a hipStream_t) from now on -- hand in the stream that produced device-resident inputs (e.g.
 * torch.cuda.current_stream().cuda_stream) and no cross-stream synchronisation is needed; NULL restores the private
 * stream.  The caller keeps owning the stream. */
int faiss_amd_StandardGpuResources_setDefaultStream(FaissAmdGpuResources* res, void* stream);

/* ---- constructors
 * faiss/gpu/GpuIndexFlat.h:62-72   GpuIndexFlat(resources, dims, metric, config)
 * faiss/gpu/GpuIndexIVFFlat.h:49-57 GpuIndexIVFFlat(resources, dims, nlist, metric, config)
 * faiss/gpu/GpuIndexIVFPQ.h:72-82  GpuIndexIVFPQ(resources, dims, nlist, subQuantizers,
 *                                  bitsPerCode, metric, config)
 * faiss/IndexShards.h:30-33        IndexShards(d, threaded, successive_ids) + add_shard */
int faiss_amd_GpuIndexFlat_new(FaissAmdIndex** p_index, FaissAmdGpuResources* res, int d,
                               FaissAmdMetricType metric);
int faiss_amd_GpuIndexIVFFlat_new(FaissAmdIndex** p_index, FaissAmdGpuResources* res, int d, int nlist,
                                  FaissAmdMetricType metric);
int faiss_amd_GpuIndexIVFPQ_new(FaissAmdIndex** p_index, FaissAmdGpuResources* res, int d, int nlist,
                                int M, int nbits, FaissAmdMetricType metric);
/* faiss/gpu/GpuIndexIVFScalarQuantizer.h:44-52  GpuIndexIVFScalarQuantizer(resources, dims, nlist, qtype, metric,
 * encodeResidual, config).  qtype: faiss::ScalarQuantizer::QuantizerType values (faiss/impl/ScalarQuantizer.h:27-34) of
 * the types the reference GPU index supports (gpu/impl/GpuScalarQuantizer.cuh:20-33): QT_8bit 0, QT_4bit 1,
 * QT_8bit_uniform 2, QT_4bit_uniform 3, QT_fp16 4, QT_8bit_direct 5, QT_6bit 6.  train() learns the coarse centroids and
 * (RS_minmax, the reference default) the value range of the residuals; codes are byte-identical to
 * faiss::ScalarQuantizer::compute_codes with the same `trained`. */
int faiss_amd_GpuIndexIVFScalarQuantizer_new(FaissAmdIndex** p_index, FaissAmdGpuResources* res, int d, int nlist,
                                             int qtype, FaissAmdMetricType metric, int encode_residual);
/* index->sq.qtype, index->by_residual, index->code_size, index->sq.trained.size() (each pointer nullable) */
int faiss_amd_IndexIVFSQ_info(const FaissAmdIndex* index, int* qtype, int* by_residual, size_t* code_size,
                              size_t* trained_size);
/* index->sq.trained (faiss/impl/ScalarQuantizer.h:72-73): {vmin, vdiff} for the uniform types, vmin[d] then vdiff[d]
 * otherwise -- the array copyFrom / copyTo move (gpu/GpuIndexIVFScalarQuantizer.cu copyFrom: sq = index->sq) */
int faiss_amd_IndexIVFSQ_get_trained(const FaissAmdIndex* index, float* out);
int faiss_amd_IndexIVFSQ_copy_trained(FaissAmdIndex* index, const float* trained, size_t n);
/* index->sq.rangestat / rangestat_arg (ScalarQuantizer.h:60-70) used by train(): RS_minmax (0) is a device reduction; RS_meanstd (1),
 * RS_quantiles (2), RS_optim (3) run on the host over the <= 100 000 (residual) training rows -- the reference GPU class trains its
 * ScalarQuantizer on the host for every statistic (gpu/GpuIndexIVFScalarQuantizer.cu:96-160) -- and give the reference's `trained`
 * byte for byte (training.cpp:209-385 restated in its operation order) */
int faiss_amd_IndexIVFSQ_set_rangestat(FaissAmdIndex* index, int rangestat, float rangestat_arg);
/* ---- the same constructors with the reference's config structs (faiss/gpu/GpuIndex.h:30-47 GpuIndexConfig,
 *      GpuIndexFlat.h:24-40 GpuIndexFlatConfig, GpuIndexIVF.h:24-38 GpuIndexIVFConfig, GpuIndexIVFPQ.h:25-49
 *      GpuIndexIVFPQConfig), field for field as plain ints.  What each field means here:
 *   device            must be -1 or the device of `res` (an index lives on its resources' device)
 *   memorySpace       0 = Device.  Unified (1) is refused: 288 GB of HBM is the design point
 *   useFloat16        GpuIndexFlat: vectors stored as fp16 only, queries converted to fp16, distances = the fp32
 *                     distances of those fp16 values (impl/FlatIndex.cu:39-135).  Halves the resident bytes.
 *   storeTransposed   accepted, ignored (the device layout is private to the kernels)
 *   indicesOptions    faiss/gpu/GpuIndicesOptions.h: INDICES_64_BIT (3), INDICES_32_BIT (2; ids are held as 64-bit anyway),
 *                     INDICES_CPU (0; the reference keeps the ids on the host and translates there -- here they stay on the
 *                     device, the results are the same user ids), INDICES_IVF (1: no user ids at all, the label of a result
 *                     is inverted list << 32 | offset in the list, impl/IVFUtilsSelect2.cu:148)
 *   flat_useFloat16   the index's own coarse quantizer stores its centroids as fp16 (GpuIndexIVFConfig::flatConfig.useFloat16,
 *                     faiss/gpu/GpuIndexIVF.h:23-35): queries are rounded to fp16 for the coarse search, residuals are taken
 *                     against the rounded centroids (what the quantizer's reconstruct returns), like the reference
 *   allowCpuCoarseQuantizer  accepted: a coarse quantizer that is not a flat index of this library runs on the reference side
 *                     (integration/faiss_amd_bridge.h: quantizer->search on the host, then search_preassigned / add_core --
 *                     faiss/gpu/impl/IVFBase.cu:526-546)
 *   useFloat16LookupTables   accepted, ignored: the tables are fp32 on a per-query power-of-two grid (>= fp16 accuracy)
 *   usePrecomputedTables     accepted, ignored: the list-dependent L2 term is always kept per stored vector (4 B each),
 *                            which is what precomputed tables buy, without the nlist x M x 256 table
 *   interleavedLayout, useMMCodeDistance  accepted, ignored (layout and table construction are fixed) */
typedef struct FaissAmdGpuIndexFlatConfig {
    int device, memorySpace, useFloat16, storeTransposed;
} FaissAmdGpuIndexFlatConfig;
typedef struct FaissAmdGpuIndexIVFConfig {
    int device, memorySpace, indicesOptions, flat_useFloat16, allowCpuCoarseQuantizer;
} FaissAmdGpuIndexIVFConfig;
typedef struct FaissAmdGpuIndexIVFPQConfig {
    FaissAmdGpuIndexIVFConfig ivf;
    int useFloat16LookupTables, usePrecomputedTables, interleavedLayout, useMMCodeDistance;
} FaissAmdGpuIndexIVFPQConfig;
int faiss_amd_GpuIndexFlat_new_with_config(FaissAmdIndex** p_index, FaissAmdGpuResources* res, int d,
                                           FaissAmdMetricType metric, const FaissAmdGpuIndexFlatConfig* config);
int faiss_amd_GpuIndexIVFFlat_new_with_config(FaissAmdIndex** p_index, FaissAmdGpuResources* res, int d, int nlist,
                                              FaissAmdMetricType metric, const FaissAmdGpuIndexIVFConfig* config);
int faiss_amd_GpuIndexIVFPQ_new_with_config(FaissAmdIndex** p_index, FaissAmdGpuResources* res, int d, int nlist, int M,
                                            int nbits, FaissAmdMetricType metric, const FaissAmdGpuIndexIVFPQConfig* config);
/* GpuIndexIVFFlat / IVFPQ / IVFScalarQuantizer(resources, coarseQuantizer, ...) (faiss/gpu/GpuIndexIVFFlat.h:49-56,
 * GpuIndexIVFPQ.h:70-79, GpuIndexIVFScalarQuantizer.h:47-55; GpuIndexIVF.cu:41-70): the coarse quantizer is the CALLER's flat
 * index of this library on the same device -- NOT owned by the new index (own_fields = false: it must outlive it), possibly
 * shared between indexes, possibly holding its nlist centroids already (the index then needs no coarse training).  null =
 * the index creates and owns one (then identical to *_new_with_config).  config may be null. */
int faiss_amd_GpuIndexIVFFlat_new_with_quantizer(FaissAmdIndex** p_index, FaissAmdGpuResources* res, FaissAmdIndex* coarse_quantizer,
                                                 int d, int nlist, FaissAmdMetricType metric, const FaissAmdGpuIndexIVFConfig* config);
int faiss_amd_GpuIndexIVFPQ_new_with_quantizer(FaissAmdIndex** p_index, FaissAmdGpuResources* res, FaissAmdIndex* coarse_quantizer,
                                               int d, int nlist, int M, int nbits, FaissAmdMetricType metric,
                                               const FaissAmdGpuIndexIVFPQConfig* config);
int faiss_amd_GpuIndexIVFScalarQuantizer_new_with_quantizer(FaissAmdIndex** p_index, FaissAmdGpuResources* res,
                                                            FaissAmdIndex* coarse_quantizer, int d, int nlist, int qtype,
                                                            FaissAmdMetricType metric, int encode_residual,
                                                            const FaissAmdGpuIndexIVFConfig* config);
/* own_fields (0: caller-owned quantizer), whether the quantizer stores fp16, the index's IndicesOptions; pointers nullable */
int faiss_amd_GpuIndexIVF_quantizer_info(const FaissAmdIndex* index, int* own_fields, int* use_float16, int* indices_options);
/* StandardGpuResources::getMemoryInfo (faiss/gpu/StandardGpuResources.cpp:676; the reference returns a map device ->
 * allocation type -> (count, bytes)): live device allocations of the library on the resources' device, their bytes, the peak
 * of those bytes, the temp-memory budget (setTempMemory), and the device's free / total memory; every pointer nullable.
 * setLogMemoryAllocations (StandardGpuResources.cpp:327): one stderr line per device allocation / release. */
int faiss_amd_StandardGpuResources_getMemoryInfo(FaissAmdGpuResources* res, size_t* allocations, size_t* bytes, size_t* peak_bytes,
                                                 size_t* temp_memory, size_t* device_free, size_t* device_total);
int faiss_amd_StandardGpuResources_setLogMemoryAllocations(FaissAmdGpuResources* res, int enable);
/* device bytes held for the database of a flat index (rows, fp16 shadow / storage, norms) */
int faiss_amd_GpuIndexFlat_resident_bytes(const FaissAmdIndex* index, size_t* p_bytes);
int faiss_amd_IndexShards_new(FaissAmdIndex** p_index, int d, int threaded, int successive_ids);
int faiss_amd_IndexShards_add_shard(FaissAmdIndex* shards, FaissAmdIndex* shard);
/* faiss/IndexReplicas.h:20-82: IndexReplicas(d, threaded) + addIndex.  Every replica holds the whole database;
 * search() deals the queries out in ceil(n / count) blocks (IndexReplicas.cpp:123-175).  This is what
 * index_cpu_to_gpu_multiple builds by default (GpuMultipleClonerOptions::shard = false, GpuClonerOptions.h:57-59) */
int faiss_amd_IndexReplicas_new(FaissAmdIndex** p_index, int d, int threaded);
int faiss_amd_IndexReplicas_add_replica(FaissAmdIndex* replicas, FaissAmdIndex* replica);
/* c_api/Index_c.h:56 faiss_Index_free */
void faiss_amd_Index_free(FaissAmdIndex* index);

/* ---- faiss::Index surface: c_api/Index_c.h:58-176, faiss/Index.h:101-431 */
int faiss_amd_Index_d(const FaissAmdIndex* index);
int faiss_amd_Index_is_trained(const FaissAmdIndex* index);
faiss_amd_idx_t faiss_amd_Index_ntotal(const FaissAmdIndex* index);
FaissAmdMetricType faiss_amd_Index_metric_type(const FaissAmdIndex* index);
/* faiss::Index::metric_arg (faiss/Index.h:114; c_api/Index_c.h faiss_Index_metric_arg): the p of METRIC_Lp */
float faiss_amd_Index_metric_arg(const FaissAmdIndex* index);
int faiss_amd_Index_set_metric_arg(FaissAmdIndex* index, float metric_arg);
int faiss_amd_Index_train(FaissAmdIndex* index, faiss_amd_idx_t n, const float* x);
int faiss_amd_Index_add(FaissAmdIndex* index, faiss_amd_idx_t n, const float* x);
int faiss_amd_Index_add_with_ids(FaissAmdIndex* index, faiss_amd_idx_t n, const float* x,
                                 const faiss_amd_idx_t* xids);
int faiss_amd_Index_search(const FaissAmdIndex* index, faiss_amd_idx_t n, const float* x,
                           faiss_amd_idx_t k, float* distances, faiss_amd_idx_t* labels);
int faiss_amd_Index_assign(FaissAmdIndex* index, faiss_amd_idx_t n, const float* x,
                           faiss_amd_idx_t* labels, faiss_amd_idx_t k);
int faiss_amd_Index_reset(FaissAmdIndex* index);
int faiss_amd_Index_reconstruct(const FaissAmdIndex* index, faiss_amd_idx_t key, float* recons);
int faiss_amd_Index_reconstruct_n(const FaissAmdIndex* index, faiss_amd_idx_t i0, faiss_amd_idx_t ni,
                                  float* recons);

/* ---- IVF surface: c_api/IndexIVF_c.h:31-62 (nlist, nprobe, quantizer, list sizes/ids),
 *      faiss/gpu/GpuIndexIVF.h:97-110 (getListLength / getListVectorData / getListIndices) */
int faiss_amd_IndexIVF_nlist(const FaissAmdIndex* index, int* p_nlist);
int faiss_amd_IndexIVF_nprobe(const FaissAmdIndex* index, int* p_nprobe);
int faiss_amd_IndexIVF_set_nprobe(FaissAmdIndex* index, int nprobe);
int faiss_amd_IndexIVF_get_list_size(const FaissAmdIndex* index, faiss_amd_idx_t list_no, size_t* p_size);
int faiss_amd_IndexIVF_get_list_ids(const FaissAmdIndex* index, faiss_amd_idx_t list_no,
                                    faiss_amd_idx_t* ids_out);
/* payload of one list in the reference's CPU layout: d floats (IVFFlat), M bytes (IVFPQ) or sq.code_size bytes
 * (IVF scalar quantizer) per entry; out must hold list_size * code_size bytes */
int faiss_amd_IndexIVF_get_list_codes(const FaissAmdIndex* index, faiss_amd_idx_t list_no, uint8_t* out);
int faiss_amd_IndexIVF_code_size(const FaissAmdIndex* index, size_t* p_code_size);
/* coarse centroids out: nlist x d floats (quantizer->reconstruct_n) */
int faiss_amd_IndexIVF_get_centroids(const FaissAmdIndex* index, float* centroids_out);
/* k-means iterations / seed of train(): ClusteringParameters niter, seed (faiss/Clustering.h:24-63) */
int faiss_amd_IndexIVF_set_clustering(FaissAmdIndex* index, int niter, int seed);

/* copyFrom(IndexIVFFlat / IndexIVFPQ / IndexIVFScalarQuantizer) decomposed into plain arrays
 * (faiss/gpu/GpuIndexIVFFlat.cu copyFrom, faiss/gpu/GpuIndexIVFPQ.cu:98-168, GpuIndexIVFScalarQuantizer.cu:96-140):
 *   centroids   nlist x d floats                 (index->quantizer, an IndexFlat)
 *   pq          M x 256 x (d/M) floats           (index->pq.centroids)
 *   list_sizes  nlist counts, codes/ids = the lists' payloads concatenated in list order
 *               (invlists->get_codes / get_ids) */
int faiss_amd_IndexIVF_copy_centroids(FaissAmdIndex* index, const float* centroids);
int faiss_amd_IndexIVFPQ_copy_pq_centroids(FaissAmdIndex* index, const float* pq);
int faiss_amd_IndexIVFPQ_get_pq_centroids(const FaissAmdIndex* index, float* pq_out);
int faiss_amd_IndexIVF_copy_lists(FaissAmdIndex* index, const uint32_t* list_sizes, const uint8_t* codes,
                                  const faiss_amd_idx_t* ids);

/* ---- k-means: faiss::Clustering::train with a GPU flat index as assignment engine
 *      (c_api/Clustering_c.h:100-116 faiss_kmeans_clustering; faiss/Clustering.cpp:255-357).
 *      centroids_out: k x d floats; obj_out (nullable): niter floats, objective per iteration */
int faiss_amd_kmeans_clustering(FaissAmdGpuResources* res, int d, faiss_amd_idx_t n, int k, const float* x,
                                int niter, int seed, float* centroids_out, float* obj_out);

/* faiss::ClusteringParameters (faiss/Clustering.h:27-60; c_api/Clustering_c.h:22-58 FaissClusteringParameters), the
 * fields the k-means loop here honours.  nredo: runs from different random starts, best final objective wins;
 * spherical: centroids L2-normalised after every update (inner-product clustering); int_centroids: rounded to integers;
 * frozen_centroids: the initial centroids handed to train_ex are never updated; update_index is accepted (flat
 * assignment engines have nothing to re-train). */
typedef struct FaissAmdClusteringParameters {
    int niter, nredo, verbose, spherical, int_centroids, update_index, frozen_centroids;
    int min_points_per_centroid, max_points_per_centroid, seed;
} FaissAmdClusteringParameters;
/* c_api/Clustering_c.h:61 faiss_ClusteringParameters_init: the reference's defaults (niter 25, nredo 1, 39 / 256, seed 1234) */
void faiss_amd_ClusteringParameters_init(FaissAmdClusteringParameters* params);
/* Clustering::train with all parameters; init_centroids (nullable): n_init x d floats placed first (Clustering::centroids
 * set before train, faiss/Clustering.cpp:330-345).  obj_out (nullable): niter floats of the winning run. */
int faiss_amd_Clustering_train_ex(FaissAmdIndex* index, faiss_amd_idx_t n, const float* x, int k,
                                  const FaissAmdClusteringParameters* params, const float* init_centroids, int n_init,
                                  float* centroids_out, float* obj_out, int* on_device_out);
/* GpuIndexIVF::cp (faiss/gpu/GpuIndexIVF.h): the clustering parameters train() uses for the coarse quantizer */
int faiss_amd_IndexIVF_set_clustering_params(FaissAmdIndex* index, const FaissAmdClusteringParameters* params);

/* faiss::Clustering::train(n, x, index) (faiss/Clustering.h:147-160; c_api/Clustering_c.h:118-122 faiss_Clustering_train):
 * k-means with `index` (dimension d, empty or not -- it is reset) as the assignment engine; on return the index holds
 * the k centroids.  With a GpuIndexFlat the whole loop runs on the device (x host or device); any other index of this
 * library is driven through add / search with host data.  on_device_out (nullable): which of the two ran. */
int faiss_amd_Clustering_train(FaissAmdIndex* index, faiss_amd_idx_t n, const float* x, int k, int niter, int seed,
                               float* centroids_out, float* obj_out, int* on_device_out);

/* ---- shard merge: faiss::merge_knn_results (faiss/utils/Heap.h merge_knn_results,
 *      faiss/utils/Heap.cpp:166-240); all_d/all_i are [nshard][n][k]; base (nullable) is added
 *      to each shard's labels (IndexShards successive_ids, faiss/IndexShards.cpp:214-237) */
int faiss_amd_merge_knn_results(FaissAmdMetricType metric, faiss_amd_idx_t n, faiss_amd_idx_t k, int nshard,
                                const float* all_d, const faiss_amd_idx_t* all_i,
                                const faiss_amd_idx_t* base, float* distances, faiss_amd_idx_t* labels);

/* same merge on the device (all_d/all_i/distances/labels are device pointers on res's device,
 * base is a host array): what the reference does only on the host; used by the
 * one-process-per-GPU sharded search after the per-shard results were gathered over RCCL */
int faiss_amd_merge_knn_results_device(FaissAmdGpuResources* res, FaissAmdMetricType metric, faiss_amd_idx_t n,
                                       faiss_amd_idx_t k, int nshard, const float* all_d,
                                       const faiss_amd_idx_t* all_i, const faiss_amd_idx_t* base,
                                       float* distances, faiss_amd_idx_t* labels);

/* ---- measurement hooks (no reference equivalent; the reference brackets searches with
 *      CpuTimer/KernelTimer, faiss/gpu/utils/Timer.h).  Per-kernel HIP-event timing on the
 *      resources' stream: enable, run searches, then read total ms / launch count by kernel
 *      name ("flat_scan_kernel", "select_k_kernel", "ivfflat_scan_kernel", "ivfpq_scan_kernel") */
int faiss_amd_profile_enable(FaissAmdGpuResources* res, int on);
int faiss_amd_profile_reset(FaissAmdGpuResources* res);
int faiss_amd_profile_get(FaissAmdGpuResources* res, const char* kernel_name, double* total_ms, long* launches);

/* ---- test hooks */
/* full distance matrix [n][ntotal] produced by the fused MFMA kernel (bfKnn-style all-pairs,
 * faiss/gpu/GpuDistance.h:32-152 with outDistances only) */
int faiss_amd_GpuIndexFlat_pairwise_distances(const FaissAmdIndex* index, faiss_amd_idx_t n, const float* x,
                                              float* out);
/* route search() through the scalar cross-check kernel (identical arithmetic, no MFMA) */
int faiss_amd_GpuIndexFlat_set_use_simple_kernel(FaissAmdIndex* index, int on);
/* fp16 MFMA candidate filter + exact fp32 re-rank for GpuIndexFlat (results bit-identical to the fp32
 * scan; on by default for databases of at least min_rows rows; min_rows < 0 keeps the current value) */
int faiss_amd_GpuIndexFlat_set_use_filter_kernel(FaissAmdIndex* index, int on, faiss_amd_idx_t min_rows);
/* did the last search tile go through the filter, and how many of its queries were re-run exactly */
int faiss_amd_GpuIndexFlat_filter_stats(const FaissAmdIndex* index, int* used_filter, int* overflow_queries);
/* ---- the rest of the faiss::Index surface a coarse quantizer / shard wrapper uses
 *      reconstruct_batch (faiss/Index.h:297-307; GpuIndexFlat.cu:294-320), compute_residual[_n]
 *      (faiss/Index.h:363-383; GpuIndexFlat.cu:323-361, impl/VectorResidual.cu:26-97): residual = x - stored[key],
 *      a key of -1 gives a row of NaNs.  Pointers host or device. */
int faiss_amd_Index_reconstruct_batch(const FaissAmdIndex* index, faiss_amd_idx_t n, const faiss_amd_idx_t* keys,
                                      float* recons);
int faiss_amd_Index_compute_residual(const FaissAmdIndex* index, const float* x, float* residual, faiss_amd_idx_t key);
int faiss_amd_Index_compute_residual_n(const FaissAmdIndex* index, faiss_amd_idx_t n, const float* xs, float* residuals,
                                       const faiss_amd_idx_t* keys);

/* ---- memory management and quantizer refresh of the reference's IVF classes (faiss/gpu/GpuIndexIVFFlat.h:64-85,
 *      GpuIndexIVFPQ.h:98-126, GpuIndexIVFScalarQuantizer.h:66-87, GpuIndexIVF.h:79):
 *      reserveMemory: room for numVecs vectors up front (no re-allocation of the list arena by the adds that follow);
 *      reclaimMemory: give back slack, holes and add-path scratch, *p_bytes = device bytes released (may be NULL);
 *      updateQuantizer: call after changing the coarse centroids from outside.
 *      IVFPQ: precomputed codes are the per-vector term that is always on here for L2 -- the setter records the request,
 *      the getter reports what is in force (1 for L2, 0 for inner product: GpuIndexIVFPQ.cu:228-241 forces it off there);
 *      getTableInfo adds the request itself and the lookup-table precision in force (always fp32: 0). */
int faiss_amd_GpuIndexIVF_reserveMemory(FaissAmdIndex* index, size_t num_vecs);
int faiss_amd_GpuIndexIVF_reclaimMemory(FaissAmdIndex* index, size_t* p_bytes);
int faiss_amd_GpuIndexIVF_updateQuantizer(FaissAmdIndex* index);
int faiss_amd_GpuIndexIVFPQ_setPrecomputedCodes(FaissAmdIndex* index, int enable);
/* getters of GpuIndexIVFPQ.h:104-113: any of the outputs may be NULL */
int faiss_amd_GpuIndexIVFPQ_getInfo(const FaissAmdIndex* index, int* precomputed_codes, int* num_sub_quantizers,
                                    int* bits_per_code, int* centroids_per_sub_quantizer);
int faiss_amd_GpuIndexIVFPQ_getTableInfo(const FaissAmdIndex* index, int* precomputed_in_force, int* precomputed_requested,
                                         int* float16_tables_in_force);

/* ---- GpuIndexIVF::add_core (faiss/gpu/GpuIndexIVF.h:84-95, GpuIndexIVF.cu:321-356; what contrib/ivf_tools.py
 *      add_preassigned drives): add n vectors whose inverted list is given by the caller (precomputed_idx [n], host or
 *      device; entries outside [0, nlist) leave their vector out).  xids may be NULL (sequential ids from ntotal). */
int faiss_amd_GpuIndexIVF_add_core(FaissAmdIndex* index, faiss_amd_idx_t n, const float* x, const faiss_amd_idx_t* xids,
                                   const faiss_amd_idx_t* precomputed_idx);

/* ---- GpuIndexIVF::search_preassigned (faiss/gpu/GpuIndexIVF.h:112-122, GpuIndexIVF.cu:408-488; the entry
 *      IndexShardsIVF and CPU-quantizer hybrids call): assign and centroid_dis are [n][nprobe] (nprobe = the
 *      index's current value), host or device, -1 = no list.  With the arrays the index's own quantizer returns
 *      (faiss_amd_IndexIVF_quantizer_search = index.quantizer->search) the result is bit-identical to search(). */
int faiss_amd_GpuIndexIVF_search_preassigned(const FaissAmdIndex* index, faiss_amd_idx_t n, const float* x,
                                             faiss_amd_idx_t k, const faiss_amd_idx_t* assign, const float* centroid_dis,
                                             float* distances, faiss_amd_idx_t* labels);
int faiss_amd_IndexIVF_quantizer_search(const FaissAmdIndex* index, faiss_amd_idx_t n, const float* x, faiss_amd_idx_t k,
                                        float* distances, faiss_amd_idx_t* labels);

/* ---- faiss::gpu::bfKnn (faiss/gpu/GpuDistance.h:32-152), float32 row-major subset: brute-force k-NN of
 *      `queries` [num_queries][dims] in `vectors` [num_vectors][dims], both host or device, never written.
 *      Same kernels, tie rule and bits as GpuIndexFlat::search.  L2 distances are squared. */
int faiss_amd_bfKnn(FaissAmdGpuResources* res, FaissAmdMetricType metric, const float* vectors,
                    faiss_amd_idx_t num_vectors, const float* queries, faiss_amd_idx_t num_queries, int dims,
                    faiss_amd_idx_t k, float* out_distances, faiss_amd_idx_t* out_indices);

/* ---- the full operator surface of faiss::gpu::bfKnn: GpuDistanceParams field for field (faiss/gpu/GpuDistance.h:32-152).
 *      vectorType / queryType: 1 = F32, 2 = F16, 3 = BF16 (DistanceDataType); row or column major; outIndicesType:
 *      1 = int64, 2 = int32; k = -1 returns all pairwise distances [numQueries][numVectors] in outDistances; vectorNorms
 *      is accepted and unused; pointers host or device.  With F16 vectors AND queries the search runs on an
 *      fp16-storage index (those exact values, half the bytes).  faiss_amd_bfKnn_tiling = faiss::gpu::bfKnn_tiling
 *      (GpuDistance.cu:430-570): 0 = "must fit", else at most that many bytes of vectors / of queries + results on the
 *      device at a time (row-major CPU inputs, k > 0); the tiles' partial results are merged under (distance, id). */
typedef struct FaissAmdGpuDistanceParams {
    int metric;
    float metricArg;
    int k, dims;
    const void* vectors;
    int vectorType, vectorsRowMajor;
    faiss_amd_idx_t numVectors;
    const float* vectorNorms;
    const void* queries;
    int queryType, queriesRowMajor;
    faiss_amd_idx_t numQueries;
    float* outDistances;
    int ignoreOutDistances, outIndicesType;
    void* outIndices;
    int device;
} FaissAmdGpuDistanceParams;
int faiss_amd_bfKnn_params(FaissAmdGpuResources* res, const FaissAmdGpuDistanceParams* args);
int faiss_amd_bfKnn_tiling(FaissAmdGpuResources* res, const FaissAmdGpuDistanceParams* args, size_t vectorsMemoryLimit,
                           size_t queriesMemoryLimit);

/* ---- faiss::InterruptCallback (faiss/impl/AuxIndexStructures.h:138-165): a process-wide hook polled between the tiles of
 *      long-running calls (as faiss/gpu/impl/Distance.cu:245,266 does); a non-zero return makes the call fail with
 *      "computation interrupted".  NULL removes it. */
typedef int (*faiss_amd_interrupt_fn)(void* user);
int faiss_amd_set_interrupt_callback(faiss_amd_interrupt_fn fn, void* user);
/* ---- faiss::gpu::GpuParameterSpace::set_index_parameter (faiss/gpu/GpuAutoTune.cpp:81-114): "nprobe" on IVF indexes,
 *      recursively through IndexReplicas / IndexShards ("use_precomputed_table" is accepted on IVFPQ: always on here) */
int faiss_amd_GpuParameterSpace_set_index_parameter(FaissAmdIndex* index, const char* name, double value);

/* ---- SearchParametersIVF (faiss/IndexIVF.h:70-80): per-call override of nprobe, as GpuIndexIVF::search honours it
 *      through getCurrentNProbe_ (faiss/gpu/GpuIndexIVF.cu:358-381).  nprobe <= 0 keeps the index's own value. */
typedef struct FaissAmdSearchParametersIVF {
    int nprobe;
} FaissAmdSearchParametersIVF;
int faiss_amd_GpuIndexIVF_search_with_params(const FaissAmdIndex* index, faiss_amd_idx_t n, const float* x,
                                             faiss_amd_idx_t k, const FaissAmdSearchParametersIVF* params,
                                             float* distances, faiss_amd_idx_t* labels);
/* ---- faiss::IDSelector (faiss/impl/IDSelector.h:21-215; C API of the reference: c_api/impl/AuxIndexStructures_c.h:51-116)
 *      and faiss::SearchParameters / SearchParametersIVF (c_api/Index_c.h:41-46, c_api/IndexIVF_c.h:22-37).
 *      A selector restricts a search to the vectors whose LABEL it admits: row numbers for GpuIndexFlat, the stored user
 *      ids for the IVF indexes; the result is bit for bit what an index holding only those vectors returns, -1 / the
 *      neutral distance fill up when fewer than k qualify.  The reference's CPU indexes honour `sel`, its GPU indexes
 *      accept and ignore it outside cuVS (faiss/gpu/GpuIndexFlat.cu:232, GpuIndexIVF.cu:402); here it runs on the device.
 *      Range: imin <= id < imax.  Batch / Array: the listed ids (copied).  Bitmap: id / 8 < n and bit id % 8 of
 *      bitmap[id / 8] (copied).  Not / And / Or / XOr combine selectors that must outlive the combination (not owned). */
typedef struct FaissAmdIDSelector_H FaissAmdIDSelector;
int faiss_amd_IDSelectorAll_new(FaissAmdIDSelector** p_sel);
int faiss_amd_IDSelectorRange_new(FaissAmdIDSelector** p_sel, faiss_amd_idx_t imin, faiss_amd_idx_t imax);
int faiss_amd_IDSelectorBatch_new(FaissAmdIDSelector** p_sel, size_t n, const faiss_amd_idx_t* ids);
int faiss_amd_IDSelectorArray_new(FaissAmdIDSelector** p_sel, size_t n, const faiss_amd_idx_t* ids);
int faiss_amd_IDSelectorBitmap_new(FaissAmdIDSelector** p_sel, size_t n, const uint8_t* bitmap);
int faiss_amd_IDSelectorNot_new(FaissAmdIDSelector** p_sel, const FaissAmdIDSelector* sel);
int faiss_amd_IDSelectorAnd_new(FaissAmdIDSelector** p_sel, const FaissAmdIDSelector* lhs, const FaissAmdIDSelector* rhs);
int faiss_amd_IDSelectorOr_new(FaissAmdIDSelector** p_sel, const FaissAmdIDSelector* lhs, const FaissAmdIDSelector* rhs);
int faiss_amd_IDSelectorXOr_new(FaissAmdIDSelector** p_sel, const FaissAmdIDSelector* lhs, const FaissAmdIDSelector* rhs);
/* 1 / 0 (host-side evaluation; the reference's faiss_IDSelector_is_member) */
int faiss_amd_IDSelector_is_member(const FaissAmdIDSelector* sel, faiss_amd_idx_t id);
void faiss_amd_IDSelector_free(FaissAmdIDSelector* sel);

typedef struct FaissAmdSearchParameters_H FaissAmdSearchParameters;
/* sel may be NULL (no restriction); it is not owned and must outlive the parameters */
int faiss_amd_SearchParameters_new(FaissAmdSearchParameters** p_sp, const FaissAmdIDSelector* sel);
/* nprobe 0 keeps the index's own value; max_codes must be 0 (the GPU indexes of the reference refuse it too) */
int faiss_amd_SearchParametersIVF_new_with(FaissAmdSearchParameters** p_sp, const FaissAmdIDSelector* sel, size_t nprobe,
                                           size_t max_codes);
void faiss_amd_SearchParameters_free(FaissAmdSearchParameters* sp);
/* faiss_Index_search_with_params (c_api/Index_c.h:127-135): any index behind this ABI; params may be NULL */
int faiss_amd_Index_search_with_params(const FaissAmdIndex* index, faiss_amd_idx_t n, const float* x, faiss_amd_idx_t k,
                                       const FaissAmdSearchParameters* params, float* distances, faiss_amd_idx_t* labels);
/* search_preassigned with search parameters (IndexIVF::search_preassigned takes them as its `params` argument) */
int faiss_amd_GpuIndexIVF_search_preassigned_with_params(const FaissAmdIndex* index, faiss_amd_idx_t n, const float* x,
                                                         faiss_amd_idx_t k, const faiss_amd_idx_t* assign,
                                                         const float* centroid_dis, const FaissAmdSearchParameters* params,
                                                         float* distances, faiss_amd_idx_t* labels);

/* vectors actually stored in the inverted lists: ntotal counts every vector add() was given (NaN rows are not
 * stored but counted, faiss/gpu/GpuIndexIVF.cu:293-298) */
int faiss_amd_GpuIndexIVF_stored_vectors(const FaissAmdIndex* index, faiss_amd_idx_t* p_stored);
/* list arena in rows: handed out (lists + slack + holes), holes left by relocated lists, allocated */
int faiss_amd_GpuIndexIVF_arena_stats(const FaissAmdIndex* index, int64_t* used_rows, int64_t* hole_rows,
                                      int64_t* allocated_rows);

/* IVF search through the unfused path (every distance as a key in HBM + select kernel) instead
 * of the fused LDS-resident scan; results are identical, the switch exists for cross-checks */
int faiss_amd_GpuIndexIVF_set_use_fused_scan(FaissAmdIndex* index, int on);

/* Which scan serves search() / search_preassigned() of an IVF index (no reference counterpart: the reference has one
 * scan per index type, faiss/gpu/impl/IVFInterleaved.cuh:33-224, PQScanMultiPassNoPrecomputed-inl.cuh:173-270, both
 * query-major).  0 = automatic: GpuIndexIVF::list_major_rule -- IVFFlat (d <= 512) and IVFPQ (d <= 128):
 * a cost model fitted to side-by-side timings picks the faster scan (query-major: the bytes it streams; list-major: fixed
 * launches + per-query work + two sweeps over the lists the batch touches; the results are the same bits either way);
 * the scalar quantizer (d <= 128) takes the list-major scan from 2048 queries on
 * (every list is read once per group of the queries probing it); 1 = query-major always; 2 = list-major always (an
 * error where unsupported); 3 = list-major on the f32 matrix pipe (round 3's scan, faiss_amd/csrc/ivf_listmajor.hip,
 * d <= 128, no IDSelector).
 * IVFFlat / IVFPQ (round 4): the list-major scan runs behind an f16 MFMA filter with a rigorous error band and
 * re-derives the survivors with the arithmetic of the query-major scan (faiss_amd/csrc/ivf_lm_filter.hip): a query
 * returns THE SAME BITS whatever the batch size, the shard / replica split or the paging of the call.  The scalar
 * quantizer's list-major scan and mode 3 sum in their own order (f32 MFMA chains).  Every scan is bit-exact against its
 * restatement in oracle/faiss_oracle.c (orc_ivf_search_ex / orc_ivfsq_search_ex, arith = last_scan_arith), all within
 * the 1e-4 relative tolerance of the reference.  scan_info: the mode set, the scan the last search used (1 query-major /
 * 2 list-major), and how many queries so far had to be redone (candidate segment overflow, fp16 range).
 * last_scan_arith: 0 = query-major arithmetic, 1 = f32 list-major arithmetic. */
int faiss_amd_GpuIndexIVF_set_scan_mode(FaissAmdIndex* index, int mode);
/* Memory of the filter path (the reference's GpuIndexIVFConfig / GpuIndexConfig::memorySpace are where a caller states its
 * memory policy, faiss/gpu/GpuIndex.h:31-47): the sweeps of the automatic mode keep their own copy of the lists -- IVFFlat an
 * fp16 shadow (+ 2 d bytes per row), IVFPQ the codes again in operand order (+ M bytes per row) -- built at the first
 * list-major search and kept up to date by add().  set_use_filter_shadow(0): never build it under mode 0 (the query-major
 * scan serves every call, same bits).  A build that runs out of device memory falls back the same way by itself; only an
 * explicit mode 2 reports the failure.  resident_bytes: device bytes of the lists and of the copies (either may be NULL). */
int faiss_amd_GpuIndexIVF_set_use_filter_shadow(FaissAmdIndex* index, int on);
int faiss_amd_GpuIndexIVF_resident_bytes(const FaissAmdIndex* index, size_t* p_lists, size_t* p_shadow);
int faiss_amd_GpuIndexIVF_scan_info(const FaissAmdIndex* index, int* mode, int* last_mode, int64_t* overflow_queries);
int faiss_amd_GpuIndexIVF_last_scan_arith(const FaissAmdIndex* index, int* p_arith);
/* *p_output = 1 when mode 0 sends a batch of n queries with this nprobe and k through the list-major scan */
int faiss_amd_GpuIndexIVF_list_major_rule(const FaissAmdIndex* index, int64_t n, int nprobe, int64_t k, int* p_output);

#ifdef __cplusplus
}
#endif
#endif
