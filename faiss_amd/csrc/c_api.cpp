// faiss_amd/csrc/c_api.cpp -- extern "C" boundary (include/faiss_amd_c.h).
// Error convention mirrors the reference C API (c_api/macros_impl.h:22-56, c_api/error_impl.cpp).
#include "../../include/faiss_amd_c.h"
#include "faiss_amd_internal.h"
#include <cstring>
#include <exception>
#include <memory>
#include <string>
#include "index.h"
#include "kernels.h"

using namespace faiss_amd;

static thread_local std::string g_last_error;

#define FA_TRY try {
#define FA_CATCH                                   \
    }                                              \
    catch (FaissAmdException & e) {                \
        g_last_error = e.what();                   \
        return -2;                                 \
    }                                              \
    catch (std::exception & e) {                   \
        g_last_error = e.what();                   \
        return -4;                                 \
    }                                              \
    catch (...) {                                  \
        g_last_error = "Unknown error";            \
        return -1;                                 \
    }                                              \
    return 0;

// same, falling through to the code behind it on success
#define FA_CATCH_RC                                \
    }                                              \
    catch (FaissAmdException & e) {                \
        g_last_error = e.what();                   \
        return -2;                                 \
    }                                              \
    catch (std::exception & e) {                   \
        g_last_error = e.what();                   \
        return -4;                                 \
    }                                              \
    catch (...) {                                  \
        g_last_error = "Unknown error";            \
        return -1;                                 \
    }

struct FaissAmdGpuResources_H {
    std::shared_ptr<GpuResources> res;
};
struct FaissAmdIndex_H {
    Index* index;
    std::shared_ptr<GpuResources> res; // keeps the resources alive as long as the index
};

static Index* I(FaissAmdIndex* h) {
    if (!h || !h->index) FA_THROW_MSG("null index handle");
    return h->index;
}
static const Index* I(const FaissAmdIndex* h) {
    if (!h || !h->index) FA_THROW_MSG("null index handle");
    return h->index;
}
template <typename T>
static T* as(FaissAmdIndex* h, const char* what) {
    T* t = dynamic_cast<T*>(I(h));
    if (!t) FA_THROW_MSG(std::string("index is not a ") + what);
    return t;
}
template <typename T>
static const T* as(const FaissAmdIndex* h, const char* what) {
    const T* t = dynamic_cast<const T*>(I(h));
    if (!t) FA_THROW_MSG(std::string("index is not a ") + what);
    return t;
}
static std::shared_ptr<GpuResources> R(FaissAmdGpuResources* r) {
    if (!r || !r->res) FA_THROW_MSG("null resources handle");
    return r->res;
}

extern "C" {

const char* faiss_amd_get_last_error(void) {
    return g_last_error.c_str();
}

int faiss_amd_metric_supported(int index_kind, int metric, int* p_output) {
    FA_TRY
    FA_THROW_IF_NOT_MSG(p_output && (index_kind == 0 || index_kind == 1), "bad argument");
    *p_output = metric_supported(index_kind, metric) ? 1 : 0;
    FA_CATCH
}

int faiss_amd_get_num_gpus(int* p_output) {
    FA_TRY
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *p_output = n;
    FA_CATCH
}

int faiss_amd_StandardGpuResources_new(FaissAmdGpuResources** p_res, int device) {
    FA_TRY
    auto* h = new FaissAmdGpuResources_H;
    try {
        h->res = std::make_shared<GpuResources>(device);
    } catch (...) {
        delete h;
        throw;
    }
    *p_res = h;
    FA_CATCH
}
void faiss_amd_StandardGpuResources_free(FaissAmdGpuResources* res) {
    delete res;
}
int faiss_amd_StandardGpuResources_sync(FaissAmdGpuResources* res) {
    FA_TRY
    R(res)->set_device();
    R(res)->sync();
    FA_CATCH
}
int faiss_amd_StandardGpuResources_getDefaultStream(FaissAmdGpuResources* res, void** p_stream) {
    FA_TRY
    *p_stream = (void*)R(res)->stream;
    FA_CATCH
}
int faiss_amd_StandardGpuResources_setPagedSearch(FaissAmdGpuResources* res, size_t min_bytes, int64_t page_queries) {
    FA_TRY
    FA_THROW_IF_NOT_MSG(page_queries >= 0, "negative page size");
    R(res)->paged_min_bytes = min_bytes;
    R(res)->paged_page_queries = page_queries;
    FA_CATCH
}
int faiss_amd_StandardGpuResources_getPagedSearchCount(FaissAmdGpuResources* res, int64_t* p_count) {
    FA_TRY
    *p_count = R(res)->paged_searches;
    FA_CATCH
}
int faiss_amd_StandardGpuResources_setDefaultStream(FaissAmdGpuResources* res, void* stream) {
    FA_TRY
    R(res)->set_default_stream((hipStream_t)stream);
    FA_CATCH
}
int faiss_amd_StandardGpuResources_setTempMemory(FaissAmdGpuResources* res, size_t bytes) {
    FA_TRY
    FA_THROW_IF_NOT_MSG(bytes >= ((size_t)64 << 20), "temp memory must be at least 64 MiB");
    R(res)->temp_budget_bytes = bytes;
    FA_CATCH
}

int faiss_amd_GpuIndexFlat_new(FaissAmdIndex** p_index, FaissAmdGpuResources* res, int d,
                               FaissAmdMetricType metric) {
    FA_TRY
    auto r = R(res);
    auto* h = new FaissAmdIndex_H{nullptr, r};
    try {
        h->index = new GpuIndexFlat(r, d, (int)metric);
    } catch (...) {
        delete h;
        throw;
    }
    *p_index = h;
    FA_CATCH
}
int faiss_amd_GpuIndexIVFFlat_new(FaissAmdIndex** p_index, FaissAmdGpuResources* res, int d, int nlist,
                                  FaissAmdMetricType metric) {
    FA_TRY
    auto r = R(res);
    auto* h = new FaissAmdIndex_H{nullptr, r};
    try {
        h->index = new GpuIndexIVFFlat(r, d, nlist, (int)metric);
    } catch (...) {
        delete h;
        throw;
    }
    *p_index = h;
    FA_CATCH
}
int faiss_amd_GpuIndexIVFPQ_new(FaissAmdIndex** p_index, FaissAmdGpuResources* res, int d, int nlist, int M,
                                int nbits, FaissAmdMetricType metric) {
    FA_TRY
    auto r = R(res);
    auto* h = new FaissAmdIndex_H{nullptr, r};
    try {
        h->index = new GpuIndexIVFPQ(r, d, nlist, M, nbits, (int)metric);
    } catch (...) {
        delete h;
        throw;
    }
    *p_index = h;
    FA_CATCH
}
static void check_common_config(const std::shared_ptr<GpuResources>& r, int device, int memorySpace) {
    FA_THROW_IF_NOT_MSG(device == -1 || device == r->device, "config.device differs from the device of the resources");
    FA_THROW_IF_NOT_MSG(memorySpace == 0, "only MemorySpace::Device is supported");
}
static void check_ivf_config(const std::shared_ptr<GpuResources>& r, const FaissAmdGpuIndexIVFConfig& c) {
    check_common_config(r, c.device, c.memorySpace);
    FA_THROW_IF_NOT_MSG(c.indicesOptions >= 0 && c.indicesOptions <= 3, "indicesOptions: not a faiss::gpu::IndicesOptions value");
}
// coarse quantizer handed in by the caller (null: the index makes its own): one of our flat indexes
static GpuIndexFlat* coarse_of(FaissAmdIndex* q) {
    if (!q) return nullptr;
    return as<GpuIndexFlat>(q, "GpuIndexFlat (the coarse quantizer of a GpuIndexIVF must be a flat index of this library; any other "
                               "faiss::Index is served on the reference side through search_preassigned, INTEGRATION.md)");
}
int faiss_amd_GpuIndexFlat_new_with_config(FaissAmdIndex** p_index, FaissAmdGpuResources* res, int d,
                                           FaissAmdMetricType metric, const FaissAmdGpuIndexFlatConfig* config) {
    FA_TRY
    auto r = R(res);
    bool f16 = false;
    if (config) {
        check_common_config(r, config->device, config->memorySpace);
        f16 = config->useFloat16 != 0;
    }
    auto* h = new FaissAmdIndex_H{nullptr, r};
    try {
        h->index = new GpuIndexFlat(r, d, (int)metric, f16);
    } catch (...) {
        delete h;
        throw;
    }
    *p_index = h;
    FA_CATCH
}
int faiss_amd_GpuIndexIVFFlat_new_with_quantizer(FaissAmdIndex** p_index, FaissAmdGpuResources* res, FaissAmdIndex* coarse_quantizer,
                                                 int d, int nlist, FaissAmdMetricType metric, const FaissAmdGpuIndexIVFConfig* config) {
    FA_TRY
    auto r = R(res);
    if (config) check_ivf_config(r, *config);
    auto* h = new FaissAmdIndex_H{nullptr, r};
    try {
        h->index = new GpuIndexIVFFlat(r, d, nlist, (int)metric, coarse_of(coarse_quantizer), config && config->flat_useFloat16,
                                       config ? config->indicesOptions : 3);
    } catch (...) {
        delete h;
        throw;
    }
    *p_index = h;
    FA_CATCH
}
int faiss_amd_GpuIndexIVFPQ_new_with_quantizer(FaissAmdIndex** p_index, FaissAmdGpuResources* res, FaissAmdIndex* coarse_quantizer,
                                               int d, int nlist, int M, int nbits, FaissAmdMetricType metric,
                                               const FaissAmdGpuIndexIVFPQConfig* config) {
    FA_TRY
    auto r = R(res);
    if (config) check_ivf_config(r, config->ivf);
    auto* h = new FaissAmdIndex_H{nullptr, r};
    try {
        h->index = new GpuIndexIVFPQ(r, d, nlist, M, nbits, (int)metric, coarse_of(coarse_quantizer),
                                     config && config->ivf.flat_useFloat16, config ? config->ivf.indicesOptions : 3);
    } catch (...) {
        delete h;
        throw;
    }
    *p_index = h;
    FA_CATCH
}
int faiss_amd_GpuIndexIVFScalarQuantizer_new_with_quantizer(FaissAmdIndex** p_index, FaissAmdGpuResources* res,
                                                            FaissAmdIndex* coarse_quantizer, int d, int nlist, int qtype,
                                                            FaissAmdMetricType metric, int encode_residual,
                                                            const FaissAmdGpuIndexIVFConfig* config) {
    FA_TRY
    auto r = R(res);
    if (config) check_ivf_config(r, *config);
    auto* h = new FaissAmdIndex_H{nullptr, r};
    try {
        h->index = new GpuIndexIVFScalarQuantizer(r, d, nlist, qtype, (int)metric, encode_residual != 0, coarse_of(coarse_quantizer),
                                                  config && config->flat_useFloat16, config ? config->indicesOptions : 3);
    } catch (...) {
        delete h;
        throw;
    }
    *p_index = h;
    FA_CATCH
}
int faiss_amd_GpuIndexIVFFlat_new_with_config(FaissAmdIndex** p_index, FaissAmdGpuResources* res, int d, int nlist,
                                              FaissAmdMetricType metric, const FaissAmdGpuIndexIVFConfig* config) {
    return faiss_amd_GpuIndexIVFFlat_new_with_quantizer(p_index, res, nullptr, d, nlist, metric, config);
}
int faiss_amd_GpuIndexIVFPQ_new_with_config(FaissAmdIndex** p_index, FaissAmdGpuResources* res, int d, int nlist, int M,
                                            int nbits, FaissAmdMetricType metric,
                                            const FaissAmdGpuIndexIVFPQConfig* config) {
    return faiss_amd_GpuIndexIVFPQ_new_with_quantizer(p_index, res, nullptr, d, nlist, M, nbits, metric, config);
}
int faiss_amd_GpuIndexIVF_quantizer_info(const FaissAmdIndex* index, int* own_fields, int* use_float16, int* indices_options) {
    FA_TRY
    auto* ivf = as<GpuIndexIVF>(index, "GpuIndexIVF");
    if (own_fields) *own_fields = ivf->own_fields ? 1 : 0;
    if (use_float16) *use_float16 = ivf->quantizer->getUseFloat16() ? 1 : 0;
    if (indices_options) *indices_options = ivf->indices_options;
    FA_CATCH
}
int faiss_amd_StandardGpuResources_getMemoryInfo(FaissAmdGpuResources* res, size_t* allocations, size_t* bytes, size_t* peak_bytes,
                                                 size_t* temp_memory, size_t* device_free, size_t* device_total) {
    FA_TRY
    auto r = R(res);
    r->set_device();
    device_memory_info(r->device, allocations, bytes, peak_bytes);
    if (temp_memory) *temp_memory = r->temp_budget_bytes;
    size_t fr = 0, tot = 0;
    HIP_CHECK(hipMemGetInfo(&fr, &tot));
    if (device_free) *device_free = fr;
    if (device_total) *device_total = tot;
    FA_CATCH
}
int faiss_amd_StandardGpuResources_setLogMemoryAllocations(FaissAmdGpuResources* res, int enable) {
    FA_TRY
    set_log_memory_allocations(R(res)->device, enable != 0);
    FA_CATCH
}
int faiss_amd_GpuIndexIVFScalarQuantizer_new(FaissAmdIndex** p_index, FaissAmdGpuResources* res, int d, int nlist,
                                             int qtype, FaissAmdMetricType metric, int encode_residual) {
    FA_TRY
    auto r = R(res);
    auto* h = new FaissAmdIndex_H{nullptr, r};
    try {
        h->index = new GpuIndexIVFScalarQuantizer(r, d, nlist, qtype, (int)metric, encode_residual != 0);
    } catch (...) {
        delete h;
        throw;
    }
    *p_index = h;
    FA_CATCH
}
int faiss_amd_IndexIVFSQ_info(const FaissAmdIndex* index, int* qtype, int* by_residual, size_t* code_size,
                              size_t* trained_size) {
    FA_TRY
    auto* sq = as<GpuIndexIVFScalarQuantizer>(index, "GpuIndexIVFScalarQuantizer");
    if (qtype) *qtype = sq->qtype;
    if (by_residual) *by_residual = sq->by_residual ? 1 : 0;
    if (code_size) *code_size = sq->code_size;
    if (trained_size) *trained_size = sq->trained.size();
    FA_CATCH
}
int faiss_amd_IndexIVFSQ_get_trained(const FaissAmdIndex* index, float* out) {
    FA_TRY
    auto* sq = as<GpuIndexIVFScalarQuantizer>(index, "GpuIndexIVFScalarQuantizer");
    memcpy(out, sq->trained.data(), sq->trained.size() * sizeof(float));
    FA_CATCH
}
int faiss_amd_IndexIVFSQ_copy_trained(FaissAmdIndex* index, const float* trained, size_t n) {
    FA_TRY
    as<GpuIndexIVFScalarQuantizer>(index, "GpuIndexIVFScalarQuantizer")->set_trained(trained, n);
    FA_CATCH
}
int faiss_amd_IndexIVFSQ_set_rangestat(FaissAmdIndex* index, int rangestat, float rangestat_arg) {
    FA_TRY
    auto* sq = as<GpuIndexIVFScalarQuantizer>(index, "GpuIndexIVFScalarQuantizer");
    sq->rangestat = rangestat;
    sq->rangestat_arg = rangestat_arg;
    FA_CATCH
}
int faiss_amd_sq_train_rangestat(int qtype, int rangestat, float rangestat_arg, int64_t n, int d, const float* rows, float* trained_out) {
    FA_TRY
    FA_THROW_IF_NOT_MSG(qtype == 0 || qtype == 1 || qtype == 2 || qtype == 3 || qtype == 6, "a quantizer type with a trained range");
    const bool uniform = qtype == 2 || qtype == 3;
    const int bits = (qtype == 1 || qtype == 3) ? 4 : qtype == 6 ? 6 : 8;
    std::vector<float> t;
    sq_train_rangestat_host(rangestat, rangestat_arg, n, d, 1 << bits, uniform, rows, t);
    memcpy(trained_out, t.data(), t.size() * 4);
    FA_CATCH
}
int faiss_amd_GpuIndexFlat_resident_bytes(const FaissAmdIndex* index, size_t* p_bytes) {
    FA_TRY
    *p_bytes = as<GpuIndexFlat>(index, "GpuIndexFlat")->resident_bytes();
    FA_CATCH
}
int faiss_amd_IndexShards_new(FaissAmdIndex** p_index, int d, int threaded, int successive_ids) {
    FA_TRY
    auto* h = new FaissAmdIndex_H{nullptr, nullptr};
    h->index = new IndexShards(d, threaded != 0, successive_ids != 0);
    *p_index = h;
    FA_CATCH
}
int faiss_amd_IndexShards_add_shard(FaissAmdIndex* shards, FaissAmdIndex* shard) {
    FA_TRY
    as<IndexShards>(shards, "IndexShards")->add_shard(I(shard));
    FA_CATCH
}
int faiss_amd_IndexReplicas_new(FaissAmdIndex** p_index, int d, int threaded) {
    FA_TRY
    auto* h = new FaissAmdIndex_H{nullptr, nullptr};
    h->index = new IndexReplicas(d, threaded != 0);
    *p_index = h;
    FA_CATCH
}
int faiss_amd_IndexReplicas_add_replica(FaissAmdIndex* replicas, FaissAmdIndex* replica) {
    FA_TRY
    as<IndexReplicas>(replicas, "IndexReplicas")->add_replica(I(replica));
    FA_CATCH
}
void faiss_amd_Index_free(FaissAmdIndex* index) {
    if (!index) return;
    delete index->index;
    delete index;
}

int faiss_amd_Index_d(const FaissAmdIndex* index) {
    return index && index->index ? index->index->d : -1;
}
int faiss_amd_Index_is_trained(const FaissAmdIndex* index) {
    return index && index->index ? (int)index->index->is_trained : 0;
}
faiss_amd_idx_t faiss_amd_Index_ntotal(const FaissAmdIndex* index) {
    return index && index->index ? index->index->ntotal : -1;
}
FaissAmdMetricType faiss_amd_Index_metric_type(const FaissAmdIndex* index) {
    return (FaissAmdMetricType)(index && index->index ? index->index->metric_type : 1);
}
float faiss_amd_Index_metric_arg(const FaissAmdIndex* index) {
    return index && index->index ? index->index->metric_arg : 0.f;
}
int faiss_amd_Index_set_metric_arg(FaissAmdIndex* index, float metric_arg) {
    FA_TRY
    I(index)->metric_arg = metric_arg;
    FA_CATCH
}
int faiss_amd_Index_train(FaissAmdIndex* index, faiss_amd_idx_t n, const float* x) {
    FA_TRY
    I(index)->train(n, x);
    FA_CATCH
}
int faiss_amd_Index_add(FaissAmdIndex* index, faiss_amd_idx_t n, const float* x) {
    FA_TRY
    I(index)->add(n, x);
    FA_CATCH
}
int faiss_amd_Index_add_with_ids(FaissAmdIndex* index, faiss_amd_idx_t n, const float* x,
                                 const faiss_amd_idx_t* xids) {
    FA_TRY
    I(index)->add_with_ids(n, x, xids);
    FA_CATCH
}
int faiss_amd_Index_search(const FaissAmdIndex* index, faiss_amd_idx_t n, const float* x, faiss_amd_idx_t k,
                           float* distances, faiss_amd_idx_t* labels) {
    FA_TRY
    I(index)->search(n, x, k, distances, labels);
    FA_CATCH
}
int faiss_amd_Index_assign(FaissAmdIndex* index, faiss_amd_idx_t n, const float* x, faiss_amd_idx_t* labels,
                           faiss_amd_idx_t k) {
    FA_TRY
    I(index)->assign(n, x, labels, k);
    FA_CATCH
}
int faiss_amd_Index_reset(FaissAmdIndex* index) {
    FA_TRY
    I(index)->reset();
    FA_CATCH
}
int faiss_amd_Index_reconstruct(const FaissAmdIndex* index, faiss_amd_idx_t key, float* recons) {
    FA_TRY
    I(index)->reconstruct(key, recons);
    FA_CATCH
}
int faiss_amd_Index_reconstruct_n(const FaissAmdIndex* index, faiss_amd_idx_t i0, faiss_amd_idx_t ni,
                                  float* recons) {
    FA_TRY
    I(index)->reconstruct_n(i0, ni, recons);
    FA_CATCH
}

int faiss_amd_Index_reconstruct_batch(const FaissAmdIndex* index, faiss_amd_idx_t n, const faiss_amd_idx_t* keys,
                                      float* recons) {
    FA_TRY
    I(index)->reconstruct_batch(n, keys, recons);
    FA_CATCH
}
int faiss_amd_Index_compute_residual(const FaissAmdIndex* index, const float* x, float* residual, faiss_amd_idx_t key) {
    FA_TRY
    I(index)->compute_residual(x, residual, key);
    FA_CATCH
}
int faiss_amd_Index_compute_residual_n(const FaissAmdIndex* index, faiss_amd_idx_t n, const float* xs, float* residuals,
                                       const faiss_amd_idx_t* keys) {
    FA_TRY
    I(index)->compute_residual_n(n, xs, residuals, keys);
    FA_CATCH
}

int faiss_amd_IndexIVF_nlist(const FaissAmdIndex* index, int* p) {
    FA_TRY
    *p = as<GpuIndexIVF>(index, "GpuIndexIVF")->nlist;
    FA_CATCH
}
int faiss_amd_IndexIVF_nprobe(const FaissAmdIndex* index, int* p) {
    FA_TRY
    *p = as<GpuIndexIVF>(index, "GpuIndexIVF")->nprobe;
    FA_CATCH
}
int faiss_amd_IndexIVF_set_nprobe(FaissAmdIndex* index, int nprobe) {
    FA_TRY
    FA_THROW_IF_NOT_MSG(nprobe >= 1 && nprobe <= kMaxSelectionK, "nprobe must be in [1, 2048]");
    as<GpuIndexIVF>(index, "GpuIndexIVF")->nprobe = nprobe;
    FA_CATCH
}
int faiss_amd_IndexIVF_get_list_size(const FaissAmdIndex* index, faiss_amd_idx_t list_no, size_t* p_size) {
    FA_TRY
    auto* ivf = as<GpuIndexIVF>(index, "GpuIndexIVF");
    FA_THROW_IF_NOT_MSG(list_no >= 0 && list_no < ivf->nlist, "list out of range");
    *p_size = ivf->getListLength(list_no);
    FA_CATCH
}
int faiss_amd_IndexIVF_get_list_ids(const FaissAmdIndex* index, faiss_amd_idx_t list_no,
                                    faiss_amd_idx_t* ids_out) {
    FA_TRY
    auto v = as<GpuIndexIVF>(index, "GpuIndexIVF")->getListIndices(list_no);
    if (!v.empty()) memcpy(ids_out, v.data(), v.size() * sizeof(idx_t));
    FA_CATCH
}
int faiss_amd_IndexIVF_get_list_codes(const FaissAmdIndex* index, faiss_amd_idx_t list_no, uint8_t* out) {
    FA_TRY
    auto v = as<GpuIndexIVF>(index, "GpuIndexIVF")->getListVectorData(list_no);
    if (!v.empty()) memcpy(out, v.data(), v.size());
    FA_CATCH
}
int faiss_amd_IndexIVF_code_size(const FaissAmdIndex* index, size_t* p) {
    FA_TRY
    *p = as<GpuIndexIVF>(index, "GpuIndexIVF")->ref_code_size();
    FA_CATCH
}
int faiss_amd_IndexIVF_get_centroids(const FaissAmdIndex* index, float* out) {
    FA_TRY
    auto* ivf = as<GpuIndexIVF>(index, "GpuIndexIVF");
    FA_THROW_IF_NOT_MSG(ivf->quantizer->ntotal == ivf->nlist, "coarse quantizer not trained");
    ivf->quantizer->reconstruct_n(0, ivf->nlist, out);
    FA_CATCH
}
int faiss_amd_IndexIVF_set_clustering(FaissAmdIndex* index, int niter, int seed) {
    FA_TRY
    auto* ivf = as<GpuIndexIVF>(index, "GpuIndexIVF");
    FA_THROW_IF_NOT_MSG(niter >= 1, "niter must be positive");
    ivf->cp_niter = niter;
    ivf->cp_seed = seed;
    FA_CATCH
}
int faiss_amd_IndexIVF_copy_centroids(FaissAmdIndex* index, const float* centroids) {
    FA_TRY
    as<GpuIndexIVF>(index, "GpuIndexIVF")->set_centroids(centroids);
    FA_CATCH
}
int faiss_amd_IndexIVFPQ_copy_pq_centroids(FaissAmdIndex* index, const float* pq) {
    FA_TRY
    as<GpuIndexIVFPQ>(index, "GpuIndexIVFPQ")->set_pq_centroids(pq);
    FA_CATCH
}
int faiss_amd_IndexIVFPQ_get_pq_centroids(const FaissAmdIndex* index, float* pq_out) {
    FA_TRY
    auto v = as<GpuIndexIVFPQ>(index, "GpuIndexIVFPQ")->get_pq_centroids();
    memcpy(pq_out, v.data(), v.size() * sizeof(float));
    FA_CATCH
}
int faiss_amd_IndexIVF_copy_lists(FaissAmdIndex* index, const uint32_t* list_sizes, const uint8_t* codes,
                                  const faiss_amd_idx_t* ids) {
    FA_TRY
    as<GpuIndexIVF>(index, "GpuIndexIVF")->set_lists(list_sizes, codes, ids);
    FA_CATCH
}

int faiss_amd_kmeans_clustering(FaissAmdGpuResources* res, int d, faiss_amd_idx_t n, int k, const float* x,
                                int niter, int seed, float* centroids_out, float* obj_out) {
    FA_TRY
    auto r = R(res);
    GpuIndexFlat index(r, d, METRIC_L2);
    Clustering clus(d, k);
    clus.niter = niter;
    clus.seed = seed;
    clus.train(n, x, index);
    memcpy(centroids_out, clus.centroids.data(), sizeof(float) * (size_t)k * d);
    if (obj_out) memcpy(obj_out, clus.obj.data(), sizeof(float) * clus.obj.size());
    FA_CATCH
}

int faiss_amd_Clustering_train(FaissAmdIndex* index, faiss_amd_idx_t n, const float* x, int k, int niter, int seed,
                               float* centroids_out, float* obj_out, int* on_device_out) {
    FA_TRY
    Index* ix = as<Index>(index, "Index");
    Clustering clus(ix->d, k);
    clus.niter = niter;
    clus.seed = seed;
    clus.train(n, x, *ix);
    memcpy(centroids_out, clus.centroids.data(), sizeof(float) * (size_t)k * ix->d);
    if (obj_out) memcpy(obj_out, clus.obj.data(), sizeof(float) * clus.obj.size());
    if (on_device_out) *on_device_out = clus.last_train_on_device ? 1 : 0;
    FA_CATCH
}

static void to_cp(const FaissAmdClusteringParameters& a, ClusteringParameters& c) {
    FA_THROW_IF_NOT_MSG(a.niter >= 0 && a.nredo >= 1, "clustering parameters: niter >= 0, nredo >= 1");
    FA_THROW_IF_NOT_MSG(a.max_points_per_centroid >= 1, "clustering parameters: max_points_per_centroid >= 1");
    c.niter = a.niter;
    c.nredo = a.nredo;
    c.verbose = a.verbose != 0;
    c.spherical = a.spherical != 0;
    c.int_centroids = a.int_centroids != 0;
    c.update_index = a.update_index != 0;
    c.frozen_centroids = a.frozen_centroids != 0;
    c.min_points_per_centroid = a.min_points_per_centroid;
    c.max_points_per_centroid = a.max_points_per_centroid;
    c.seed = a.seed;
}
void faiss_amd_ClusteringParameters_init(FaissAmdClusteringParameters* p) {
    if (!p) return;
    const ClusteringParameters d;
    *p = FaissAmdClusteringParameters{d.niter, d.nredo, 0, 0, 0, 0, 0, d.min_points_per_centroid, d.max_points_per_centroid, d.seed};
}
int faiss_amd_Clustering_train_ex(FaissAmdIndex* index, faiss_amd_idx_t n, const float* x, int k,
                                  const FaissAmdClusteringParameters* params, const float* init_centroids, int n_init,
                                  float* centroids_out, float* obj_out, int* on_device_out) {
    FA_TRY
    Index* ix = as<Index>(index, "Index");
    FA_THROW_IF_NOT_MSG(params && centroids_out, "null argument");
    FA_THROW_IF_NOT_MSG(n_init >= 0 && n_init <= k && (n_init == 0 || init_centroids), "bad initial centroids");
    Clustering clus(ix->d, k);
    to_cp(*params, clus);
    if (n_init) clus.centroids.assign(init_centroids, init_centroids + (size_t)n_init * ix->d);
    clus.train(n, x, *ix);
    memcpy(centroids_out, clus.centroids.data(), sizeof(float) * (size_t)k * ix->d);
    if (obj_out) memcpy(obj_out, clus.obj.data(), sizeof(float) * clus.obj.size());
    if (on_device_out) *on_device_out = clus.last_train_on_device ? 1 : 0;
    FA_CATCH
}
int faiss_amd_IndexIVF_set_clustering_params(FaissAmdIndex* index, const FaissAmdClusteringParameters* params) {
    FA_TRY
    auto* ivf = as<GpuIndexIVF>(index, "GpuIndexIVF");
    FA_THROW_IF_NOT_MSG(params, "null parameters");
    // validated before anything of the index changes; niter = 0 is legal (k-means that keeps its initial centroids,
    // faiss/Clustering.cpp:351-356 -- what faiss_amd_Clustering_train accepts too)
    FA_THROW_IF_NOT_MSG(params->niter >= 0, "niter must not be negative");
    ClusteringParameters tmp = ivf->cp;
    to_cp(*params, tmp);
    ivf->cp = tmp;
    ivf->cp_niter = params->niter;
    ivf->cp_seed = params->seed;
    FA_CATCH
}

int faiss_amd_merge_knn_results(FaissAmdMetricType metric, faiss_amd_idx_t n, faiss_amd_idx_t k, int nshard,
                                const float* all_d, const faiss_amd_idx_t* all_i,
                                const faiss_amd_idx_t* base, float* distances, faiss_amd_idx_t* labels) {
    FA_TRY
    FA_THROW_IF_NOT_MSG(nshard >= 1 && k >= 1 && n >= 0, "bad arguments");
    merge_knn_results((int)metric, n, k, nshard, all_d, all_i, base, distances, labels);
    FA_CATCH
}

int faiss_amd_merge_knn_results_device(FaissAmdGpuResources* res, FaissAmdMetricType metric, faiss_amd_idx_t n,
                                       faiss_amd_idx_t k, int nshard, const float* all_d,
                                       const faiss_amd_idx_t* all_i, const faiss_amd_idx_t* base,
                                       float* distances, faiss_amd_idx_t* labels) {
    FA_TRY
    merge_knn_results_device(*R(res), (int)metric, (int)n, (int)k, nshard, all_d, all_i, base, distances,
                             labels);
    FA_CATCH
}

int faiss_amd_profile_enable(FaissAmdGpuResources* res, int on) {
    FA_TRY
    R(res)->set_device();
    R(res)->collect();
    R(res)->profiling = on != 0;
    FA_CATCH
}
int faiss_amd_profile_reset(FaissAmdGpuResources* res) {
    FA_TRY
    R(res)->set_device();
    R(res)->reset_profile();
    FA_CATCH
}
int faiss_amd_profile_get(FaissAmdGpuResources* res, const char* kernel_name, double* total_ms,
                          long* launches) {
    FA_TRY
    auto r = R(res);
    r->set_device();
    r->collect();
    auto it = r->totals.find(kernel_name);
    if (it == r->totals.end()) {
        *total_ms = 0.0;
        *launches = 0;
    } else {
        *total_ms = it->second.first;
        *launches = it->second.second;
    }
    FA_CATCH
}

int faiss_amd_GpuIndexFlat_pairwise_distances(const FaissAmdIndex* index, faiss_amd_idx_t n, const float* x,
                                              float* out) {
    FA_TRY
    as<GpuIndexFlat>(index, "GpuIndexFlat")->pairwise_distances(n, x, out);
    FA_CATCH
}
int faiss_amd_GpuIndexFlat_set_use_simple_kernel(FaissAmdIndex* index, int on) {
    FA_TRY
    as<GpuIndexFlat>(index, "GpuIndexFlat")->use_simple_kernel = on != 0;
    FA_CATCH
}

int faiss_amd_GpuIndexFlat_set_use_filter_kernel(FaissAmdIndex* index, int on, faiss_amd_idx_t min_rows) {
    FA_TRY
    auto* f = as<GpuIndexFlat>(index, "GpuIndexFlat");
    f->use_filter_kernel = on != 0;
    if (min_rows >= 0) f->filter_min_rows = min_rows;
    FA_CATCH
}
int faiss_amd_GpuIndexFlat_filter_stats(const FaissAmdIndex* index, int* used_filter, int* overflow_queries) {
    FA_TRY
    auto* f = as<GpuIndexFlat>(index, "GpuIndexFlat");
    if (used_filter) *used_filter = f->last_used_filter ? 1 : 0;
    if (overflow_queries) *overflow_queries = f->last_filter_overflow;
    FA_CATCH
}
int faiss_amd_GpuIndexFlat_filter_scores(const FaissAmdIndex* index, faiss_amd_idx_t n, const float* x, float* scores,
                                         float* err_bound) {
    FA_TRY
    as<GpuIndexFlat>(index, "GpuIndexFlat")->filter_scores(n, x, scores, err_bound);
    FA_CATCH
}
int faiss_amd_GpuIndexIVF_reserveMemory(FaissAmdIndex* index, size_t num_vecs) {
    FA_TRY
    as<GpuIndexIVF>(index, "GpuIndexIVF")->reserveMemory(num_vecs);
    FA_CATCH
}
int faiss_amd_GpuIndexIVF_reclaimMemory(FaissAmdIndex* index, size_t* p_bytes) {
    FA_TRY
    const size_t b = as<GpuIndexIVF>(index, "GpuIndexIVF")->reclaimMemory();
    if (p_bytes) *p_bytes = b;
    FA_CATCH
}
int faiss_amd_GpuIndexIVF_updateQuantizer(FaissAmdIndex* index) {
    FA_TRY
    as<GpuIndexIVF>(index, "GpuIndexIVF")->updateQuantizer();
    FA_CATCH
}
int faiss_amd_GpuIndexIVFPQ_setPrecomputedCodes(FaissAmdIndex* index, int enable) {
    FA_TRY
    as<GpuIndexIVFPQ>(index, "GpuIndexIVFPQ")->setPrecomputedCodes(enable != 0);
    FA_CATCH
}
int faiss_amd_GpuIndexIVFPQ_getInfo(const FaissAmdIndex* index, int* precomputed_codes, int* num_sub_quantizers,
                                    int* bits_per_code, int* centroids_per_sub_quantizer) {
    FA_TRY
    const GpuIndexIVFPQ* pq = as<GpuIndexIVFPQ>(index, "GpuIndexIVFPQ");
    if (precomputed_codes) *precomputed_codes = pq->getPrecomputedCodes() ? 1 : 0;
    if (num_sub_quantizers) *num_sub_quantizers = pq->getNumSubQuantizers();
    if (bits_per_code) *bits_per_code = pq->getBitsPerCode();
    if (centroids_per_sub_quantizer) *centroids_per_sub_quantizer = pq->getCentroidsPerSubQuantizer();
    FA_CATCH
}
int faiss_amd_GpuIndexIVFPQ_getTableInfo(const FaissAmdIndex* index, int* precomputed_in_force, int* precomputed_requested,
                                         int* float16_tables_in_force) {
    FA_TRY
    const GpuIndexIVFPQ* pq = as<GpuIndexIVFPQ>(index, "GpuIndexIVFPQ");
    if (precomputed_in_force) *precomputed_in_force = pq->getPrecomputedCodes() ? 1 : 0;
    if (precomputed_requested) *precomputed_requested = pq->getPrecomputedCodesRequested() ? 1 : 0;
    if (float16_tables_in_force) *float16_tables_in_force = pq->getFloat16LookupTables() ? 1 : 0;
    FA_CATCH
}
int faiss_amd_GpuIndexIVF_add_core(FaissAmdIndex* index, faiss_amd_idx_t n, const float* x, const faiss_amd_idx_t* xids,
                                   const faiss_amd_idx_t* precomputed_idx) {
    FA_TRY
    as<GpuIndexIVF>(index, "GpuIndexIVF")->add_core(n, x, xids, precomputed_idx);
    FA_CATCH
}
int faiss_amd_GpuIndexIVF_search_preassigned(const FaissAmdIndex* index, faiss_amd_idx_t n, const float* x,
                                             faiss_amd_idx_t k, const faiss_amd_idx_t* assign, const float* centroid_dis,
                                             float* distances, faiss_amd_idx_t* labels) {
    FA_TRY
    as<GpuIndexIVF>(index, "GpuIndexIVF")->search_preassigned(n, x, k, assign, centroid_dis, distances, labels);
    FA_CATCH
}
int faiss_amd_IndexIVF_quantizer_search(const FaissAmdIndex* index, faiss_amd_idx_t n, const float* x, faiss_amd_idx_t k,
                                        float* distances, faiss_amd_idx_t* labels) {
    FA_TRY
    as<GpuIndexIVF>(index, "GpuIndexIVF")->quantizer->search(n, x, k, distances, labels);
    FA_CATCH
}
int faiss_amd_bfKnn(FaissAmdGpuResources* res, FaissAmdMetricType metric, const float* vectors,
                    faiss_amd_idx_t num_vectors, const float* queries, faiss_amd_idx_t num_queries, int dims,
                    faiss_amd_idx_t k, float* out_distances, faiss_amd_idx_t* out_indices) {
    FA_TRY
    bfKnn(R(res), (int)metric, vectors, num_vectors, queries, num_queries, dims, k, out_distances, out_indices);
    FA_CATCH
}
int faiss_amd_test_select(FaissAmdGpuResources* res, int which, FaissAmdMetricType metric, int rows, int cols, int k,
                          const float* vals, float* out_distances, faiss_amd_idx_t* out_indices) {
    FA_TRY
    auto r = R(res);
    r->set_device();
    FA_THROW_IF_NOT_MSG(which >= 0 && which <= 2 && rows >= 0 && cols >= 1, "bad arguments");
    DevBuf dv, dk, dc, dd, di;
    const size_t n = (size_t)rows * cols;
    dv.ensure(std::max<size_t>(n * 4, 16));
    dk.ensure(std::max<size_t>(n * 8, 16));
    dc.ensure(std::max<size_t>((size_t)rows * 4, 16));
    dd.ensure(std::max<size_t>((size_t)rows * k * 4, 16));
    di.ensure(std::max<size_t>((size_t)rows * k * 8, 16));
    HIP_CHECK(hipMemcpyAsync(dv.p, vals, n * 4, hipMemcpyHostToDevice, r->stream));
    launch_select_test(which, (int)metric, dv.as<float>(), rows, cols, k, dd.as<float>(), di.as<idx_t>(),
                       dk.as<unsigned long long>(), dc.as<uint32_t>(), r->stream);
    HIP_CHECK(hipMemcpyAsync(out_distances, dd.p, (size_t)rows * k * 4, hipMemcpyDeviceToHost, r->stream));
    HIP_CHECK(hipMemcpyAsync(out_indices, di.p, (size_t)rows * k * 8, hipMemcpyDeviceToHost, r->stream));
    r->sync();
    FA_CATCH
}
int faiss_amd_GpuIndexIVF_search_with_params(const FaissAmdIndex* index, faiss_amd_idx_t n, const float* x,
                                             faiss_amd_idx_t k, const FaissAmdSearchParametersIVF* params,
                                             float* distances, faiss_amd_idx_t* labels) {
    FA_TRY
    SearchParametersIVF sp;
    if (params) {
        FA_THROW_IF_NOT_MSG(params->nprobe <= kMaxSelectionK, "nprobe must be in [1, 2048]");
        sp.nprobe = params->nprobe;
    }
    as<GpuIndexIVF>(index, "GpuIndexIVF")->search(n, x, k, distances, labels, &sp);
    FA_CATCH
}
// ---- IDSelector / SearchParameters
struct FaissAmdIDSelector_H {
    std::unique_ptr<IDSelector> sel;
};
struct FaissAmdSearchParameters_H {
    std::unique_ptr<SearchParameters> sp;
};
static const IDSelector* S(const FaissAmdIDSelector* h) {
    if (!h || !h->sel) FA_THROW_MSG("null IDSelector handle");
    return h->sel.get();
}
static int new_selector(FaissAmdIDSelector** p_sel, IDSelector* sel) {
    std::unique_ptr<IDSelector> guard(sel);
    if (!p_sel) FA_THROW_MSG("null output handle");
    *p_sel = new FaissAmdIDSelector_H{std::move(guard)};
    return 0;
}
int faiss_amd_IDSelectorAll_new(FaissAmdIDSelector** p_sel) {
    FA_TRY
    new_selector(p_sel, new IDSelectorAll());
    FA_CATCH
}
int faiss_amd_IDSelectorRange_new(FaissAmdIDSelector** p_sel, faiss_amd_idx_t imin, faiss_amd_idx_t imax) {
    FA_TRY
    new_selector(p_sel, new IDSelectorRange(imin, imax));
    FA_CATCH
}
int faiss_amd_IDSelectorBatch_new(FaissAmdIDSelector** p_sel, size_t n, const faiss_amd_idx_t* ids) {
    FA_TRY
    FA_THROW_IF_NOT_MSG(n == 0 || ids, "null id array");
    new_selector(p_sel, new IDSelectorBatch(n, ids));
    FA_CATCH
}
int faiss_amd_IDSelectorArray_new(FaissAmdIDSelector** p_sel, size_t n, const faiss_amd_idx_t* ids) {
    return faiss_amd_IDSelectorBatch_new(p_sel, n, ids);
}
int faiss_amd_IDSelectorBitmap_new(FaissAmdIDSelector** p_sel, size_t n, const uint8_t* bitmap) {
    FA_TRY
    FA_THROW_IF_NOT_MSG(n == 0 || bitmap, "null bitmap");
    new_selector(p_sel, new IDSelectorBitmap(n, bitmap));
    FA_CATCH
}
int faiss_amd_IDSelectorNot_new(FaissAmdIDSelector** p_sel, const FaissAmdIDSelector* sel) {
    FA_TRY
    new_selector(p_sel, new IDSelectorNot(S(sel)));
    FA_CATCH
}
int faiss_amd_IDSelectorAnd_new(FaissAmdIDSelector** p_sel, const FaissAmdIDSelector* lhs, const FaissAmdIDSelector* rhs) {
    FA_TRY
    new_selector(p_sel, new IDSelectorBinary(SEL_AND, S(lhs), S(rhs)));
    FA_CATCH
}
int faiss_amd_IDSelectorOr_new(FaissAmdIDSelector** p_sel, const FaissAmdIDSelector* lhs, const FaissAmdIDSelector* rhs) {
    FA_TRY
    new_selector(p_sel, new IDSelectorBinary(SEL_OR, S(lhs), S(rhs)));
    FA_CATCH
}
int faiss_amd_IDSelectorXOr_new(FaissAmdIDSelector** p_sel, const FaissAmdIDSelector* lhs, const FaissAmdIDSelector* rhs) {
    FA_TRY
    new_selector(p_sel, new IDSelectorBinary(SEL_XOR, S(lhs), S(rhs)));
    FA_CATCH
}
int faiss_amd_IDSelector_is_member(const FaissAmdIDSelector* sel, faiss_amd_idx_t id) {
    FA_TRY
    return S(sel)->is_member(id) ? 1 : 0;
    FA_CATCH
}
void faiss_amd_IDSelector_free(FaissAmdIDSelector* sel) {
    delete sel;
}
int faiss_amd_SearchParameters_new(FaissAmdSearchParameters** p_sp, const FaissAmdIDSelector* sel) {
    FA_TRY
    if (!p_sp) FA_THROW_MSG("null output handle");
    std::unique_ptr<SearchParameters> sp(new SearchParameters());
    sp->sel = sel ? S(sel) : nullptr;
    *p_sp = new FaissAmdSearchParameters_H{std::move(sp)};
    FA_CATCH
}
int faiss_amd_SearchParametersIVF_new_with(FaissAmdSearchParameters** p_sp, const FaissAmdIDSelector* sel, size_t nprobe,
                                           size_t max_codes) {
    FA_TRY
    if (!p_sp) FA_THROW_MSG("null output handle");
    FA_THROW_IF_NOT_MSG(max_codes == 0, "max_codes is not supported (faiss/gpu/GpuIndexIVF.cu:372-375)");
    FA_THROW_IF_NOT_MSG(nprobe <= (size_t)kMaxSelectionK, "nprobe must be in [1, 2048]");
    std::unique_ptr<SearchParametersIVF> sp(new SearchParametersIVF());
    sp->sel = sel ? S(sel) : nullptr;
    sp->nprobe = (int)nprobe;
    *p_sp = new FaissAmdSearchParameters_H{std::move(sp)};
    FA_CATCH
}
void faiss_amd_SearchParameters_free(FaissAmdSearchParameters* sp) {
    delete sp;
}
int faiss_amd_Index_search_with_params(const FaissAmdIndex* index, faiss_amd_idx_t n, const float* x, faiss_amd_idx_t k,
                                       const FaissAmdSearchParameters* params, float* distances, faiss_amd_idx_t* labels) {
    FA_TRY
    I(index)->search(n, x, k, distances, labels, params ? params->sp.get() : nullptr);
    FA_CATCH
}
int faiss_amd_GpuIndexIVF_search_preassigned_with_params(const FaissAmdIndex* index, faiss_amd_idx_t n, const float* x,
                                                         faiss_amd_idx_t k, const faiss_amd_idx_t* assign,
                                                         const float* centroid_dis, const FaissAmdSearchParameters* params,
                                                         float* distances, faiss_amd_idx_t* labels) {
    FA_TRY
    as<GpuIndexIVF>(index, "GpuIndexIVF")
            ->search_preassigned(n, x, k, assign, centroid_dis, distances, labels, params ? params->sp.get() : nullptr);
    FA_CATCH
}
int faiss_amd_GpuIndexIVF_stored_vectors(const FaissAmdIndex* index, faiss_amd_idx_t* p_stored) {
    FA_TRY
    *p_stored = as<GpuIndexIVF>(index, "GpuIndexIVF")->stored_vectors();
    FA_CATCH
}
int faiss_amd_GpuIndexIVF_arena_stats(const FaissAmdIndex* index, int64_t* used_rows, int64_t* hole_rows,
                                      int64_t* allocated_rows) {
    FA_TRY
    as<GpuIndexIVF>(index, "GpuIndexIVF")->arena_stats(used_rows, hole_rows, allocated_rows);
    FA_CATCH
}
int faiss_amd_set_interrupt_callback(faiss_amd_interrupt_fn fn, void* user) {
    FA_TRY
    set_interrupt_callback(fn, user);
    FA_CATCH
}
int faiss_amd_GpuParameterSpace_set_index_parameter(FaissAmdIndex* index, const char* name, double value) {
    FA_TRY
    FA_THROW_IF_NOT_MSG(name, "null parameter name");
    set_index_parameter(I(index), name, value);
    FA_CATCH
}
static DistanceParams to_params(const FaissAmdGpuDistanceParams* a) {
    FA_THROW_IF_NOT_MSG(a, "null args");
    DistanceParams p;
    p.metric = a->metric;
    p.metricArg = a->metricArg;
    p.k = a->k;
    p.dims = a->dims;
    p.vectors = a->vectors;
    p.vectorType = a->vectorType;
    p.vectorsRowMajor = a->vectorsRowMajor != 0;
    p.numVectors = a->numVectors;
    p.vectorNorms = a->vectorNorms;
    p.queries = a->queries;
    p.queryType = a->queryType;
    p.queriesRowMajor = a->queriesRowMajor != 0;
    p.numQueries = a->numQueries;
    p.outDistances = a->outDistances;
    p.ignoreOutDistances = a->ignoreOutDistances != 0;
    p.outIndicesType = a->outIndicesType;
    p.outIndices = a->outIndices;
    p.device = a->device;
    return p;
}
int faiss_amd_bfKnn_params(FaissAmdGpuResources* res, const FaissAmdGpuDistanceParams* args) {
    FA_TRY
    bfKnn(R(res), to_params(args));
    FA_CATCH
}
int faiss_amd_bfKnn_tiling(FaissAmdGpuResources* res, const FaissAmdGpuDistanceParams* args, size_t vectorsMemoryLimit,
                           size_t queriesMemoryLimit) {
    FA_TRY
    bfKnn_tiling(R(res), to_params(args), vectorsMemoryLimit, queriesMemoryLimit);
    FA_CATCH
}
int faiss_amd_GpuIndexIVF_set_use_fused_scan(FaissAmdIndex* index, int on) {
    FA_TRY
    as<GpuIndexIVF>(index, "GpuIndexIVF")->use_fused_scan = on != 0;
    FA_CATCH
}
int faiss_amd_GpuIndexIVF_set_scan_mode(FaissAmdIndex* index, int mode) {
    FA_TRY
    FA_THROW_IF_NOT_MSG(mode >= 0 && mode <= 3,
                        "scan mode: 0 = automatic, 1 = query-major, 2 = list-major, 3 = list-major on the f32 matrix pipe");
    as<GpuIndexIVF>(index, "GpuIndexIVF")->scan_mode = mode;
    FA_CATCH
}
int faiss_amd_GpuIndexIVF_set_use_filter_shadow(FaissAmdIndex* index, int on) {
    FA_TRY
    as<GpuIndexIVF>(index, "GpuIndexIVF")->use_filter_shadow = on != 0;
    FA_CATCH
}
int faiss_amd_GpuIndexIVF_resident_bytes(const FaissAmdIndex* index, size_t* p_lists, size_t* p_shadow) {
    FA_TRY
    as<GpuIndexIVF>(const_cast<FaissAmdIndex*>(index), "GpuIndexIVF")->resident_bytes(p_lists, p_shadow);
    FA_CATCH
}
int faiss_amd_GpuIndexIVF_scan_info(const FaissAmdIndex* index, int* mode, int* last_mode, int64_t* overflow_queries) {
    FA_TRY
    auto* ix = as<GpuIndexIVF>(const_cast<FaissAmdIndex*>(index), "GpuIndexIVF");
    if (mode) *mode = ix->scan_mode;
    if (last_mode) *last_mode = ix->last_scan_mode();
    if (overflow_queries) *overflow_queries = ix->list_major_overflows();
    FA_CATCH
}
int faiss_amd_GpuIndexIVF_set_lmf_tuning(FaissAmdIndex* index, int rows_per_item, int gran_blocks, int cand_cap, int min_stride) {
    FA_TRY
    auto* ix = as<GpuIndexIVF>(index, "GpuIndexIVF");
    FA_THROW_IF_NOT_MSG(rows_per_item >= 0 && gran_blocks >= 0 && cand_cap >= 0 && min_stride >= 0, "negative tuning value");
    // the ranges the sweeps are built for: a bad value must fail HERE, not as an internal assertion of the next search
    FA_THROW_IF_NOT_MSG(rows_per_item <= 65280, "rows_per_item: at most 65280 rows of a list per work item");
    FA_THROW_IF_NOT_MSG(gran_blocks <= 32 && (gran_blocks & (gran_blocks - 1)) == 0,
                        "gran_blocks: 0 (rule) or a power of two up to 32");
    FA_THROW_IF_NOT_MSG(min_stride <= 8, "min_stride: 0 (rule) ... 8");
    FA_THROW_IF_NOT_MSG(cand_cap == 0 || (cand_cap >= 256 && cand_cap <= 16384), "cand_cap: 0 (rule) or 256 ... 16384 per query");
    ix->lmf_rows_per_item = rows_per_item;
    ix->lmf_gran_blocks = gran_blocks;
    ix->lmf_cand_cap = cand_cap;
    ix->lmf_min_stride = min_stride;
    FA_CATCH
}
int faiss_amd_GpuIndexIVF_set_lmf_sampling(FaissAmdIndex* index, int sample_shift) {
    FA_TRY
    FA_THROW_IF_NOT_MSG(sample_shift >= -1 && sample_shift <= 4, "sample_shift: -1 (all rows), 0 (rule), 1 ... 4");
    as<GpuIndexIVF>(index, "GpuIndexIVF")->lmf_sample_shift = sample_shift;
    FA_CATCH
}
int faiss_amd_GpuIndexIVF_set_lmf_pair(FaissAmdIndex* index, int on) {
    FA_TRY
    FA_THROW_IF_NOT_MSG(on >= 0 && on <= 2, "0 off, 1 sweep 1 only, 2 both sweeps");
    as<GpuIndexIVF>(index, "GpuIndexIVF")->lmf_pair = on;
    FA_CATCH
}
int faiss_amd_Index_set_small_fused(FaissAmdIndex* index, int on) {
    FA_TRY
    if (auto* ivf = dynamic_cast<GpuIndexIVF*>(I(index))) ivf->quantizer->use_small_fused = on != 0;
    else as<GpuIndexFlat>(index, "GpuIndexFlat")->use_small_fused = on != 0;
    FA_CATCH
}
int faiss_amd_GpuIndexIVFPQ_set_lmf_fast_gather(FaissAmdIndex* index, int on) {
    FA_TRY
    as<GpuIndexIVFPQ>(index, "GpuIndexIVFPQ")->lmf_fast_gather = on != 0;
    FA_CATCH
}
int faiss_amd_GpuIndexIVFPQ_set_lmf_two_copies(FaissAmdIndex* index, int on) {
    FA_TRY
    as<GpuIndexIVFPQ>(index, "GpuIndexIVFPQ")->set_lmf_two_copies(on != 0);
    FA_CATCH
}
int faiss_amd_GpuIndexIVF_test_filter_dump(const FaissAmdIndex* index, int64_t n, const float* x, int nprobe, int64_t k, int64_t stride,
                                           uint64_t* keys_out, float* band_out) {
    FA_TRY
    as<GpuIndexIVF>(const_cast<FaissAmdIndex*>(index), "GpuIndexIVF")
            ->test_filter_dump(n, x, nprobe, k, stride, (unsigned long long*)keys_out, band_out);
    FA_CATCH
}
int faiss_amd_GpuIndexIVF_last_scan_arith(const FaissAmdIndex* index, int* p_arith) {
    FA_TRY
    FA_THROW_IF_NOT_MSG(p_arith, "null output");
    *p_arith = as<GpuIndexIVF>(const_cast<FaissAmdIndex*>(index), "GpuIndexIVF")->last_scan_arith();
    FA_CATCH
}
int faiss_amd_GpuIndexIVF_list_major_rule(const FaissAmdIndex* index, int64_t n, int nprobe, int64_t k, int* p_output) {
    FA_TRY
    FA_THROW_IF_NOT_MSG(p_output, "null output");
    *p_output = as<GpuIndexIVF>(const_cast<FaissAmdIndex*>(index), "GpuIndexIVF")->list_major_rule(n, nprobe, k, false) ? 1 : 0;
    FA_CATCH
}

} // extern "C"
