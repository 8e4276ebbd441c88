// oracle/ref_shim.cpp -- plain-C access to the UNMODIFIED reference library (faiss v1.15.0 CPU
// path compiled from /root/reference by oracle/Makefile.ref) for ctypes.
//
// TEST INFRASTRUCTURE ONLY: loaded by tests/ and by bench.py's cpu_baseline leg.  This file
// is our own code; it includes the reference's public headers and calls its public API.
//
// Besides the oracle calls it contains the drop-in proof: `AmdIndexAdapter`, a faiss::Index
// subclass that forwards add/search/reset to a faiss_amd C-ABI handle (function pointers are
// handed in by the caller, so this library does not link against HIP).  With it the reference's
// own callers of the hot path -- faiss::Clustering::train (faiss/Clustering.cpp:255-357) and
// faiss::IndexShards (faiss/IndexShards.cpp:135-265) -- run unchanged on the MI355X backend.
#include <faiss/Clustering.h>
#include <faiss/Index.h>
#include <faiss/IndexFlat.h>
#include <faiss/IndexIVF.h>
#include <faiss/IndexIVFFlat.h>
#include <faiss/IndexIVFPQ.h>
#include <faiss/IndexShards.h>
#include <faiss/index_factory.h>
#include <faiss/impl/FaissException.h>
#include <faiss/utils/distances.h>
#include <omp.h>
#include <cstring>
#include <string>

using faiss::idx_t;

static thread_local std::string g_err;
#define SHIM_TRY try {
#define SHIM_CATCH            \
    }                         \
    catch (std::exception & e) { \
        g_err = e.what();     \
        return -1;            \
    }                         \
    return 0;

// ---------------------------------------------------------------- adapter (drop-in boundary)
typedef int (*amd_add_fn)(void*, int64_t, const float*);
typedef int (*amd_add_ids_fn)(void*, int64_t, const float*, const int64_t*);
typedef int (*amd_search_fn)(const void*, int64_t, const float*, int64_t, float*, int64_t*);
typedef int (*amd_reset_fn)(void*);
typedef int (*amd_train_fn)(void*, int64_t, const float*);
typedef int64_t (*amd_ntotal_fn)(const void*);
typedef int (*amd_trained_fn)(const void*);
typedef const char* (*amd_err_fn)(void);

struct AmdIndexAdapter : faiss::Index {
    void* h;
    amd_add_fn f_add;
    amd_add_ids_fn f_add_ids;
    amd_search_fn f_search;
    amd_reset_fn f_reset;
    amd_train_fn f_train;
    amd_ntotal_fn f_ntotal;
    amd_trained_fn f_trained;
    amd_err_fn f_err;

    AmdIndexAdapter(int d_, faiss::MetricType m) : faiss::Index(d_, m) {}
    void check(int rc) const {
        if (rc != 0) FAISS_THROW_MSG(f_err ? f_err() : "faiss_amd error");
    }
    void sync() {
        ntotal = f_ntotal(h);
        is_trained = f_trained(h) != 0;
    }
    void train(idx_t n, const float* x) override {
        check(f_train(h, n, x));
        sync();
    }
    void add(idx_t n, const float* x) override {
        check(f_add(h, n, x));
        sync();
    }
    void add_with_ids(idx_t n, const float* x, const idx_t* xids) override {
        check(f_add_ids(h, n, x, xids));
        sync();
    }
    void search(idx_t n, const float* x, idx_t k, float* distances, idx_t* labels,
                const faiss::SearchParameters* params = nullptr) const override {
        FAISS_THROW_IF_NOT_MSG(!params, "search params not supported");
        check(f_search(h, n, x, k, distances, labels));
    }
    void reset() override {
        check(f_reset(h));
        sync();
    }
};

extern "C" {

const char* ref_last_error() {
    return g_err.c_str();
}
int ref_version(int* major, int* minor, int* patch) {
    *major = FAISS_VERSION_MAJOR;
    *minor = FAISS_VERSION_MINOR;
    *patch = FAISS_VERSION_PATCH;
    return 0;
}
int ref_set_omp_threads(int n) {
    omp_set_num_threads(n);
    return 0;
}
int ref_get_max_threads() {
    return omp_get_max_threads();
}

void* ref_index_factory(int d, const char* desc, int metric) {
    try {
        return faiss::index_factory(d, desc, (faiss::MetricType)metric);
    } catch (std::exception& e) {
        g_err = e.what();
        return nullptr;
    }
}
void ref_index_free(void* p) {
    delete (faiss::Index*)p;
}
int ref_index_train(void* p, idx_t n, const float* x) {
    SHIM_TRY((faiss::Index*)p)->train(n, x);
    SHIM_CATCH
}
int ref_index_add(void* p, idx_t n, const float* x) {
    SHIM_TRY((faiss::Index*)p)->add(n, x);
    SHIM_CATCH
}
int ref_index_add_with_ids(void* p, idx_t n, const float* x, const idx_t* ids) {
    SHIM_TRY((faiss::Index*)p)->add_with_ids(n, x, ids);
    SHIM_CATCH
}
int ref_index_search(void* p, idx_t n, const float* x, idx_t k, float* D, idx_t* I) {
    SHIM_TRY((faiss::Index*)p)->search(n, x, k, D, I);
    SHIM_CATCH
}
int ref_index_reset(void* p) {
    SHIM_TRY((faiss::Index*)p)->reset();
    SHIM_CATCH
}
idx_t ref_index_ntotal(void* p) {
    return ((faiss::Index*)p)->ntotal;
}
int ref_index_is_trained(void* p) {
    return ((faiss::Index*)p)->is_trained;
}

static faiss::IndexIVF* ivf(void* p) {
    auto* r = dynamic_cast<faiss::IndexIVF*>((faiss::Index*)p);
    if (!r) FAISS_THROW_MSG("not an IndexIVF");
    return r;
}
int ref_ivf_set_nprobe(void* p, int nprobe) {
    SHIM_TRY ivf(p)->nprobe = nprobe;
    SHIM_CATCH
}
int ref_ivf_nlist(void* p) {
    try {
        return (int)ivf(p)->nlist;
    } catch (...) {
        return -1;
    }
}
int ref_ivf_code_size(void* p) {
    try {
        return (int)ivf(p)->code_size;
    } catch (...) {
        return -1;
    }
}
int ref_ivf_get_centroids(void* p, float* out) {
    SHIM_TRY auto* i = ivf(p);
    i->quantizer->reconstruct_n(0, i->nlist, out);
    SHIM_CATCH
}
int ref_ivf_list_sizes(void* p, uint32_t* out) {
    SHIM_TRY auto* i = ivf(p);
    for (size_t l = 0; l < i->nlist; l++) out[l] = (uint32_t)i->invlists->list_size(l);
    SHIM_CATCH
}
// codes / ids of all lists concatenated in list order (invlists->get_codes / get_ids)
int ref_ivf_get_lists(void* p, uint8_t* codes, idx_t* ids) {
    SHIM_TRY auto* i = ivf(p);
    size_t cs = i->code_size, off = 0;
    for (size_t l = 0; l < i->nlist; l++) {
        size_t n = i->invlists->list_size(l);
        if (!n) continue;
        faiss::InvertedLists::ScopedCodes sc(i->invlists, l);
        faiss::InvertedLists::ScopedIds si(i->invlists, l);
        memcpy(codes + off * cs, sc.get(), n * cs);
        memcpy(ids + off, si.get(), n * sizeof(idx_t));
        off += n;
    }
    SHIM_CATCH
}
int ref_ivfpq_info(void* p, int* M, int* dsub, int* nbits, int* use_precomputed_table) {
    SHIM_TRY auto* i = dynamic_cast<faiss::IndexIVFPQ*>((faiss::Index*)p);
    FAISS_THROW_IF_NOT_MSG(i, "not an IndexIVFPQ");
    *M = (int)i->pq.M;
    *dsub = (int)i->pq.dsub;
    *nbits = (int)i->pq.nbits;
    *use_precomputed_table = i->use_precomputed_table;
    SHIM_CATCH
}
int ref_ivfpq_get_pq_centroids(void* p, float* out) {
    SHIM_TRY auto* i = dynamic_cast<faiss::IndexIVFPQ*>((faiss::Index*)p);
    FAISS_THROW_IF_NOT_MSG(i, "not an IndexIVFPQ");
    memcpy(out, i->pq.centroids.data(), sizeof(float) * i->pq.centroids.size());
    SHIM_CATCH
}
// 0 = compute the table per (query, list) as the GPU reference does; 1 = precomputed tables
int ref_ivfpq_set_precomputed_table(void* p, int use) {
    SHIM_TRY auto* i = dynamic_cast<faiss::IndexIVFPQ*>((faiss::Index*)p);
    FAISS_THROW_IF_NOT_MSG(i, "not an IndexIVFPQ");
    i->use_precomputed_table = use;
    i->precompute_table();
    SHIM_CATCH
}

// copyTo in the other direction: install a trained state (coarse centroids nlist x d, PQ codebook
// M x 256 x dsub) into an untrained reference IndexIVFPQ, so that a CPU baseline can run on exactly the
// quantizers the GPU index trained (faiss/gpu/GpuIndexIVFPQ.cu:170-217 copyTo does the same).
int ref_ivfpq_set_trained(void* p, const float* centroids, const float* pq_centroids) {
    SHIM_TRY auto* i = dynamic_cast<faiss::IndexIVFPQ*>((faiss::Index*)p);
    FAISS_THROW_IF_NOT_MSG(i, "not an IndexIVFPQ");
    i->quantizer->reset();
    i->quantizer->add(i->nlist, centroids);
    i->quantizer->is_trained = true;
    FAISS_THROW_IF_NOT(i->pq.centroids.size() == (size_t)i->pq.M * i->pq.ksub * i->pq.dsub);
    memcpy(i->pq.centroids.data(), pq_centroids, sizeof(float) * i->pq.centroids.size());
    i->is_trained = true;
    i->use_precomputed_table = 0; // automatic choice, as after train()
    i->precompute_table();
    SHIM_CATCH
}

// same for any IndexIVF whose only trained state is the coarse quantizer (IndexIVFFlat)
int ref_ivf_set_centroids(void* p, const float* centroids) {
    SHIM_TRY auto* i = ivf(p);
    i->quantizer->reset();
    i->quantizer->add(i->nlist, centroids);
    i->quantizer->is_trained = true;
    i->is_trained = true;
    SHIM_CATCH
}

// k-means iteration counts of an untrained IVF(PQ) index: coarse quantizer (IndexIVF::cp) and product
// quantizer (ProductQuantizer::cp); <= 0 leaves a value unchanged.  Used to bound the training time of
// the CPU baseline (search speed does not depend on it).
int ref_ivf_set_train_niter(void* p, int niter_coarse, int niter_pq) {
    SHIM_TRY auto* ivf = dynamic_cast<faiss::IndexIVF*>((faiss::Index*)p);
    FAISS_THROW_IF_NOT_MSG(ivf, "not an IndexIVF");
    if (niter_coarse > 0) ivf->cp.niter = niter_coarse;
    auto* pq = dynamic_cast<faiss::IndexIVFPQ*>(ivf);
    if (pq && niter_pq > 0) pq->pq.cp.niter = niter_pq;
    SHIM_CATCH
}

// k-means with the reference's own CPU assignment index; returns the final objective
int ref_kmeans(int d, idx_t n, int k, const float* x, int niter, int seed, float* centroids, float* obj_out) {
    SHIM_TRY faiss::ClusteringParameters cp;
    cp.niter = niter;
    cp.seed = seed;
    faiss::Clustering clus(d, k, cp);
    faiss::IndexFlatL2 index(d);
    clus.train(n, x, index);
    memcpy(centroids, clus.centroids.data(), sizeof(float) * (size_t)k * d);
    if (obj_out) *obj_out = clus.iteration_stats.back().obj;
    SHIM_CATCH
}

// ------------------------------------------------------------ drop-in: adapter over the C ABI
void* ref_amd_adapter_new(int d, int metric, void* handle, void* f_add, void* f_add_ids, void* f_search,
                          void* f_reset, void* f_train, void* f_ntotal, void* f_trained, void* f_err) {
    auto* a = new AmdIndexAdapter(d, (faiss::MetricType)metric);
    a->h = handle;
    a->f_add = (amd_add_fn)f_add;
    a->f_add_ids = (amd_add_ids_fn)f_add_ids;
    a->f_search = (amd_search_fn)f_search;
    a->f_reset = (amd_reset_fn)f_reset;
    a->f_train = (amd_train_fn)f_train;
    a->f_ntotal = (amd_ntotal_fn)f_ntotal;
    a->f_trained = (amd_trained_fn)f_trained;
    a->f_err = (amd_err_fn)f_err;
    a->sync();
    return (faiss::Index*)a;
}
// faiss::Clustering::train driving ANY faiss::Index* (e.g. an adapter) as assignment engine
int ref_kmeans_with_index(int d, idx_t n, int k, const float* x, int niter, int seed, void* index,
                          float* centroids, float* obj_out) {
    SHIM_TRY faiss::ClusteringParameters cp;
    cp.niter = niter;
    cp.seed = seed;
    faiss::Clustering clus(d, k, cp);
    clus.train(n, x, *(faiss::Index*)index);
    memcpy(centroids, clus.centroids.data(), sizeof(float) * (size_t)k * d);
    if (obj_out) *obj_out = clus.iteration_stats.back().obj;
    SHIM_CATCH
}
// the reference's own IndexShards over arbitrary sub-indexes
void* ref_shards_new(int d, int threaded, int successive_ids) {
    return (faiss::Index*)new faiss::IndexShards(d, threaded != 0, successive_ids != 0);
}
int ref_shards_add(void* shards, void* sub) {
    SHIM_TRY auto* s = dynamic_cast<faiss::IndexShards*>((faiss::Index*)shards);
    FAISS_THROW_IF_NOT_MSG(s, "not an IndexShards");
    s->add_shard((faiss::Index*)sub);
    SHIM_CATCH
}

} // extern "C"
