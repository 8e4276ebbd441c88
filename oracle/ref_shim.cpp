// oracle/ref_shim.cpp -- plain-C access to the UNMODIFIED reference library (faiss v1.15.0 CPU
// path compiled from /root/reference by oracle/Makefile.ref) for ctypes.
//
// TEST INFRASTRUCTURE ONLY: loaded by tests/ and by bench.py's cpu_baseline leg.  This file
// is our own code; it includes the reference's public headers and calls its public API.
//
// Besides the oracle calls it compiles the drop-in proof: integration/faiss_amd_bridge.h, the reference-side
// binding of the backend (faiss::Index / IndexIVFInterface subclasses over the C ABI + the cloner functions), against
// the unmodified reference headers, and exposes it to the tests.  With it the reference's own callers of the hot
// path -- faiss::Clustering::train (faiss/Clustering.cpp:255-357), faiss::IndexShards (faiss/IndexShards.cpp:135-265),
// faiss::IndexShardsIVF, faiss::IndexReplicas, faiss::IndexIVF with the backend as coarse quantizer
// (faiss/IndexIVF.cpp:194,336-342) -- run unchanged on the MI355X backend.  (This library links libfaiss_amd.so.)
#include <faiss/AutoTune.h>
#include <faiss/Clustering.h>
#include <faiss/Index.h>
#include <faiss/IndexFlat.h>
#include <faiss/IndexIVF.h>
#include <faiss/IndexIVFFlat.h>
#include <faiss/IndexIVFPQ.h>
#include <faiss/IndexScalarQuantizer.h>
#include <faiss/IndexShards.h>
#include <faiss/index_factory.h>
#include <faiss/index_io.h>
#include <faiss/impl/FaissException.h>
#include <faiss/utils/distances.h>
#include <omp.h>
#include "../integration/faiss_amd_bridge.h"
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
int ref_index_set_metric_arg(void* p, float arg) {
    SHIM_TRY((faiss::Index*)p)->metric_arg = arg;
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
// (IndexIVFInterface: the reference's IndexIVF and the bridge's AmdIndexIVF alike -- nprobe / cp are its members)
static faiss::IndexIVFInterface* ivfi(void* p) {
    auto* i = dynamic_cast<faiss::IndexIVFInterface*>((faiss::Index*)p);
    FAISS_THROW_IF_NOT_MSG(i, "not an IVF index");
    return i;
}
int ref_ivf_set_nprobe(void* p, int nprobe) {
    SHIM_TRY ivfi(p)->nprobe = nprobe;
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
// n entries (ids + code_size-byte codes, the reference's plain [entry][code_size] payload) appended to list `list_no` of
// a trained IndexIVF: how a CPU index of nb = 10M-100M rows is filled with exactly the lists the GPU index holds (read
// back list by list) for the measured CPU baselines of bench.py -- invlists->add_entries is what IndexIVF::add_core and
// GpuIndexIVF::copyTo (faiss/gpu/impl/IVFBase.cu:328-344 copyInvertedListsTo) end in
int ref_ivf_add_list_entries(void* p, idx_t list_no, idx_t n, const idx_t* ids, const uint8_t* codes) {
    SHIM_TRY auto* i = ivf(p);
    FAISS_THROW_IF_NOT(i->is_trained && list_no >= 0 && (size_t)list_no < i->nlist);
    if (n > 0) {
        i->invlists->add_entries((size_t)list_no, (size_t)n, ids, codes);
        i->ntotal += n;
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

// ---- IndexIVFScalarQuantizer
void* ref_ivfsq_new(int d, int nlist, int qtype, int metric, int by_residual) {
    try {
        auto* q = new faiss::IndexFlat(d, (faiss::MetricType)metric);
        auto* i = new faiss::IndexIVFScalarQuantizer(q, d, nlist, (faiss::ScalarQuantizer::QuantizerType)qtype,
                                                     (faiss::MetricType)metric, by_residual != 0);
        i->own_fields = true;
        return i;
    } catch (std::exception& e) {
        g_err = e.what();
        return nullptr;
    }
}
int ref_ivfsq_info(void* p, int* qtype, int* by_residual, size_t* code_size, size_t* trained_size) {
    SHIM_TRY auto* i = dynamic_cast<faiss::IndexIVFScalarQuantizer*>((faiss::Index*)p);
    FAISS_THROW_IF_NOT_MSG(i, "not an IndexIVFScalarQuantizer");
    *qtype = (int)i->sq.qtype;
    *by_residual = i->by_residual ? 1 : 0;
    *code_size = i->code_size;
    *trained_size = i->sq.trained.size();
    SHIM_CATCH
}
int ref_ivfsq_get_trained(void* p, float* out) {
    SHIM_TRY auto* i = dynamic_cast<faiss::IndexIVFScalarQuantizer*>((faiss::Index*)p);
    FAISS_THROW_IF_NOT_MSG(i, "not an IndexIVFScalarQuantizer");
    memcpy(out, i->sq.trained.data(), sizeof(float) * i->sq.trained.size());
    SHIM_CATCH
}
// install a trained state (coarse centroids nlist x d + ScalarQuantizer::trained), like copyTo of the GPU index
int ref_ivfsq_set_trained(void* p, const float* centroids, const float* trained, size_t n) {
    SHIM_TRY auto* i = dynamic_cast<faiss::IndexIVFScalarQuantizer*>((faiss::Index*)p);
    FAISS_THROW_IF_NOT_MSG(i, "not an IndexIVFScalarQuantizer");
    i->quantizer->reset();
    i->quantizer->add(i->nlist, centroids);
    i->quantizer->is_trained = true;
    i->sq.trained.assign(trained, trained + n);
    i->is_trained = true;
    SHIM_CATCH
}
// ScalarQuantizer::decode of n codes (the stored values, without the list centroid)
// faiss::ScalarQuantizer(d, qtype) with the given range statistic trained on x [n][d]: its `trained` vector
int ref_sq_train(int d, int qtype, int rangestat, float rangestat_arg, idx_t n, const float* x, float* trained_out, size_t* n_out) {
    SHIM_TRY faiss::ScalarQuantizer sq(d, (faiss::ScalarQuantizer::QuantizerType)qtype);
    sq.rangestat = (faiss::ScalarQuantizer::RangeStat)rangestat;
    sq.rangestat_arg = rangestat_arg;
    sq.train(n, x);
    memcpy(trained_out, sq.trained.data(), sizeof(float) * sq.trained.size());
    *n_out = sq.trained.size();
    SHIM_CATCH
}
int ref_ivfsq_decode(void* p, idx_t n, const uint8_t* codes, float* out) {
    SHIM_TRY auto* i = dynamic_cast<faiss::IndexIVFScalarQuantizer*>((faiss::Index*)p);
    FAISS_THROW_IF_NOT_MSG(i, "not an IndexIVFScalarQuantizer");
    i->sq.decode(codes, out, n);
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
    SHIM_TRY auto* ivf = ivfi(p);
    if (niter_coarse > 0) ivf->cp.niter = niter_coarse;
    auto* pq = dynamic_cast<faiss::IndexIVFPQ*>((faiss::Index*)p);
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

// ------------------------------------------------------------ drop-in: the bridge over the C ABI
// a faiss::Index view of a backend handle the caller keeps owning
void* ref_amd_wrap(void* handle) {
    try {
        auto* h = (FaissAmdIndex*)handle;
        int nlist = 0;
        if (faiss_amd_IndexIVF_nlist(h, &nlist) == 0) {
            auto* a = new faiss::amd::AmdIndexIVF(h, (size_t)nlist);
            a->own_handle = false;
            int np = 1;
            faiss_amd_IndexIVF_nprobe(h, &np);
            a->nprobe = np;
            return (faiss::Index*)a;
        }
        return (faiss::Index*)new faiss::amd::AmdIndex(h, false);
    } catch (std::exception& e) {
        g_err = e.what();
        return nullptr;
    }
}
void* ref_amd_resources_new(int device) {
    try {
        return new faiss::amd::AmdGpuResources(device);
    } catch (std::exception& e) {
        g_err = e.what();
        return nullptr;
    }
}
void ref_amd_resources_free(void* r) {
    delete (faiss::amd::AmdGpuResources*)r;
}
// index_cpu_to_gpu / index_cpu_to_gpu_multiple / index_gpu_to_cpu of the bridge
void* ref_amd_index_cpu_to_gpu(void* res, void* cpu_index) {
    try {
        return faiss::amd::index_cpu_to_gpu((faiss::amd::AmdGpuResources*)res, (const faiss::Index*)cpu_index);
    } catch (std::exception& e) {
        g_err = e.what();
        return nullptr;
    }
}
void* ref_amd_index_cpu_to_gpu_multiple(void** res, int nres, void* cpu_index, int shard, int shard_type,
                                        int common_ivf_quantizer) {
    try {
        std::vector<faiss::amd::AmdGpuResources*> v;
        for (int i = 0; i < nres; i++) v.push_back((faiss::amd::AmdGpuResources*)res[i]);
        faiss::amd::AmdClonerOptions opt;
        opt.shard = shard != 0;
        opt.shard_type = shard_type;
        opt.common_ivf_quantizer = common_ivf_quantizer != 0;
        return faiss::amd::index_cpu_to_gpu_multiple(v, (const faiss::Index*)cpu_index, &opt);
    } catch (std::exception& e) {
        g_err = e.what();
        return nullptr;
    }
}
void* ref_amd_index_gpu_to_cpu(void* gpu_index) {
    try {
        return faiss::amd::index_gpu_to_cpu((const faiss::Index*)gpu_index);
    } catch (std::exception& e) {
        g_err = e.what();
        return nullptr;
    }
}
// a flat index of the backend as a faiss::Index (AmdIndexFlat): e.g. the caller-owned coarse quantizer of an IVF index
void* ref_amd_flat_new(void* res, int d, int metric) {
    try {
        return (faiss::Index*)new faiss::amd::AmdIndexFlat((faiss::amd::AmdGpuResources*)res, d, (faiss::MetricType)metric);
    } catch (std::exception& e) {
        g_err = e.what();
        return nullptr;
    }
}
// GpuIndexIVFFlat / IVFPQ / IVFScalarQuantizer(provider, Index* coarseQuantizer, ...) of the bridge: `quantizer` is ANY
// faiss::Index* -- a backend flat index (handed to the device side) or a host index (CPU coarse quantizer).  kind 0 IVFFlat,
// 1 IVFPQ (arg = M), 2 scalar quantizer (arg = qtype, residual encoding)
void* ref_amd_ivf_new_with_quantizer(void* res, void* quantizer, int kind, int d, int nlist, int arg, int metric, int coarse_f16,
                                     int indices_options) {
    try {
        auto* r = (faiss::amd::AmdGpuResources*)res;
        auto* q = (faiss::Index*)quantizer;
        FaissAmdGpuIndexIVFPQConfig cfg;
        memset(&cfg, 0, sizeof(cfg));
        cfg.ivf.device = -1;
        cfg.ivf.indicesOptions = indices_options;
        cfg.ivf.flat_useFloat16 = coarse_f16;
        const auto mt = (faiss::MetricType)metric;
        if (kind == 0) return (faiss::Index*)new faiss::amd::AmdIndexIVFFlat(r, q, d, (size_t)nlist, mt, &cfg.ivf);
        if (kind == 1) return (faiss::Index*)new faiss::amd::AmdIndexIVFPQ(r, q, d, (size_t)nlist, arg, 8, mt, &cfg);
        return (faiss::Index*)new faiss::amd::AmdIndexIVFScalarQuantizer(r, q, d, (size_t)nlist, (faiss::ScalarQuantizer::QuantizerType)arg,
                                                                        mt, true, &cfg.ivf);
    } catch (std::exception& e) {
        g_err = e.what();
        return nullptr;
    }
}
// GpuParameterSpace of the bridge + the reference's ParameterSpace::explore on a backend index (faiss/AutoTune.cpp:632-737):
// 1-recall@1 against `gt` [nq] as the criterion; returns the optimal operating points (perf, seconds, key strings joined by ';')
int ref_amd_autotune(void* index, idx_t nq, const float* xq, idx_t k, const idx_t* gt, int cap, int* n_ranges, int* n_pts, double* perf,
                     double* t, char* keys, int keys_cap) {
    SHIM_TRY auto* ix = (faiss::Index*)index;
    faiss::amd::AmdParameterSpace ps;
    ps.initialize(ix);
    *n_ranges = (int)ps.parameter_ranges.size();
    ps.verbose = 0;
    ps.n_experiments = 0; // try all combinations
    faiss::OneRecallAtRCriterion crit(nq, 1);
    crit.set_groundtruth(1, nullptr, gt);
    crit.nnn = k;
    faiss::OperatingPoints ops;
    ps.explore(ix, nq, xq, crit, &ops);
    std::string ks;
    int n = 0;
    for (const auto& op : ops.optimal_pts) {
        if (n >= cap) break;
        perf[n] = op.perf;
        t[n] = op.t;
        ks += op.key + ";";
        n++;
    }
    *n_pts = n;
    snprintf(keys, keys_cap, "%s", ks.c_str());
    SHIM_CATCH
}
// faiss::write_index / read_index (faiss/index_io.h; impl/index_write.cpp, impl/index_read.cpp): the checkpoint path of a GPU
// index is index_gpu_to_cpu -> write_index, and back read_index -> index_cpu_to_gpu (faiss/gpu/test/test_gpu_index_serialize.py)
int ref_write_index(void* p, const char* path) {
    SHIM_TRY faiss::write_index((const faiss::Index*)p, path);
    SHIM_CATCH
}
void* ref_read_index(const char* path) {
    try {
        return faiss::read_index(path);
    } catch (std::exception& e) {
        g_err = e.what();
        return nullptr;
    }
}
// a reference IndexIVFFlat whose coarse quantizer is ANY faiss::Index* (e.g. a backend flat index): the caller of
// quantizer->search / assign (faiss/IndexIVF.cpp:194, 336-342)
void* ref_ivfflat_with_quantizer(void* quantizer, int d, int nlist, int metric) {
    try {
        return (faiss::Index*)new faiss::IndexIVFFlat((faiss::Index*)quantizer, d, nlist, (faiss::MetricType)metric);
    } catch (std::exception& e) {
        g_err = e.what();
        return nullptr;
    }
}
int ref_index_search_nprobe(void* p, idx_t n, const float* x, idx_t k, int nprobe, float* D, idx_t* I) {
    SHIM_TRY faiss::SearchParametersIVF sp;
    sp.nprobe = nprobe;
    ((faiss::Index*)p)->search(n, x, k, D, I, &sp);
    SHIM_CATCH
}
// search restricted by a faiss::IDSelector (SearchParameters::sel) described by a few scalars, so that the CPU indexes of
// the reference and the bridge indexes can be driven with the SAME selector objects from the tests:
//   kind 0 IDSelectorRange [a, b)   1 IDSelectorBatch(nsel ids)   2 IDSelectorArray(nsel ids)   3 IDSelectorBitmap(nsel bytes)
//   4 IDSelectorAnd(IDSelectorRange [a, b), IDSelectorNot(IDSelectorBatch(nsel ids)))
//   5 a selector type the bridge does not know (id % a == b): exercises its tabulating fallback
//   6 IDSelectorXOr(IDSelectorRange [a, b), IDSelectorOr(IDSelectorBitmap(nsel bytes), IDSelectorAll)) = not in [a, b)
// negate wraps the result in IDSelectorNot.  nprobe > 0: SearchParametersIVF, else SearchParameters.
struct ModuloSelector : faiss::IDSelector {
    idx_t m, r;
    ModuloSelector(idx_t m_, idx_t r_) : m(m_), r(r_) {}
    bool is_member(idx_t id) const override {
        return id % m == r;
    }
};
int ref_index_search_sel(void* p, idx_t n, const float* x, idx_t k, int nprobe, int kind, idx_t a, idx_t b, size_t nsel,
                         const void* data, int negate, float* D, idx_t* I) {
    SHIM_TRY std::vector<std::unique_ptr<faiss::IDSelector>> own;
    auto mk = [&](faiss::IDSelector* s) {
        own.emplace_back(s);
        return s;
    };
    faiss::IDSelector* sel = nullptr;
    switch (kind) {
        case 0: sel = mk(new faiss::IDSelectorRange(a, b)); break;
        case 1: sel = mk(new faiss::IDSelectorBatch(nsel, (const idx_t*)data)); break;
        case 2: sel = mk(new faiss::IDSelectorArray(nsel, (const idx_t*)data)); break;
        case 3: sel = mk(new faiss::IDSelectorBitmap(nsel, (const uint8_t*)data)); break;
        case 4: {
            auto* r = mk(new faiss::IDSelectorRange(a, b));
            auto* bt = mk(new faiss::IDSelectorBatch(nsel, (const idx_t*)data));
            auto* nb = mk(new faiss::IDSelectorNot(bt));
            sel = mk(new faiss::IDSelectorAnd(r, nb));
            break;
        }
        case 5: sel = mk(new ModuloSelector(a, b)); break;
        case 6: {
            auto* r = mk(new faiss::IDSelectorRange(a, b));
            auto* bm = mk(new faiss::IDSelectorBitmap(nsel, (const uint8_t*)data));
            auto* all = mk(new faiss::IDSelectorAll());
            auto* o = mk(new faiss::IDSelectorOr(bm, all));
            sel = mk(new faiss::IDSelectorXOr(r, o));
            break;
        }
        default: FAISS_THROW_MSG("unknown selector kind");
    }
    if (negate) sel = mk(new faiss::IDSelectorNot(sel));
    if (nprobe > 0) {
        faiss::SearchParametersIVF sp;
        sp.nprobe = nprobe;
        sp.sel = sel;
        ((faiss::Index*)p)->search(n, x, k, D, I, &sp);
    } else {
        faiss::SearchParameters sp;
        sp.sel = sel;
        ((faiss::Index*)p)->search(n, x, k, D, I, &sp);
    }
    SHIM_CATCH
}
// IndexIVF::add_core / GpuIndexIVF::add_core with a caller-supplied list assignment (contrib/ivf_tools.py add_preassigned)
int ref_ivf_add_core(void* p, idx_t n, const float* x, const idx_t* xids, const idx_t* assign) {
    SHIM_TRY if (auto* a = dynamic_cast<faiss::amd::AmdIndexIVF*>((faiss::Index*)p)) {
        a->add_core(n, x, xids, assign);
    } else {
        ivf(p)->add_core(n, x, xids, assign);
    }
    SHIM_CATCH
}
int ref_index_assign(void* p, idx_t n, const float* x, idx_t* labels, idx_t k) {
    SHIM_TRY((faiss::Index*)p)->assign(n, x, labels, k);
    SHIM_CATCH
}
int ref_index_reconstruct_n(void* p, idx_t i0, idx_t ni, float* out) {
    SHIM_TRY((faiss::Index*)p)->reconstruct_n(i0, ni, out);
    SHIM_CATCH
}
int ref_index_compute_residual_n(void* p, idx_t n, const float* x, float* res, const idx_t* keys) {
    SHIM_TRY((faiss::Index*)p)->compute_residual_n(n, x, res, keys);
    SHIM_CATCH
}
int ref_index_type(void* p, char* out, int cap) {
    SHIM_TRY const char* name = typeid(*(faiss::Index*)p).name();
    strncpy(out, name, cap - 1);
    out[cap - 1] = 0;
    SHIM_CATCH
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
