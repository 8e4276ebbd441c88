// integration/faiss_amd_bridge.h -- the reference-side binding of the MI355X backend: what a faiss maintainer adds to
// the reference tree to make `libfaiss_amd.so` a drop-in for faiss/gpu on this path.
//
// Header-only C++ over (a) the reference's PUBLIC headers and (b) the C ABI of include/faiss_amd_c.h -- no HIP, no
// torch, no access to the backend's internals.  It provides
//   * faiss::amd::AmdIndex / AmdIndexFlat / AmdIndexIVFFlat / AmdIndexIVFPQ: faiss::Index subclasses (the IVF ones also
//     faiss::IndexIVFInterface, like faiss::gpu::GpuIndexIVF, faiss/gpu/GpuIndexIVF.h:37-40) that forward the whole
//     add/search surface the reference's callers use -- train, add, add_with_ids, search (+ SearchParameters: nprobe and
//     IDSelector, translated to the backend's device-side selectors),
//     assign, reset, reconstruct, reconstruct_n, reconstruct_batch, compute_residual[_n], search_preassigned -- so
//     faiss::IndexShards / IndexReplicas / IndexShardsIVF / Clustering / IndexIVF (as coarse quantizer) / IndexIDMap /
//     IndexPreTransform run on them unchanged;
//   * copyFrom / copyTo with the CPU index types (GpuIndexFlat.cu:125-173, GpuIndexIVFFlat.cu copyFrom/copyTo,
//     GpuIndexIVFPQ.cu:98-217), including the translation between the reference's inverted-list payloads and the
//     backend's device layout (done behind faiss_amd_IndexIVF_copy_lists / get_list_codes);
//   * index_cpu_to_gpu, index_cpu_to_gpu_multiple, index_gpu_to_cpu with the semantics of faiss/gpu/GpuCloner.cpp:
//     139-255 (one device), :325-499 (replicas, or shards with shard_type 1 / 2 / 4, optionally one common coarse
//     quantizer = IndexShardsIVF) and :43-137 (back to the CPU types).
//
// This file is compiled against the UNMODIFIED reference by oracle/Makefile.ref (test infrastructure) and exercised by
// tests/test_gpu_bridge.py; the product library does not depend on it.
#pragma once

#include <faiss/AutoTune.h>
#include <faiss/Clustering.h>
#include <faiss/Index.h>
#include <faiss/IndexFlat.h>
#include <faiss/IndexIVF.h>
#include <faiss/IndexIVFFlat.h>
#include <faiss/IndexIVFPQ.h>
#include <faiss/IndexPreTransform.h>
#include <faiss/IndexReplicas.h>
#include <faiss/IndexScalarQuantizer.h>
#include <faiss/IndexShards.h>
#include <faiss/IndexShardsIVF.h>
#include <faiss/impl/FaissAssert.h>
#include <faiss/impl/IDSelector.h>
#include <faiss/invlists/InvertedLists.h>

#include <cstring>
#include <memory>
#include <vector>

#include "../include/faiss_amd_c.h"

namespace faiss {
namespace amd {

inline void amd_check(int rc) {
    if (rc != 0) {
        FAISS_THROW_FMT("faiss_amd error %d: %s", rc, faiss_amd_get_last_error());
    }
}

/// faiss::gpu::StandardGpuResources counterpart: one stream + scratch per device
struct AmdGpuResources {
    FaissAmdGpuResources* h = nullptr;
    explicit AmdGpuResources(int device = 0) {
        amd_check(faiss_amd_StandardGpuResources_new(&h, device));
    }
    AmdGpuResources(const AmdGpuResources&) = delete;
    ~AmdGpuResources() {
        faiss_amd_StandardGpuResources_free(h);
    }
};

/// faiss::IDSelector -> the backend's selector objects.  The reference's concrete selectors (Range, Array, Batch,
/// Bitmap, All, Not, And, Or, XOr; faiss/impl/IDSelector.h:71-215) are translated structurally; any other subclass is
/// tabulated through is_member() over the labels [0, domain) when the index has such a domain (flat: row numbers) and
/// refused otherwise.  Owns the translated objects; `root` is what SearchParameters point at.
struct AmdSelector {
    std::vector<FaissAmdIDSelector*> owned;
    const FaissAmdIDSelector* root = nullptr;
    AmdSelector(const faiss::IDSelector* sel, idx_t domain) {
        root = build(sel, domain);
    }
    AmdSelector(const AmdSelector&) = delete;
    ~AmdSelector() {
        for (auto it = owned.rbegin(); it != owned.rend(); ++it) {
            faiss_amd_IDSelector_free(*it);
        }
    }
    const FaissAmdIDSelector* keep(FaissAmdIDSelector* s) {
        owned.push_back(s);
        return s;
    }
    const FaissAmdIDSelector* build(const faiss::IDSelector* sel, idx_t domain) {
        FAISS_THROW_IF_NOT_MSG(sel, "null IDSelector");
        FaissAmdIDSelector* out = nullptr;
        if (auto r = dynamic_cast<const faiss::IDSelectorRange*>(sel)) {
            amd_check(faiss_amd_IDSelectorRange_new(&out, r->imin, r->imax));
        } else if (auto a = dynamic_cast<const faiss::IDSelectorArray*>(sel)) {
            amd_check(faiss_amd_IDSelectorArray_new(&out, a->n, a->ids));
        } else if (auto b = dynamic_cast<const faiss::IDSelectorBatch*>(sel)) {
            std::vector<idx_t> ids(b->set.begin(), b->set.end());
            amd_check(faiss_amd_IDSelectorBatch_new(&out, ids.size(), ids.data()));
        } else if (auto m = dynamic_cast<const faiss::IDSelectorBitmap*>(sel)) {
            amd_check(faiss_amd_IDSelectorBitmap_new(&out, m->n, m->bitmap));
        } else if (dynamic_cast<const faiss::IDSelectorAll*>(sel)) {
            amd_check(faiss_amd_IDSelectorAll_new(&out));
        } else if (auto nt = dynamic_cast<const faiss::IDSelectorNot*>(sel)) {
            const FaissAmdIDSelector* inner = build(nt->sel, domain);
            amd_check(faiss_amd_IDSelectorNot_new(&out, inner));
        } else if (auto x = dynamic_cast<const faiss::IDSelectorAnd*>(sel)) {
            const FaissAmdIDSelector *l = build(x->lhs, domain), *r2 = build(x->rhs, domain);
            amd_check(faiss_amd_IDSelectorAnd_new(&out, l, r2));
        } else if (auto x = dynamic_cast<const faiss::IDSelectorOr*>(sel)) {
            const FaissAmdIDSelector *l = build(x->lhs, domain), *r2 = build(x->rhs, domain);
            amd_check(faiss_amd_IDSelectorOr_new(&out, l, r2));
        } else if (auto x = dynamic_cast<const faiss::IDSelectorXOr*>(sel)) {
            const FaissAmdIDSelector *l = build(x->lhs, domain), *r2 = build(x->rhs, domain);
            amd_check(faiss_amd_IDSelectorXOr_new(&out, l, r2));
        } else {
            FAISS_THROW_IF_NOT_MSG(
                    domain >= 0, "this IDSelector type cannot be evaluated on the device for arbitrary stored ids");
            std::vector<uint8_t> bits((size_t)(domain + 7) / 8, 0);
            for (idx_t i = 0; i < domain; i++) {
                if (sel->is_member(i)) {
                    bits[i >> 3] |= (uint8_t)(1u << (i & 7));
                }
            }
            amd_check(faiss_amd_IDSelectorBitmap_new(&out, bits.size(), bits.data()));
        }
        return keep(out);
    }
};

/// SearchParameters of one call on the backend side (sel translated, nprobe when the caller passed IVF parameters)
struct AmdSearchParams {
    std::unique_ptr<AmdSelector> sel;
    FaissAmdSearchParameters* h = nullptr;
    /// ivf: create SearchParametersIVF with this nprobe; domain: see AmdSelector
    AmdSearchParams(const faiss::IDSelector* s, idx_t domain, bool ivf, size_t nprobe) {
        if (s) {
            sel.reset(new AmdSelector(s, domain));
        }
        if (ivf) {
            amd_check(faiss_amd_SearchParametersIVF_new_with(&h, sel ? sel->root : nullptr, nprobe, 0));
        } else {
            amd_check(faiss_amd_SearchParameters_new(&h, sel ? sel->root : nullptr));
        }
    }
    AmdSearchParams(const AmdSearchParams&) = delete;
    ~AmdSearchParams() {
        faiss_amd_SearchParameters_free(h);
    }
};

/// faiss::Index over a backend handle (faiss::gpu::GpuIndex counterpart)
struct AmdIndex : faiss::Index {
    FaissAmdIndex* h;
    bool own_handle;

    explicit AmdIndex(FaissAmdIndex* handle, bool own = true)
            : faiss::Index(faiss_amd_Index_d(handle), (MetricType)faiss_amd_Index_metric_type(handle)),
              h(handle),
              own_handle(own) {
        sync();
    }
    ~AmdIndex() override {
        if (own_handle) {
            faiss_amd_Index_free(h);
        }
    }
    /// public data members callers read directly (IndexShards::syncWithSubIndexes, faiss/IndexShards.cpp:87-110)
    void sync() {
        ntotal = faiss_amd_Index_ntotal(h);
        is_trained = faiss_amd_Index_is_trained(h) != 0;
    }
    void train(idx_t n, const float* x) override {
        amd_check(faiss_amd_Index_train(h, n, x));
        sync();
    }
    void add(idx_t n, const float* x) override {
        amd_check(faiss_amd_Index_add(h, n, x));
        sync();
    }
    void add_with_ids(idx_t n, const float* x, const idx_t* xids) override {
        amd_check(faiss_amd_Index_add_with_ids(h, n, x, xids));
        sync();
    }
    /// labels an IDSelector of unknown type can be tabulated over: [0, result), -1 = arbitrary ids (IVF)
    virtual idx_t selector_domain() const {
        return -1;
    }
    /// params->sel is honoured (IndexFlat::search does, faiss/IndexFlat.cpp:36-58)
    void search(idx_t n, const float* x, idx_t k, float* distances, idx_t* labels,
                const SearchParameters* params = nullptr) const override {
        if (params && params->sel) {
            AmdSearchParams sp(params->sel, selector_domain(), false, 0);
            amd_check(faiss_amd_Index_search_with_params(h, n, x, k, sp.h, distances, labels));
            return;
        }
        amd_check(faiss_amd_Index_search(h, n, x, k, distances, labels));
    }
    void assign(idx_t n, const float* x, idx_t* labels, idx_t k = 1) const override {
        amd_check(faiss_amd_Index_assign(h, n, x, labels, k));
    }
    void reset() override {
        amd_check(faiss_amd_Index_reset(h));
        sync();
    }
    void reconstruct(idx_t key, float* recons) const override {
        amd_check(faiss_amd_Index_reconstruct(h, key, recons));
    }
    void reconstruct_n(idx_t i0, idx_t ni, float* recons) const override {
        amd_check(faiss_amd_Index_reconstruct_n(h, i0, ni, recons));
    }
    void reconstruct_batch(idx_t n, const idx_t* keys, float* recons) const override {
        amd_check(faiss_amd_Index_reconstruct_batch(h, n, keys, recons));
    }
    void compute_residual(const float* x, float* residual, idx_t key) const override {
        amd_check(faiss_amd_Index_compute_residual(h, x, residual, key));
    }
    void compute_residual_n(idx_t n, const float* xs, float* residuals, const idx_t* keys) const override {
        amd_check(faiss_amd_Index_compute_residual_n(h, n, xs, residuals, keys));
    }
};

/// faiss::gpu::GpuIndexFlat counterpart
struct AmdIndexFlat : AmdIndex {
    static FaissAmdIndex* make(AmdGpuResources* res, int d, MetricType metric) {
        FAISS_THROW_IF_NOT_MSG(
                metric == METRIC_L2 || metric == METRIC_INNER_PRODUCT || metric == METRIC_L1 || metric == METRIC_Linf ||
                        metric == METRIC_Lp || metric == METRIC_Canberra || metric == METRIC_BrayCurtis ||
                        metric == METRIC_JensenShannon || metric == METRIC_Jaccard,
                "AmdIndexFlat: unsupported metric");
        FaissAmdIndex* handle = nullptr;
        amd_check(faiss_amd_GpuIndexFlat_new(&handle, res->h, d, (FaissAmdMetricType)metric));
        return handle;
    }
    AmdIndexFlat(AmdGpuResources* res, int d, MetricType metric = METRIC_L2) : AmdIndex(make(res, d, metric)) {}
    idx_t selector_domain() const override {
        return ntotal; // labels are row numbers
    }
    /// metric_arg is a public data member callers assign to (index.metric_arg = 3 for METRIC_Lp): it travels with the call
    void search(idx_t n, const float* x, idx_t k, float* distances, idx_t* labels,
                const SearchParameters* params = nullptr) const override {
        if (metric_type == METRIC_Lp) {
            amd_check(faiss_amd_Index_set_metric_arg(h, metric_arg));
        }
        AmdIndex::search(n, x, k, distances, labels, params);
    }
    /// GpuIndexFlat(resources, const IndexFlat*) / copyFrom (faiss/gpu/GpuIndexFlat.cu:125-148)
    AmdIndexFlat(AmdGpuResources* res, const faiss::IndexFlat* index) : AmdIndex(make(res, index->d, index->metric_type)) {
        copyFrom(index);
    }
    void copyFrom(const faiss::IndexFlat* index) {
        FAISS_THROW_IF_NOT(index->d == d && index->metric_type == metric_type);
        metric_arg = index->metric_arg;
        reset();
        if (index->ntotal > 0) {
            add(index->ntotal, index->get_xb());
        }
    }
    /// faiss/gpu/GpuIndexFlat.cu:150-173
    void copyTo(faiss::IndexFlat* index) const {
        FAISS_THROW_IF_NOT(index->d == d);
        index->metric_type = metric_type;
        index->metric_arg = metric_arg;
        index->reset();
        const idx_t bs = 1 << 18;
        std::vector<float> buf((size_t)std::min<idx_t>(bs, std::max<idx_t>(ntotal, 1)) * d);
        for (idx_t i0 = 0; i0 < ntotal; i0 += bs) {
            const idx_t ni = std::min(bs, ntotal - i0);
            reconstruct_n(i0, ni, buf.data());
            index->add(ni, buf.data());
        }
    }
};

/// the coarse quantizer of an IVF backend index seen as a faiss::Index (IndexIVFInterface::quantizer): what
/// IndexShardsIVF::train installs centroids through and what callers search / reconstruct
struct AmdQuantizerView : faiss::Index {
    FaissAmdIndex* ivf;
    size_t nlist;
    AmdQuantizerView(FaissAmdIndex* ivf_handle, int d, MetricType metric, size_t nlist_in)
            : faiss::Index(d, metric), ivf(ivf_handle), nlist(nlist_in) {
        is_trained = true;
    }
    void add(idx_t n, const float* x) override {
        FAISS_THROW_IF_NOT_MSG((size_t)n == nlist, "the coarse quantizer of an IVF index takes exactly nlist centroids");
        amd_check(faiss_amd_IndexIVF_copy_centroids(ivf, x));
        ntotal = n;
    }
    void search(idx_t n, const float* x, idx_t k, float* distances, idx_t* labels,
                const SearchParameters* params = nullptr) const override {
        FAISS_THROW_IF_NOT_MSG(!params, "search params not supported for this index");
        amd_check(faiss_amd_IndexIVF_quantizer_search(ivf, n, x, k, distances, labels));
    }
    void reset() override {
        ntotal = 0; // centroids are replaced by the next add()
    }
    void reconstruct_n(idx_t i0, idx_t ni, float* recons) const override {
        FAISS_THROW_IF_NOT(i0 >= 0 && (size_t)(i0 + ni) <= nlist);
        std::vector<float> all(nlist * d);
        amd_check(faiss_amd_IndexIVF_get_centroids(ivf, all.data()));
        memcpy(recons, all.data() + (size_t)i0 * d, sizeof(float) * (size_t)ni * d);
    }
    void reconstruct(idx_t key, float* recons) const override {
        reconstruct_n(key, 1, recons);
    }
};

/// faiss::gpu::GpuIndexIVF counterpart
struct AmdIndexIVF : AmdIndex, faiss::IndexIVFInterface {
    AmdQuantizerView quantizer_view;
    /// GpuIndexIVF(provider, Index* coarseQuantizer, ...) (faiss/gpu/GpuIndexIVF.cu:41-70): the caller's coarse quantizer, NOT
    /// owned (own_fields = false).  A flat index of this backend (AmdIndexFlat) is handed to the device side as it is
    /// (faiss_amd_GpuIndexIVF*_new_with_quantizer); ANY OTHER faiss::Index -- IndexFlat, IndexHNSWFlat, ... on the host -- is a
    /// "CPU coarse quantizer" (GpuIndexIVFConfig::allowCpuCoarseQuantizer, faiss/gpu/impl/IVFBase.cu:526-546): its search / assign
    /// run on the host and feed search_preassigned / add_core; the device keeps a copy of its reconstructed centroids for the
    /// residuals (IVFPQ, scalar quantizer), like the reference's ivfCentroids_ (IVFBase.cu:480-507 updateQuantizer).
    faiss::Index* user_quantizer = nullptr;
    bool cpu_coarse = false;

    /// backend handle of a caller's quantizer when it is a flat index of this backend, else null
    static FaissAmdIndex* native_quantizer(faiss::Index* q) {
        auto* f = dynamic_cast<AmdIndexFlat*>(q);
        return f ? f->h : nullptr;
    }

    AmdIndexIVF(FaissAmdIndex* handle, size_t nlist_in, faiss::Index* coarse_quantizer = nullptr)
            : AmdIndex(handle),
              faiss::IndexIVFInterface(nullptr, nlist_in),
              quantizer_view(handle, d, metric_type, nlist_in) {
        quantizer = &quantizer_view;
        own_fields = false;
        if (coarse_quantizer) {
            FAISS_THROW_IF_NOT_MSG(coarse_quantizer->d == d, "the coarse quantizer's dimension differs from the index's");
            user_quantizer = coarse_quantizer;
            quantizer = coarse_quantizer;
            cpu_coarse = native_quantizer(coarse_quantizer) == nullptr;
            if (cpu_coarse && coarse_quantizer->is_trained && coarse_quantizer->ntotal == (idx_t)nlist) push_cpu_centroids_();
            sync();
        }
        refresh_quantizer_();
    }
    /// the device's copy of a CPU quantizer's centroids (what the residuals are taken against)
    void push_cpu_centroids_() {
        std::vector<float> c(nlist * (size_t)d);
        quantizer->reconstruct_n(0, nlist, c.data());
        amd_check(faiss_amd_IndexIVF_copy_centroids(h, c.data()));
    }
    void refresh_quantizer_() {
        // (is the coarse quantizer in place? get_centroids fails cleanly when it is not)
        std::vector<float> c(nlist * (size_t)d);
        quantizer_view.ntotal = faiss_amd_IndexIVF_get_centroids(h, c.data()) == 0 ? (idx_t)nlist : 0;
        if (auto* f = dynamic_cast<AmdIndex*>(user_quantizer)) f->sync(); // (a native quantizer is filled on the device side)
    }
    void add(idx_t n, const float* x) override {
        if (!cpu_coarse) return AmdIndex::add(n, x);
        add_with_ids(n, x, nullptr);
    }
    void add_with_ids(idx_t n, const float* x, const idx_t* xids) override {
        if (!cpu_coarse) return AmdIndex::add_with_ids(n, x, xids);
        FAISS_THROW_IF_NOT_MSG(is_trained, "index must be trained before adding vectors");
        std::vector<idx_t> a((size_t)n);
        quantizer->assign(n, x, a.data()); // IndexIVF::add_with_ids: quantizer->assign, then add_core (faiss/IndexIVF.cpp:194-215)
        add_core(n, x, xids, a.data());
    }
    void train(idx_t n, const float* x) override {
        if (cpu_coarse) {
            if (quantizer->ntotal != (idx_t)nlist) {
                // Level1Quantizer::train_q1 (faiss/IndexIVF.cpp:59-127, quantizer_trains_alone = 0) = GpuIndexIVF::trainQuantizer_
                // (faiss/gpu/GpuIndexIVF.cu:508-538): k-means with the quantizer itself as the assignment index
                faiss::Clustering clus(d, nlist, cp);
                quantizer->reset();
                clus.train(n, x, *quantizer);
                quantizer->is_trained = true;
                FAISS_THROW_IF_NOT(quantizer->ntotal == (idx_t)nlist);
            }
            push_cpu_centroids_();
        }
        // GpuIndexIVF::cp (Level1Quantizer::cp, faiss/IndexIVF.h:60): the clustering parameters of the coarse quantizer
        FaissAmdClusteringParameters p;
        faiss_amd_ClusteringParameters_init(&p);
        p.niter = cp.niter;
        p.nredo = cp.nredo;
        p.verbose = cp.verbose ? 1 : 0;
        p.spherical = cp.spherical ? 1 : 0;
        p.int_centroids = cp.int_centroids ? 1 : 0;
        p.update_index = cp.update_index ? 1 : 0;
        p.frozen_centroids = cp.frozen_centroids ? 1 : 0;
        p.min_points_per_centroid = cp.min_points_per_centroid;
        p.max_points_per_centroid = cp.max_points_per_centroid;
        p.seed = cp.seed;
        amd_check(faiss_amd_IndexIVF_set_clustering_params(h, &p));
        AmdIndex::train(n, x);
        refresh_quantizer_();
    }
    /// GpuIndexIVFFlat / IVFPQ / IVFScalarQuantizer::reserveMemory, reclaimMemory; GpuIndexIVF::updateQuantizer
    void reserveMemory(size_t numVecs) {
        amd_check(faiss_amd_GpuIndexIVF_reserveMemory(h, numVecs));
    }
    size_t reclaimMemory() {
        size_t bytes = 0;
        amd_check(faiss_amd_GpuIndexIVF_reclaimMemory(h, &bytes));
        return bytes;
    }
    void updateQuantizer() {
        if (cpu_coarse && quantizer->ntotal == (idx_t)nlist) push_cpu_centroids_();
        amd_check(faiss_amd_GpuIndexIVF_updateQuantizer(h));
        sync();
        refresh_quantizer_();
    }
    /// GpuIndexIVF::add_core (faiss/gpu/GpuIndexIVF.h:84-95): the inverted list of every vector comes from the caller
    /// (contrib/ivf_tools.py add_preassigned; IndexIVF::add_core has the same signature, faiss/IndexIVF.h:261-266)
    void add_core(idx_t n, const float* x, const idx_t* xids, const idx_t* precomputed_idx,
                  void* inverted_list_context = nullptr) {
        FAISS_THROW_IF_NOT_MSG(inverted_list_context == nullptr, "add_core does not support inverted_list_context");
        amd_check(faiss_amd_GpuIndexIVF_add_core(h, n, x, xids, precomputed_idx));
        sync();
    }
    /// nprobe is a public data member callers assign to (index.nprobe = 32): it travels with every call
    void search(idx_t n, const float* x, idx_t k, float* distances, idx_t* labels,
                const SearchParameters* params = nullptr) const override {
        size_t np = nprobe;
        const faiss::IDSelector* sel = nullptr;
        if (params) {
            auto ivf_params = dynamic_cast<const SearchParametersIVF*>(params);
            FAISS_THROW_IF_NOT_MSG(ivf_params, "IndexIVF params have incorrect type");
            FAISS_THROW_IF_NOT_MSG(ivf_params->max_codes == 0, "max_codes is not supported (faiss/gpu/GpuIndexIVF.cu:372)");
            np = ivf_params->nprobe;
            sel = ivf_params->sel; // on the stored ids, as IndexIVF::search applies it (faiss/IndexIVF.cpp scan_codes)
        }
        AmdSearchParams sp(sel, -1, true, np);
        if (cpu_coarse) {
            // CPU coarse quantizer (faiss/gpu/impl/IVFBase.cu:549-590): its search on the host, the lists on the device
            np = std::min(np, nlist);
            std::vector<float> cd((size_t)n * np);
            std::vector<idx_t> ci((size_t)n * np);
            quantizer->search(n, x, (idx_t)np, cd.data(), ci.data());
            AmdSearchParams sp2(sel, -1, true, np);
            amd_check(faiss_amd_GpuIndexIVF_search_preassigned_with_params(h, n, x, k, ci.data(), cd.data(), sp2.h, distances,
                                                                           labels));
            return;
        }
        amd_check(faiss_amd_Index_search_with_params(h, n, x, k, sp.h, distances, labels));
    }
    void search_preassigned(idx_t n, const float* x, idx_t k, const idx_t* assign, const float* centroid_dis,
                            float* distances, idx_t* labels, bool store_pairs,
                            const IVFSearchParameters* params = nullptr, IndexIVFStats* stats = nullptr) const override {
        FAISS_THROW_IF_NOT_MSG(!store_pairs, "store_pairs is not supported (faiss/gpu/GpuIndexIVF.cu:424-426)");
        FAISS_THROW_IF_NOT_MSG(!stats, "IndexIVFStats are not supported");
        const size_t np = params ? params->nprobe : nprobe; // assign / centroid_dis are [n][np]
        AmdSearchParams sp(params ? params->sel : nullptr, -1, true, np);
        amd_check(faiss_amd_GpuIndexIVF_search_preassigned_with_params(h, n, x, k, assign, centroid_dis, sp.h, distances,
                                                                       labels));
    }
    void range_search_preassigned(idx_t, const float*, float, const idx_t*, const float*, RangeSearchResult*, bool,
                                  const IVFSearchParameters*, IndexIVFStats*) const override {
        FAISS_THROW_MSG("range search is not implemented (faiss/gpu/GpuIndexIVF.cu:490-502 neither)");
    }

    /// copyFrom(const IndexIVF*): coarse quantizer + inverted lists (faiss/gpu/GpuIndexIVF.cu:139-230,
    /// impl/IVFBase.cu:328-390 copyInvertedListsFrom).  The derived class installs its own trained state first.
    void copy_lists_from(const faiss::IndexIVF* index) {
        FAISS_THROW_IF_NOT(index->nlist == nlist && index->d == d);
        const InvertedLists* il = index->invlists;
        const size_t cs = il ? il->code_size : 0;
        std::vector<uint32_t> sizes(nlist, 0);
        size_t total = 0;
        for (size_t l = 0; il && l < nlist; l++) {
            sizes[l] = (uint32_t)il->list_size(l);
            total += sizes[l];
        }
        std::vector<uint8_t> codes(total * cs);
        std::vector<idx_t> ids(total);
        size_t off = 0;
        for (size_t l = 0; il && l < nlist; l++) {
            if (!sizes[l]) continue;
            InvertedLists::ScopedCodes sc(il, l);
            InvertedLists::ScopedIds si(il, l);
            memcpy(codes.data() + off * cs, sc.get(), (size_t)sizes[l] * cs);
            memcpy(ids.data() + off, si.get(), (size_t)sizes[l] * sizeof(idx_t));
            off += sizes[l];
        }
        amd_check(faiss_amd_IndexIVF_copy_lists(h, sizes.data(), codes.data(), ids.data()));
        nprobe = index->nprobe;
        sync();
    }
    void copy_quantizer_from(const faiss::IndexIVF* index) {
        FAISS_THROW_IF_NOT_MSG(index->quantizer && index->quantizer->ntotal == (idx_t)nlist, "untrained coarse quantizer");
        std::vector<float> c(nlist * (size_t)d);
        index->quantizer->reconstruct_n(0, nlist, c.data());
        if (cpu_coarse) { // the caller's host-side quantizer takes the centroids too (GpuIndexIVF::copyFrom, GpuIndexIVF.cu:139-230)
            quantizer->reset();
            quantizer->add(nlist, c.data());
        }
        amd_check(faiss_amd_IndexIVF_copy_centroids(h, c.data()));
        quantizer_view.ntotal = nlist;
        refresh_quantizer_();
    }
    /// copyTo(IndexIVF*): a flat CPU quantizer holding our centroids + ArrayInvertedLists with our lists
    /// (faiss/gpu/GpuIndexIVF.cu:232-268, impl/IVFBase.cu:346-390 copyInvertedListsTo)
    void copy_to_ivf(faiss::IndexIVF* index) const {
        FAISS_THROW_IF_NOT(index->nlist == nlist && index->d == d);
        index->metric_type = metric_type;
        index->nprobe = nprobe;
        if (index->own_fields) delete index->quantizer;
        auto* q = new faiss::IndexFlat(d, metric_type);
        index->quantizer = q;
        index->own_fields = true;
        index->quantizer_trains_alone = 0;
        if (quantizer_view.ntotal == (idx_t)nlist) {
            std::vector<float> c(nlist * (size_t)d);
            amd_check(faiss_amd_IndexIVF_get_centroids(h, c.data()));
            q->add(nlist, c.data());
        }
        auto* il = new faiss::ArrayInvertedLists(nlist, index->code_size);
        index->replace_invlists(il, true);
        size_t cs = 0;
        amd_check(faiss_amd_IndexIVF_code_size(h, &cs));
        FAISS_THROW_IF_NOT(cs == index->code_size);
        idx_t stored = 0;
        std::vector<uint8_t> codes;
        std::vector<idx_t> ids;
        for (size_t l = 0; l < nlist; l++) {
            size_t n = 0;
            amd_check(faiss_amd_IndexIVF_get_list_size(h, l, &n));
            if (!n) continue;
            codes.resize(n * cs);
            ids.resize(n);
            amd_check(faiss_amd_IndexIVF_get_list_codes(h, l, codes.data()));
            amd_check(faiss_amd_IndexIVF_get_list_ids(h, l, ids.data()));
            il->add_entries(l, n, ids.data(), codes.data());
            stored += n;
        }
        index->ntotal = stored;
        index->is_trained = is_trained;
    }
};

/// faiss::gpu::GpuIndexIVFFlat counterpart
struct AmdIndexIVFFlat : AmdIndexIVF {
    static FaissAmdIndex* make(AmdGpuResources* res, int d, size_t nlist, MetricType metric) {
        FaissAmdIndex* handle = nullptr;
        amd_check(faiss_amd_GpuIndexIVFFlat_new(&handle, res->h, d, (int)nlist, (FaissAmdMetricType)metric));
        return handle;
    }
    AmdIndexIVFFlat(AmdGpuResources* res, int d, size_t nlist, MetricType metric = METRIC_L2)
            : AmdIndexIVF(make(res, d, nlist, metric), nlist) {}
    /// GpuIndexIVFFlat(provider, Index* coarseQuantizer, dims, nlist, metric, config) (faiss/gpu/GpuIndexIVFFlat.h:49-56)
    static FaissAmdIndex* make_q(AmdGpuResources* res, faiss::Index* q, int d, size_t nlist, MetricType metric,
                                 const FaissAmdGpuIndexIVFConfig* config) {
        FaissAmdIndex* handle = nullptr;
        amd_check(faiss_amd_GpuIndexIVFFlat_new_with_quantizer(&handle, res->h, native_quantizer(q), d, (int)nlist,
                                                               (FaissAmdMetricType)metric, config));
        return handle;
    }
    AmdIndexIVFFlat(AmdGpuResources* res, faiss::Index* coarseQuantizer, int d, size_t nlist, MetricType metric = METRIC_L2,
                    const FaissAmdGpuIndexIVFConfig* config = nullptr)
            : AmdIndexIVF(make_q(res, coarseQuantizer, d, nlist, metric, config), nlist, coarseQuantizer) {}
    AmdIndexIVFFlat(AmdGpuResources* res, const faiss::IndexIVFFlat* index)
            : AmdIndexIVF(make(res, index->d, index->nlist, index->metric_type), index->nlist) {
        copyFrom(index);
    }
    void copyFrom(const faiss::IndexIVFFlat* index) {
        reset();
        nprobe = index->nprobe;
        if (index->quantizer->ntotal != (idx_t)nlist) return; // untrained: nothing to copy
        copy_quantizer_from(index);
        copy_lists_from(index);
    }
    void copyTo(faiss::IndexIVFFlat* index) const {
        copy_to_ivf(index);
    }
};

/// faiss::gpu::GpuIndexIVFPQ counterpart
struct AmdIndexIVFPQ : AmdIndexIVF {
    int M, nbits;
    static FaissAmdIndex* make(AmdGpuResources* res, int d, size_t nlist, int M, int nbits, MetricType metric) {
        FaissAmdIndex* handle = nullptr;
        amd_check(faiss_amd_GpuIndexIVFPQ_new(&handle, res->h, d, (int)nlist, M, nbits, (FaissAmdMetricType)metric));
        return handle;
    }
    AmdIndexIVFPQ(AmdGpuResources* res, int d, size_t nlist, int M_, int nbits_, MetricType metric = METRIC_L2)
            : AmdIndexIVF(make(res, d, nlist, M_, nbits_, metric), nlist), M(M_), nbits(nbits_) {}
    /// GpuIndexIVFPQ(provider, Index* coarseQuantizer, dims, nlist, subQuantizers, bitsPerCode, metric, config) (GpuIndexIVFPQ.h:70-79)
    static FaissAmdIndex* make_q(AmdGpuResources* res, faiss::Index* q, int d, size_t nlist, int M, int nbits, MetricType metric,
                                 const FaissAmdGpuIndexIVFPQConfig* config) {
        FaissAmdIndex* handle = nullptr;
        amd_check(faiss_amd_GpuIndexIVFPQ_new_with_quantizer(&handle, res->h, native_quantizer(q), d, (int)nlist, M, nbits,
                                                             (FaissAmdMetricType)metric, config));
        return handle;
    }
    AmdIndexIVFPQ(AmdGpuResources* res, faiss::Index* coarseQuantizer, int d, size_t nlist, int M_, int nbits_,
                  MetricType metric = METRIC_L2, const FaissAmdGpuIndexIVFPQConfig* config = nullptr)
            : AmdIndexIVF(make_q(res, coarseQuantizer, d, nlist, M_, nbits_, metric, config), nlist, coarseQuantizer),
              M(M_),
              nbits(nbits_) {}
    /// GpuIndexIVFPQ(resources, const IndexIVFPQ*) (faiss/gpu/GpuIndexIVFPQ.cu:48-62, copyFrom :98-168)
    AmdIndexIVFPQ(AmdGpuResources* res, const faiss::IndexIVFPQ* index)
            : AmdIndexIVF(make(res, index->d, index->nlist, (int)index->pq.M, (int)index->pq.nbits, index->metric_type),
                          index->nlist),
              M((int)index->pq.M),
              nbits((int)index->pq.nbits) {
        copyFrom(index);
    }
    void copyFrom(const faiss::IndexIVFPQ* index) {
        // the restrictions of the reference GPU index (faiss/gpu/GpuIndexIVFPQ.cu:118-130)
        FAISS_THROW_IF_NOT_MSG(index->by_residual, "only by_residual IVFPQ indexes are supported");
        FAISS_THROW_IF_NOT_MSG(index->polysemous_ht == 0, "polysemous codes are not supported");
        FAISS_THROW_IF_NOT((int)index->pq.M == M && (int)index->pq.nbits == nbits);
        reset();
        nprobe = index->nprobe;
        if (!index->is_trained) return;
        copy_quantizer_from(index);
        amd_check(faiss_amd_IndexIVFPQ_copy_pq_centroids(h, index->pq.centroids.data()));
        copy_lists_from(index);
    }
    /// faiss/gpu/GpuIndexIVFPQ.cu:170-217
    void copyTo(faiss::IndexIVFPQ* index) const {
        index->pq = faiss::ProductQuantizer(d, M, nbits);
        index->code_size = index->pq.code_size;
        index->by_residual = true;
        index->use_precomputed_table = 0;
        index->polysemous_ht = 0;
        if (is_trained) {
            amd_check(faiss_amd_IndexIVFPQ_get_pq_centroids(h, index->pq.centroids.data()));
        }
        copy_to_ivf(index);
        if (is_trained) {
            index->precompute_table();
        }
    }
};

/// faiss::gpu::GpuIndexIVFScalarQuantizer counterpart (faiss/gpu/GpuIndexIVFScalarQuantizer.h:27-131)
struct AmdIndexIVFScalarQuantizer : AmdIndexIVF {
    faiss::ScalarQuantizer sq; ///< like the reference: the quantizer parameters, mirrored on this side
    bool by_residual;
    static FaissAmdIndex* make(AmdGpuResources* res, int d, size_t nlist, int qtype, MetricType metric, bool by_residual) {
        FaissAmdIndex* handle = nullptr;
        amd_check(faiss_amd_GpuIndexIVFScalarQuantizer_new(&handle, res->h, d, (int)nlist, qtype, (FaissAmdMetricType)metric,
                                                           by_residual ? 1 : 0));
        return handle;
    }
    AmdIndexIVFScalarQuantizer(AmdGpuResources* res, int d, size_t nlist, faiss::ScalarQuantizer::QuantizerType qtype,
                               MetricType metric = METRIC_L2, bool encodeResidual = true)
            : AmdIndexIVF(make(res, d, nlist, (int)qtype, metric, encodeResidual), nlist),
              sq(d, qtype),
              by_residual(encodeResidual) {}
    /// GpuIndexIVFScalarQuantizer(provider, Index* coarseQuantizer, dims, nlist, qtype, metric, encodeResidual, config)
    /// (faiss/gpu/GpuIndexIVFScalarQuantizer.h:47-55)
    static FaissAmdIndex* make_q(AmdGpuResources* res, faiss::Index* q, int d, size_t nlist, int qtype, MetricType metric,
                                 bool by_residual, const FaissAmdGpuIndexIVFConfig* config) {
        FaissAmdIndex* handle = nullptr;
        amd_check(faiss_amd_GpuIndexIVFScalarQuantizer_new_with_quantizer(&handle, res->h, native_quantizer(q), d, (int)nlist, qtype,
                                                                          (FaissAmdMetricType)metric, by_residual ? 1 : 0, config));
        return handle;
    }
    AmdIndexIVFScalarQuantizer(AmdGpuResources* res, faiss::Index* coarseQuantizer, int d, size_t nlist,
                               faiss::ScalarQuantizer::QuantizerType qtype, MetricType metric = METRIC_L2, bool encodeResidual = true,
                               const FaissAmdGpuIndexIVFConfig* config = nullptr)
            : AmdIndexIVF(make_q(res, coarseQuantizer, d, nlist, (int)qtype, metric, encodeResidual, config), nlist, coarseQuantizer),
              sq(d, qtype),
              by_residual(encodeResidual) {}
    /// GpuIndexIVFScalarQuantizer(resources, const IndexIVFScalarQuantizer*) (GpuIndexIVFScalarQuantizer.cu:27-43)
    AmdIndexIVFScalarQuantizer(AmdGpuResources* res, const faiss::IndexIVFScalarQuantizer* index)
            : AmdIndexIVF(make(res, index->d, index->nlist, (int)index->sq.qtype, index->metric_type, index->by_residual),
                          index->nlist),
              sq(index->sq),
              by_residual(index->by_residual) {
        copyFrom(index);
    }
    /// GpuIndexIVFScalarQuantizer.cu:96-140
    void copyFrom(const faiss::IndexIVFScalarQuantizer* index) {
        FAISS_THROW_IF_NOT(index->sq.qtype == sq.qtype && index->by_residual == by_residual);
        reset();
        nprobe = index->nprobe;
        sq = index->sq;
        if (!index->is_trained) return;
        copy_quantizer_from(index);
        if (!sq.trained.empty()) amd_check(faiss_amd_IndexIVFSQ_copy_trained(h, sq.trained.data(), sq.trained.size()));
        is_trained = faiss_amd_Index_is_trained(h) != 0;
        copy_lists_from(index);
    }
    /// GpuIndexIVFScalarQuantizer.cu:142-160
    void copyTo(faiss::IndexIVFScalarQuantizer* index) const {
        size_t nt = 0;
        amd_check(faiss_amd_IndexIVFSQ_info(h, nullptr, nullptr, nullptr, &nt));
        faiss::ScalarQuantizer q(d, sq.qtype);
        q.rangestat = sq.rangestat;
        q.rangestat_arg = sq.rangestat_arg;
        q.trained.resize(nt);
        if (nt) amd_check(faiss_amd_IndexIVFSQ_get_trained(h, q.trained.data()));
        index->sq = q;
        index->code_size = q.code_size;
        index->by_residual = by_residual;
        copy_to_ivf(index);
    }
    void train(idx_t n, const float* x) override {
        AmdIndexIVF::train(n, x);
        size_t nt = 0;
        amd_check(faiss_amd_IndexIVFSQ_info(h, nullptr, nullptr, nullptr, &nt));
        sq.trained.resize(nt);
        if (nt) amd_check(faiss_amd_IndexIVFSQ_get_trained(h, sq.trained.data()));
    }
};

// ------------------------------------------------------------------ GpuParameterSpace (faiss/gpu/GpuAutoTune.h, GpuAutoTune.cpp:33-114)
/// The tunable parameters of backend indexes for the reference's own AutoTune machinery: `initialize` lists them (nprobe in powers
/// of two below nlist and up to the k-selection limit, through IndexPreTransform / IndexReplicas / IndexShards like the
/// reference), `set_index_parameter` applies one, and the inherited `ParameterSpace::explore` (faiss/AutoTune.cpp:632-737) then
/// measures operating points on the backend unchanged.
struct AmdParameterSpace : faiss::ParameterSpace {
    void initialize(const faiss::Index* index) override {
        if (auto* pt = dynamic_cast<const faiss::IndexPreTransform*>(index)) {
            initialize(pt->index);
            return;
        }
        if (dynamic_cast<const faiss::IndexShardsIVF*>(index)) {
            faiss::ParameterSpace::initialize(index);
            return;
        }
        if (auto* rep = dynamic_cast<const faiss::IndexReplicas*>(index)) {
            if (rep->count() == 0) return;
            index = rep->at(0);
        }
        if (auto* sh = dynamic_cast<const faiss::IndexShards*>(index)) {
            if (sh->count() == 0) return;
            index = sh->at(0);
        }
        if (auto* ivf = dynamic_cast<const AmdIndexIVF*>(index)) {
            faiss::ParameterRange& pr = add_range("nprobe");
            for (int i = 0; i < 12; i++) {
                const size_t np = (size_t)1 << i;
                if (np >= ivf->nlist || np > 2048) break; // (getMaxKSelection: 2048, faiss/gpu/impl/IndexUtils.cu:28-42)
                pr.values.push_back((double)np);
            }
            // a host-side coarse quantizer (CPU coarse quantizer mode) brings its own parameters, prefixed like the reference's
            if (ivf->cpu_coarse) {
                faiss::ParameterSpace qs;
                qs.initialize(ivf->quantizer);
                for (const faiss::ParameterRange& q : qs.parameter_ranges) add_range("quantizer_" + q.name).values = q.values;
            }
        }
    }
    void set_index_parameter(faiss::Index* index, const std::string& name, double val) const override {
        if (auto* rep = dynamic_cast<faiss::IndexReplicas*>(index)) {
            for (int i = 0; i < rep->count(); i++) set_index_parameter(rep->at(i), name, val);
            return;
        }
        if (auto* ivf = dynamic_cast<AmdIndexIVF*>(index)) {
            if (name == "nprobe") {
                ivf->nprobe = (size_t)val;
                return;
            }
            if (name.rfind("quantizer_", 0) == 0 && ivf->cpu_coarse) {
                faiss::ParameterSpace().set_index_parameter(ivf->quantizer, name.substr(strlen("quantizer_")), val);
                return;
            }
            amd_check(faiss_amd_GpuParameterSpace_set_index_parameter(ivf->h, name.c_str(), val)); // (use_precomputed_table ...)
            return;
        }
        faiss::ParameterSpace::set_index_parameter(index, name, val);
    }
};

// ------------------------------------------------------------------ cloners (faiss/gpu/GpuCloner.cpp)
struct AmdClonerOptions {
    bool shard = false;               ///< GpuMultipleClonerOptions::shard: shards instead of replicas
    int shard_type = 1;               ///< IndexIVF::copy_subset_to subset type (1 = id mod n, 2 = id range, 4 = lists)
    bool common_ivf_quantizer = false; ///< one coarse quantizer in front of the shards (IndexShardsIVF)
};

/// index_cpu_to_gpu (GpuCloner.cpp:139-255), for the three index types of this backend
inline faiss::Index* index_cpu_to_gpu(AmdGpuResources* res, const faiss::Index* index) {
    if (auto ifl = dynamic_cast<const faiss::IndexFlat*>(index)) {
        return new AmdIndexFlat(res, ifl);
    } else if (auto ivf = dynamic_cast<const faiss::IndexIVFFlat*>(index)) {
        return new AmdIndexIVFFlat(res, ivf);
    } else if (auto ipq = dynamic_cast<const faiss::IndexIVFPQ*>(index)) {
        return new AmdIndexIVFPQ(res, ipq);
    } else if (auto isq = dynamic_cast<const faiss::IndexIVFScalarQuantizer*>(index)) {
        return new AmdIndexIVFScalarQuantizer(res, isq);
    }
    FAISS_THROW_MSG("This index type is not implemented on the MI355X backend.");
}

/// index_gpu_to_cpu (GpuCloner.cpp:43-137): backend index (or IndexShards / IndexReplicas of them) -> CPU index
inline faiss::Index* index_gpu_to_cpu(const faiss::Index* index) {
    if (auto ifl = dynamic_cast<const AmdIndexFlat*>(index)) {
        auto* res = new faiss::IndexFlat(ifl->d, ifl->metric_type);
        ifl->copyTo(res);
        return res;
    } else if (auto ivf = dynamic_cast<const AmdIndexIVFFlat*>(index)) {
        auto* res = new faiss::IndexIVFFlat(new faiss::IndexFlat(ivf->d, ivf->metric_type), ivf->d, ivf->nlist,
                                            ivf->metric_type);
        res->own_fields = true;
        ivf->copyTo(res);
        return res;
    } else if (auto ipq = dynamic_cast<const AmdIndexIVFPQ*>(index)) {
        auto* res = new faiss::IndexIVFPQ(new faiss::IndexFlat(ipq->d, ipq->metric_type), ipq->d, ipq->nlist, ipq->M,
                                          ipq->nbits, ipq->metric_type);
        res->own_fields = true;
        ipq->copyTo(res);
        return res;
    } else if (auto isq = dynamic_cast<const AmdIndexIVFScalarQuantizer*>(index)) {
        auto* res = new faiss::IndexIVFScalarQuantizer(new faiss::IndexFlat(isq->d, isq->metric_type), isq->d, isq->nlist,
                                                       isq->sq.qtype, isq->metric_type, isq->by_residual);
        res->own_fields = true;
        isq->copyTo(res);
        return res;
    } else if (auto ipr = dynamic_cast<const faiss::IndexReplicas*>(index)) {
        FAISS_THROW_IF_NOT(ipr->count() > 0);
        return index_gpu_to_cpu(ipr->at(0)); // any replica holds everything
    } else if (auto ish = dynamic_cast<const faiss::IndexShards*>(index)) {
        // merge the shards back into one CPU index (GpuCloner.cpp:104-113 merge_index)
        FAISS_THROW_IF_NOT(ish->count() > 0);
        faiss::Index* res = index_gpu_to_cpu(ish->at(0));
        for (int i = 1; i < ish->count(); i++) {
            std::unique_ptr<faiss::Index> part(index_gpu_to_cpu(ish->at(i)));
            if (auto rf = dynamic_cast<faiss::IndexFlat*>(res)) {
                auto pf = dynamic_cast<faiss::IndexFlat*>(part.get());
                FAISS_THROW_IF_NOT(pf);
                rf->add(pf->ntotal, pf->get_xb());
            } else {
                auto ri = dynamic_cast<faiss::IndexIVF*>(res);
                auto pi = dynamic_cast<faiss::IndexIVF*>(part.get());
                FAISS_THROW_IF_NOT(ri && pi);
                ri->merge_from(*pi, ish->successive_ids ? ri->ntotal : 0);
            }
        }
        return res;
    }
    FAISS_THROW_MSG("index_gpu_to_cpu: not a backend index");
}

/// index_cpu_to_gpu_multiple (GpuCloner.cpp:325-522): replicas of the whole index, or shards
inline faiss::Index* index_cpu_to_gpu_multiple(const std::vector<AmdGpuResources*>& res, const faiss::Index* index,
                                               const AmdClonerOptions* options = nullptr) {
    AmdClonerOptions opt;
    if (options) opt = *options;
    const idx_t n = (idx_t)res.size();
    FAISS_THROW_IF_NOT(n >= 1);
    if (n == 1) return index_cpu_to_gpu(res[0], index);
    auto index_ivf = dynamic_cast<const faiss::IndexIVF*>(index);
    auto index_ivfpq = dynamic_cast<const faiss::IndexIVFPQ*>(index);
    auto index_ivfflat = dynamic_cast<const faiss::IndexIVFFlat*>(index);
    auto index_flat = dynamic_cast<const faiss::IndexFlat*>(index);
    auto index_ivfsq = dynamic_cast<const faiss::IndexIVFScalarQuantizer*>(index);
    FAISS_THROW_IF_NOT_MSG(index_ivfpq || index_ivfflat || index_flat || index_ivfsq,
                           "multi-device cloning is implemented for IndexFlat, IndexIVFFlat, IndexIVFPQ and "
                           "IndexIVFScalarQuantizer");
    if (!opt.shard) {
        auto* rep = new faiss::IndexReplicas();
        for (auto* r : res) rep->addIndex(index_cpu_to_gpu(r, index));
        rep->own_indices = true;
        return rep;
    }
    std::vector<faiss::Index*> shards(n);
    for (idx_t i = 0; i < n; i++) {
        // a short-lived CPU index holding shard i's subset, translated to the device right away
        // (const_casts as in the reference: the quantizer is only read)
        if (index_ivf) {
            std::unique_ptr<faiss::IndexIVF> idx2;
            if (index_ivfpq) {
                auto* p = new faiss::IndexIVFPQ(const_cast<faiss::Index*>(index_ivf->quantizer), index->d, index_ivf->nlist,
                                                index_ivfpq->pq.M, index_ivfpq->pq.nbits);
                p->metric_type = index->metric_type;
                p->pq = index_ivfpq->pq;
                p->use_precomputed_table = 0;
                idx2.reset(p);
            } else if (index_ivfsq) {
                auto* p = new faiss::IndexIVFScalarQuantizer(const_cast<faiss::Index*>(index_ivf->quantizer), index->d,
                                                             index_ivf->nlist, index_ivfsq->sq.qtype, index->metric_type,
                                                             index_ivfsq->by_residual);
                p->sq = index_ivfsq->sq;
                idx2.reset(p);
            } else {
                idx2.reset(new faiss::IndexIVFFlat(const_cast<faiss::Index*>(index_ivf->quantizer), index->d,
                                                   index_ivf->nlist, index->metric_type));
            }
            idx2->nprobe = index_ivf->nprobe;
            idx2->is_trained = index->is_trained;
            if (opt.shard_type == 2) {
                index_ivf->copy_subset_to(*idx2, InvertedLists::SUBSET_TYPE_ID_RANGE, i * index->ntotal / n,
                                          (i + 1) * index->ntotal / n);
            } else if (opt.shard_type == 1) {
                index_ivf->copy_subset_to(*idx2, InvertedLists::SUBSET_TYPE_ID_MOD, n, i);
            } else if (opt.shard_type == 4) {
                index_ivf->copy_subset_to(*idx2, InvertedLists::SUBSET_TYPE_INVLIST, i * index_ivf->nlist / n,
                                          (i + 1) * index_ivf->nlist / n);
            } else {
                FAISS_THROW_FMT("shard_type %d not implemented", opt.shard_type);
            }
            shards[i] = index_cpu_to_gpu(res[i], idx2.get());
        } else {
            auto* f = new AmdIndexFlat(res[i], index->d, index->metric_type);
            const idx_t i0 = index->ntotal * i / n, i1 = index->ntotal * (i + 1) / n;
            if (i1 > i0) f->add(i1 - i0, index_flat->get_xb() + (size_t)i0 * index->d);
            shards[i] = f;
        }
    }
    faiss::IndexShards* out;
    if (opt.common_ivf_quantizer && index_ivf) {
        // one coarse quantization in front of all shards, search_preassigned behind it (IndexShardsIVF.cpp:162-249)
        std::unique_ptr<faiss::IndexFlat> cq(new faiss::IndexFlat(index->d, index->metric_type));
        std::vector<float> c(index_ivf->nlist * (size_t)index->d);
        index_ivf->quantizer->reconstruct_n(0, index_ivf->nlist, c.data());
        cq->add(index_ivf->nlist, c.data());
        auto* sivf = new faiss::IndexShardsIVF(new AmdIndexFlat(res[0], cq.get()), index_ivf->nlist, true, false);
        sivf->own_fields = true;
        out = sivf;
    } else {
        out = new faiss::IndexShards(index->d, true, index_flat != nullptr);
    }
    out->own_indices = true;
    for (idx_t i = 0; i < n; i++) out->add_shard(shards[i]);
    FAISS_THROW_IF_NOT(out->ntotal == index->ntotal);
    return out;
}

} // namespace amd
} // namespace faiss
