// faiss_amd/csrc/index.h -- host-side C++ mirror of the reference interface for the hot path.
//
// Class and method names follow the reference so that its tests read the same here:
//   faiss::Index                      faiss/Index.h:101-431
//   faiss::gpu::StandardGpuResources  faiss/gpu/StandardGpuResources.h
//   faiss::gpu::GpuIndexFlat          faiss/gpu/GpuIndexFlat.h:41-153
//   faiss::gpu::GpuIndexIVF           faiss/gpu/GpuIndexIVF.h:37-153
//   faiss::gpu::GpuIndexIVFFlat       faiss/gpu/GpuIndexIVFFlat.h:33-126
//   faiss::gpu::GpuIndexIVFPQ         faiss/gpu/GpuIndexIVFPQ.h:53-176
//   faiss::IndexShards                faiss/IndexShards.h:19-108
//   faiss::Clustering                 faiss/Clustering.h:88-196
// The implementation is new (HIP runtime + the kernels in kernels.h); no reference code.
#pragma once
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <random>
#include <vector>
#include "common.h"

namespace faiss_amd {

// ------------------------------------------------------------------ device memory helpers
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int dev = 0; // device the allocation was made on (accounting)
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf();
    // grow (never shrinks); contents preserved up to keep_bytes
    void ensure(size_t bytes, size_t keep_bytes = 0, hipStream_t stream = nullptr);
    void release();
    template <typename T>
    T* as() const {
        return (T*)p;
    }
};

bool is_device_pointer(const void* p);
// device memory held by the library on `device` (every allocation is a DevBuf): live allocations, their bytes, the peak
// (StandardGpuResources::getMemoryInfo, faiss/gpu/StandardGpuResources.cpp:676); log = one stderr line per alloc / free
// (setLogMemoryAllocations, :327)
void device_memory_info(int device, size_t* allocations, size_t* bytes, size_t* peak_bytes);
void set_log_memory_allocations(int device, bool on);

// ------------------------------------------------------------------ resources
// One per device (and per host thread driving it), like StandardGpuResources: owns the
// stream all work of its indexes is ordered on, and per-kernel HIP-event timing.
class GpuResources {
   public:
    explicit GpuResources(int device = 0);
    ~GpuResources();
    int device;
    hipStream_t stream = nullptr; // the stream all work is ordered on: own_stream_ or the caller's (set_default_stream)
    hipStream_t own_stream_ = nullptr;
    // StandardGpuResources::setDefaultStream: order all further work on the caller's stream (null: back to our own).
    // Pending work on the old stream is drained first.
    void set_default_stream(hipStream_t s);
    int num_cus = 256;
    size_t temp_budget_bytes = (size_t)4 << 30; // cap for per-search scratch (reservoirs, IVF keys)

    void set_device() const;
    void sync() const;

    // ---- paged search of host-resident query batches (reference: GpuIndex::searchFromCpuPaged_, faiss/gpu/GpuIndex.cu:
    // 554-774): pinned double buffers + a copy stream, so that the H2D of page p+1 and the D2H of page p-1 run under the
    // kernels of page p.  Batches below paged_min_bytes take the direct path (one pageable copy each way).
    size_t paged_min_bytes = (size_t)64 << 20; // reference: 256 MiB (GpuIndex.cu:29); ours pipelines earlier
    idx_t paged_page_queries = 0;              // 0 = automatic; tests set small pages
    struct Pager {
        hipStream_t copy_stream = nullptr;
        void* pin_q[2] = {nullptr, nullptr};
        void* pin_d[2] = {nullptr, nullptr};
        void* pin_i[2] = {nullptr, nullptr};
        size_t pin_q_cap = 0, pin_r_cap = 0;
        hipEvent_t q_ready[2], r_ready[2], r_copied[2];
        bool events = false;
        std::mutex mu; // one paged search at a time per resources object (indexes sharing it may be searched from several threads)
    };
    mutable Pager pager;
    mutable long paged_searches = 0; // statistics: calls that took the paged path

    // ---- per-kernel timing with HIP events on `stream`
    bool profiling = false;
    struct Span {
        hipEvent_t a, b;
        std::string name;
    };
    mutable std::vector<Span> spans;
    mutable std::map<std::string, std::pair<double, long>> totals; // name -> (ms, launches)
    void begin_span(const char* name) const;
    void end_span() const;
    void collect() const; // sync + fold spans into totals
    void reset_profile() const;
};

struct SpanGuard {
    const GpuResources* r;
    SpanGuard(const GpuResources* r_, const char* name) : r(r_) {
        if (r->profiling) r->begin_span(name);
    }
    ~SpanGuard() {
        if (r->profiling) r->end_span();
    }
};

// ------------------------------------------------------------------ interruption
// faiss::InterruptCallback (faiss/impl/AuxIndexStructures.h:138-165): a process-wide hook polled between the tiles of
// long-running calls (search tiles, add pages, k-means iterations -- where faiss/gpu/impl/Distance.cu:245,266 polls);
// when it returns non-zero the call throws "computation interrupted".  null = none.
typedef int (*InterruptFn)(void*);
void set_interrupt_callback(InterruptFn fn, void* user);
void check_interrupt(); // throws FaissAmdException when the callback says so

// ------------------------------------------------------------------ IDSelector / SearchParameters
// faiss::IDSelector and its concrete selectors (faiss/impl/IDSelector.h:21-215): a predicate on the LABELS a search may
// return (row numbers for GpuIndexFlat, the stored user ids for the IVF indexes).  A search with a selector returns the
// k best among the selected vectors -- bit for bit what a search of an index holding only those vectors returns.
// The reference's GPU indexes take the parameter and ignore it outside the cuVS back-end (faiss/gpu/GpuIndexFlat.cu:232,
// GpuIndexIVF.cu:402); the CPU indexes honour it (IndexFlat.cpp:36-58, IndexIVF.cpp scan_codes), and so do the
// indexes here: the selector is compiled to a small device program (kernels.h SelProgram), evaluated once per stored
// row and search call into a bit mask that the scan kernels consult.
struct SelProgram;
struct IDSelector {
    virtual bool is_member(idx_t id) const = 0;
    virtual ~IDSelector() {}
    // append this selector's postfix instructions; device copies of id arrays / bitmaps are made (once per device) on
    // `stream` and owned by the selector
    virtual void compile(SelProgram& prog, int device, hipStream_t stream) const = 0;
};
struct IDSelectorAll : IDSelector {
    bool is_member(idx_t) const override { return true; }
    void compile(SelProgram& prog, int device, hipStream_t stream) const override;
};
struct IDSelectorRange : IDSelector {
    idx_t imin, imax; // imin <= id < imax
    bool assume_sorted;
    IDSelectorRange(idx_t imin_, idx_t imax_, bool assume_sorted_ = false)
            : imin(imin_), imax(imax_), assume_sorted(assume_sorted_) {}
    bool is_member(idx_t id) const override { return id >= imin && id < imax; }
    void compile(SelProgram& prog, int device, hipStream_t stream) const override;
};
// IDSelectorArray and IDSelectorBatch of the reference (a list of ids; linear search / hash set + Bloom filter there):
// one class here, a sorted array of the distinct ids, binary search on host and device.  The ids are copied.
struct IDSelectorBatch : IDSelector {
    std::vector<idx_t> ids; // sorted, distinct
    IDSelectorBatch(size_t n, const idx_t* indices);
    bool is_member(idx_t id) const override;
    void compile(SelProgram& prog, int device, hipStream_t stream) const override;

   private:
    // one device copy per device, made on first use under the lock: the shards of a threaded IndexShards search with
    // the same selector object concurrently, possibly on different devices
    mutable std::mutex mu_;
    mutable std::map<int, DevBuf> dev_;
};
typedef IDSelectorBatch IDSelectorArray;
// id selected iff id / 8 < n and bit id % 8 of bitmap[id / 8] is set (IDSelector.cpp:123-129); the bitmap is copied
struct IDSelectorBitmap : IDSelector {
    std::vector<uint8_t> bitmap;
    IDSelectorBitmap(size_t n, const uint8_t* bits) : bitmap(bits, bits + n) {}
    bool is_member(idx_t id) const override {
        const uint64_t u = (uint64_t)id;
        return (u >> 3) < bitmap.size() && ((bitmap[u >> 3] >> (u & 7)) & 1);
    }
    void compile(SelProgram& prog, int device, hipStream_t stream) const override;

   private:
    mutable std::mutex mu_;
    mutable std::map<int, DevBuf> dev_; // per device, see IDSelectorBatch
};
// the operands are not owned (as in the reference)
struct IDSelectorNot : IDSelector {
    const IDSelector* sel;
    explicit IDSelectorNot(const IDSelector* s) : sel(s) {}
    bool is_member(idx_t id) const override { return !sel->is_member(id); }
    void compile(SelProgram& prog, int device, hipStream_t stream) const override;
};
struct IDSelectorBinary : IDSelector {
    const IDSelector *lhs, *rhs;
    int op; // SelOp: SEL_AND / SEL_OR / SEL_XOR
    IDSelectorBinary(int op_, const IDSelector* l, const IDSelector* r) : lhs(l), rhs(r), op(op_) {}
    bool is_member(idx_t id) const override;
    void compile(SelProgram& prog, int device, hipStream_t stream) const override;
};
// faiss::SearchParameters (faiss/Index.h:86-93) / faiss::SearchParametersIVF (faiss/IndexIVF.h:70-80; nprobe is honoured
// by the reference GPU index through getCurrentNProbe_, faiss/gpu/GpuIndexIVF.cu:358-381).  nprobe <= 0: the index's own.
struct SearchParameters {
    const IDSelector* sel = nullptr; // not owned
    virtual ~SearchParameters() {}
};
struct SearchParametersIVF : SearchParameters {
    int nprobe = 0;
};

// ------------------------------------------------------------------ faiss::Index mirror
struct Index {
    int d = 0;
    idx_t ntotal = 0;
    bool verbose = false;
    bool is_trained = true;
    int metric_type = METRIC_L2;
    float metric_arg = 0.f; // faiss::Index::metric_arg (faiss/Index.h:114): the p of METRIC_Lp

    explicit Index(int d_ = 0, int metric = METRIC_L2) : d(d_), metric_type(metric) {}
    virtual ~Index() {}

    virtual void train(idx_t n, const float* x) {}
    virtual void add(idx_t n, const float* x) = 0;
    virtual void add_with_ids(idx_t n, const float* x, const idx_t* xids);
    virtual void search(idx_t n, const float* x, idx_t k, float* distances, idx_t* labels,
                        const SearchParameters* params = nullptr) const = 0;
    virtual void assign(idx_t n, const float* x, idx_t* labels, idx_t k = 1) const;
    virtual void reset() = 0;
    virtual void reconstruct(idx_t key, float* recons) const;
    virtual void reconstruct_n(idx_t i0, idx_t ni, float* recons) const;
    virtual void compute_residual(const float* x, float* residual, idx_t key) const;
    // residuals[i] = xs[i] - reconstruct(keys[i])  (faiss/Index.h:375-383)
    virtual void compute_residual_n(idx_t n, const float* xs, float* residuals, const idx_t* keys) const;
    // recons[i] = reconstruct(keys[i])  (faiss/Index.h:297-307)
    virtual void reconstruct_batch(idx_t n, const idx_t* keys, float* recons) const;
    virtual int device() const { return -1; }
};

// ------------------------------------------------------------------ GpuIndexFlat
class GpuIndexFlat : public Index {
   public:
    // use_float16: GpuIndexFlatConfig::useFloat16 (faiss/gpu/GpuIndexFlat.h:24-40, impl/FlatIndex.cu:39-135): vectors
    // are stored as fp16 only (half the resident bytes), queries are converted to fp16 as well, and distances are the
    // fp32 distances between those fp16 values
    GpuIndexFlat(std::shared_ptr<GpuResources> res, int dims, int metric, bool use_float16 = false);
    ~GpuIndexFlat() override;
    bool getUseFloat16() const { return use_float16_; }

    void add(idx_t n, const float* x) override;
    // params->sel: only rows the selector admits are returned (labels = row numbers)
    void search(idx_t n, const float* x, idx_t k, float* distances, idx_t* labels,
                const SearchParameters* params = nullptr) const override;
    void reset() override;
    void reconstruct(idx_t key, float* recons) const override;
    void reconstruct_n(idx_t i0, idx_t ni, float* recons) const override;
    // on the device, any pointer host or device; a key of -1 gives a NaN row
    // (faiss/gpu/GpuIndexFlat.cu:294-361, impl/VectorResidual.cu:26-60)
    void compute_residual(const float* x, float* residual, idx_t key) const override;
    void compute_residual_n(idx_t n, const float* xs, float* residuals, const idx_t* keys) const override;
    void reconstruct_batch(idx_t n, const idx_t* keys, float* recons) const override;
    int device() const override { return res_->device; }
    size_t getNumVecs() const { return (size_t)ntotal; }

    // device-resident search used internally (IVF coarse quantizer): xq_pad is [n][dpad] on the
    // device; results stay on the device.
    // defer_bad (nullable, device, [n]): the one-launch path (small databases) does not read its overflow count back -- a query it
    // cannot serve gets defer_bad[q] = 1 and labels of -1, and the CALLER redoes it (GpuIndexIVF: together with its own redo set, one
    // host round trip per search less); every other path serves all queries at once and writes zeros
    void search_device(int n, const float* xq_pad, int k, float* dD, idx_t* dI, uint32_t* defer_bad = nullptr) const;
    // search() without taking the lock / choosing the host path (x, distances, labels each host or device)
    void search_body_(idx_t n, const float* x, idx_t k, float* distances, idx_t* labels) const;
    // test hook: full distance matrix [n][ntotal] through the MFMA kernel (host or device out)
    void pairwise_distances(idx_t n, const float* x, float* out) const;
    // when true, search() uses the scalar cross-check kernel instead of the MFMA kernel
    bool use_simple_kernel = false;
    // fp16 MFMA candidate filter + exact fp32 re-rank (flat_filter.hip); results are bit-identical
    // to the fp32 MFMA scan, which remains the path for small databases, large k, data outside the
    // fp16 range and any query whose error band overflows its reservoir
    bool use_filter_kernel = true;
    idx_t filter_min_rows = 16384;
    // databases of <= 4096 rows, k <= 64 (an IVF coarse quantizer): the whole filter path in one launch (flat_small_fused_kernel,
    // round 6); off = the general launches (A/B knob, results never change)
    bool use_small_fused = true;
    // statistics of the last search() tile (tests / bench): queries re-run through the exact scan
    mutable int last_filter_overflow = 0;
    mutable bool last_used_filter = false;
    // test hook: approximate scores [n][ntotal] of the filter kernel and its per-query error bound
    void filter_scores(idx_t n, const float* x, float* scores, float* err_bound) const;

    int dpad() const { return dpad_; }
    const float* device_vectors() const; // fp32 rows [ntotal][dpad] (not available with fp16 storage)
    // the rows as fp32 [ntotal][dpad] written to dst (device): a copy (fp32 storage) or the widened fp16 values
    void rows_to_f32(float* dst) const;
    size_t resident_bytes() const;       // device bytes held for the database (rows, shadow copy, norms)
    std::shared_ptr<GpuResources> resources() const { return res_; }

   private:
    std::shared_ptr<GpuResources> res_;
    int dpad_;
    bool use_float16_ = false;
    DevBuf xb_;  // [cap][dpad] fp32 rows (absent with fp16 storage)
    DevBuf xbn_; // [cap]
    // fp32 view of the rows for the paths that need one (exact scan, test hooks): xb_ itself, or -- fp16 storage -- a
    // temporary widened copy in `tmp`
    const float* rows_f32_(DevBuf& tmp) const;
    // fp16 shadow copy for the filter kernel: rows padded to dh_ (multiple of 128) halfs, |y|^2/2
    int dh_;
    DevBuf xbh_, xbhn_;
    // operand-major copy of xbh_ for the one-launch kernel of small databases (flat_small_fused_kernel); valid for xbo_rows_ rows
    mutable DevBuf xbo_;
    mutable idx_t xbo_rows_ = -1;
    DevBuf scal_;            // device scalars: [0] max |x| bits, [1] max |y|^2 bits, [2] overflow counter
    // pinned host word the overflow count of a filter search is copied to (a pageable destination would make the 4-byte
    // read-back a staged, blocking copy: tens of microseconds on every search)
    mutable unsigned* h_novf_ = nullptr;
    float yn_max_ = 0.f;     // max squared norm over the database
    bool db_f16_ok_ = true;  // every database value inside the fp16 range (no NaN/inf)
    mutable DevBuf qh_, flags_, thr_, maxes_, ovf_list_, ovf_q_, ovf_d_, ovf_i_;
    void search_tile_exact_(int n, const float* xq_pad, int k, float* dD, idx_t* dI) const;
    // the "extra" metrics (L1, Linf, Lp, Canberra, BrayCurtis, JensenShannon, Jaccard): every distance as a key + select
    void search_tile_general_(int n, const float* xq_pad, int k, float* dD, idx_t* dI) const;
    bool filter_applicable_(int k) const;
    void plan_filter_(int n, int k, int& geom, int& nsplit, int& tstride, int& cap, int& gcap) const;
    // last plan (searches come in runs of one batch size): key = (n, k, ntotal, knob string)
    mutable struct {
        int n = -1, k = -1, geom = 0, nsplit = 0, tstride = 0, cap = 0, gcap = 0;
        idx_t ntotal = -1;
        std::string knobs;
    } plan_cache_;
    mutable std::mutex mu_;
    // IDSelector of the search in flight (under mu_): row mask, and the start values / norms with the excluded rows
    // set to -inf / +inf -- the filter and scan kernels then skip those rows without knowing about selectors
    mutable bool sel_active_ = false;
    mutable DevBuf sel_mask_, sel_xbhn_, sel_xbn_;
    void prepare_selector_(const IDSelector& sel) const;
    // persistent scratch
    mutable DevBuf q_raw_, q_pad_, q_norm_, res_keys_, res_cnt_, out_d_, out_i_, all_keys_, one_cnt_;
    void search_tile_(int n, const float* xq_pad, int k, float* dD, idx_t* dI, uint32_t* defer_bad = nullptr) const;
};

// faiss::ClusteringParameters (faiss/Clustering.h:27-60), the fields of the k-means loop proper
struct ClusteringParameters {
    int niter = 25;
    int nredo = 1;                 // runs from different random starts; the best objective wins
    int seed = 1234;
    int max_points_per_centroid = 256;
    int min_points_per_centroid = 39;
    bool verbose = false;
    bool spherical = false;        // L2-normalise the centroids after every update (inner-product clustering)
    bool int_centroids = false;    // round the centroid coordinates to integers after every update
    bool update_index = false;     // re-train the assignment index after every update (flat engines: nothing to train)
    bool frozen_centroids = false; // centroids given as input (Clustering::centroids before train) are never updated
};

// ------------------------------------------------------------------ GpuIndexIVF
class GpuIndexIVF : public Index {
   public:
    // coarse_quantizer: null = the index creates and owns a GpuIndexFlat (fp16 storage when coarse_f16:
    // GpuIndexIVFConfig::flatConfig.useFloat16, faiss/gpu/GpuIndexIVF.h:23-35); otherwise the CALLER's flat index on the same
    // device, not owned (own_fields = false, faiss/gpu/GpuIndexIVF.cu:41-70,107-108) -- it may hold its nlist centroids already
    // (the index is then trained as far as the coarse level goes) and may be shared between indexes.
    // indices_options: faiss/gpu/GpuIndicesOptions.h (INDICES_IVF: labels are list << 32 | offset).
    GpuIndexIVF(std::shared_ptr<GpuResources> res, int dims, int metric, int nlist, GpuIndexFlat* coarse_quantizer = nullptr,
                bool coarse_f16 = false, int indices_options = 3);
    ~GpuIndexIVF() override;

    int nlist;
    int nprobe = 1;
    GpuIndexFlat* quantizer; // owned iff own_fields
    bool own_fields = true;
    int indices_options = 3; // INDICES_CPU 0 (ids are kept on the device all the same), INDICES_IVF 1, INDICES_32_BIT 2, INDICES_64_BIT 3

    void train(idx_t n, const float* x) override;
    void add(idx_t n, const float* x) override;
    void add_with_ids(idx_t n, const float* x, const idx_t* xids) override;
    // GpuIndexIVF::add_core (faiss/gpu/GpuIndexIVF.cu:321-356; what contrib/ivf_tools.py add_preassigned calls): add
    // with the inverted-list assignment supplied by the caller instead of the coarse quantizer's.  precomputed_idx: [n],
    // host or device; an entry outside [0, nlist) leaves its vector out (it still counts in ntotal, like a NaN vector).
    // xids may be null (ids ntotal, ntotal + 1, ...).  Residuals (IVFPQ, IVFSQ) are taken against the GIVEN list's centroid.
    void add_core(idx_t n, const float* x, const idx_t* xids, const idx_t* precomputed_idx);
    // params: SearchParameters (sel) or SearchParametersIVF (sel, nprobe); the selector applies to the stored ids
    void search(idx_t n, const float* x, idx_t k, float* distances, idx_t* labels,
                const SearchParameters* params = nullptr) const override;
    // search with the coarse quantization supplied by the caller: assign / centroid_dis are [n][nprobe]
    // (host or device), -1 = no list (faiss/gpu/GpuIndexIVF.cu:408-488; IndexIVF::search_preassigned is what
    // IndexShardsIVF and the hybrid CPU-quantizer benchmarks call).  centroid_dis must be the quantizer's
    // distances for those lists (IVFPQ L2 adds them as the first term, as the reference does).
    void search_preassigned(idx_t n, const float* x, idx_t k, const idx_t* assign, const float* centroid_dis,
                            float* distances, idx_t* labels, const SearchParameters* params = nullptr) const;
    void reset() override;
    int device() const override { return res_->device; }

    // mirrors GpuIndexIVF::getListLength / getListIndices (faiss/gpu/GpuIndexIVF.h:97-110)
    size_t getListLength(idx_t list) const { return list_len_[list]; }
    std::vector<idx_t> getListIndices(idx_t list) const;
    // list payload in the reference's CPU layout (d floats or M bytes per entry)
    std::vector<uint8_t> getListVectorData(idx_t list) const;
    // copyFrom-style bulk load (train state + inverted lists), see include/faiss_amd_c.h
    virtual void set_centroids(const float* centroids);
    void set_lists(const uint32_t* list_sizes, const uint8_t* codes, const idx_t* ids);
    // GpuIndexIVFFlat / IVFPQ / IVFScalarQuantizer::reserveMemory (faiss/gpu/GpuIndexIVFFlat.h:64): room for numVecs
    // vectors up front, so that the adds that follow do not re-allocate the list arena
    void reserveMemory(size_t numVecs);
    // ...::reclaimMemory (GpuIndexIVFFlat.h:76): give back what the lists do not need (slack, holes of relocated lists,
    // add-path scratch); returns the bytes released.  The lists keep their order and contents.
    size_t reclaimMemory();
    // GpuIndexIVF::updateQuantizer (faiss/gpu/GpuIndexIVF.h:79): the coarse centroids were changed from outside
    // (quantizer->reset() / add()): re-derive what depends on them (training state, the per-vector IVFPQ term)
    void updateQuantizer();
    idx_t getNumLists() const { return nlist; }
    // vectors actually stored (ntotal counts the vectors add() was given, NaN rows included, like the reference:
    // faiss/gpu/GpuIndexIVF.cu:293-298)
    idx_t stored_vectors() const { return nstored_; }
    // bytes per list entry as the reference counts them (IndexIVF::code_size)
    size_t ref_code_size() const { return ref_row_bytes_(); }
    // arena statistics (rows): used (lists + slack + holes), holes left behind by relocated lists, allocated
    void arena_stats(int64_t* used, int64_t* holes, int64_t* allocated) const;
    // Clustering parameters used by train() (reference default niter=10 for the GPU IVF
    // quantizer, faiss/gpu/GpuIndexIVF.cu:80)
    // filter sweeps of IVFFlat / the scalar quantizer (rows of <= 128 coordinates): two-wave workgroups that walk sibling items -- the
    // query groups of one (list, row chunk) -- in lock-step (ivf_lm_filter.hip PAIR, round 6); off = a wavefront per item (A/B knob)
    int lmf_pair = 1; // 0 off, 1 sweep 1 only (measured: the only place it pays), 2 both sweeps
    int cp_niter = 10;
    int cp_seed = 1234;
    // the rest of GpuIndexIVF::cp (faiss/gpu/GpuIndexIVF.h, faiss/Clustering.h:27-60); niter / seed above win
    ClusteringParameters cp;

   protected:
    std::shared_ptr<GpuResources> res_;
    int dpad_;
    // the coarse centroids as fp32 rows [nlist][dpad_] on the device: the quantizer's own rows, or -- fp16 quantizer -- a
    // widened copy (residuals are taken against the fp16-rounded centroids, as the reference's reconstruct gives them);
    // refreshed after train / set_centroids / updateQuantizer
    const float* centroids_dev_() const;
    mutable DevBuf cent_f32_;
    mutable bool cent_dirty_ = true;
    size_t code_bytes_ = 0; // bytes per arena row (dpad*4 for IVFFlat, M for IVFPQ)
    int granule_ = 8;       // list capacities and starts are multiples of this many rows
    bool use_t2_ = false;   // IVFPQ L2: per-row term in arena_t2_
    bool use_rn_ = false;   // L2: squared norm of every stored (decoded) row in arena_rn_ (list-major scan)
    std::vector<uint32_t> list_len_, list_cap_;
    std::vector<int64_t> list_start_;
    int64_t arena_rows_ = 0;     // rows handed out so far (next free row)
    int64_t arena_cap_rows_ = 0; // rows allocated
    int64_t hole_rows_ = 0;      // rows of abandoned ranges (relocated lists)
    idx_t nstored_ = 0;
    DevBuf d_list_len_, d_list_start_, arena_, arena_ids_, arena_t2_, arena_rn_;
    mutable std::mutex mu_;
    mutable DevBuf q_raw_, q_pad_, c_dis_, c_ids_, c_bad_, prefix_, totals_, q_off_, keys_, out_d_, out_i_, one_cnt_;
    mutable int nprobe_eff_ = 1; // min(nprobe, nlist) of the search in flight
    // IDSelector of the search in flight (under mu_): one bit per arena row, null = none
    mutable DevBuf sel_mask_;
    mutable const uint32_t* cur_sel_mask_ = nullptr;
    // add-path scratch
    DevBuf a_xpad_, a_lab_, a_dis_, a_dest_, a_ids_, a_hist_, a_newlen_, a_jobs_;

    virtual void train_residual_(idx_t n, const float* x_dev_pad) {}
    // everything add()/search() needs besides the coarse centroids is in place (IVFPQ: the codebook)
    virtual bool extra_trained_() const { return true; }
    void update_is_trained_();
    void adopt_quantizer_();
    // called (under mu_) when rows [lists] were bulk-loaded or the quantizers changed: derived per-row data
    virtual void lists_changed_() {}
    // encode/scatter n staged vectors (device, padded) with given labels into arena rows dest
    virtual void append_(int n, const float* x_pad, const int64_t* d_labels, const int64_t* d_dest) = 0;
    virtual void scan_(int nq, const float* xq_pad, int k, const int64_t* h_qoff) const = 0;
    // fused search (ivf_fused.hip): kind/M and the type-specific pointers of IvfFusedParams
    virtual void fill_fused_(struct IvfFusedParams& p) const = 0;
    virtual int fused_kind_() const = 0;
    virtual int fused_M_() const { return 0; }
    // bytes per entry of the reference's inverted-list payload (invlists->code_size): what copy_lists takes and
    // getListVectorData returns
    virtual size_t ref_row_bytes_() const { return code_bytes_; }
    virtual int sq_chunk_bytes_() const { return 0; } // scalar quantizer: bytes per 16-component chunk
    mutable DevBuf part_keys_, part_cnt_, probe_len_, probe_start_;
    // ---- list-major search of large batches (ivf_listmajor.hip, kernels.h IvfLmParams)
    virtual bool lm_capable_() const { return false; }
    virtual bool lm_pq_lds_capable_() const { return false; } // IVFPQ: the codebook-in-LDS kernel serves this shape
    virtual void fill_lm_(struct IvfLmParams& p) const {}
    mutable DevBuf lm_prefix_, lm_p0_, lm_cnt_, lm_bucket_, lm_bstart_, lm_pairs_, lm_items_, lm_bounds_, lm_thr_, lm_keys_,
            lm_ovf_, lm_qn_;
    mutable uint32_t* h_lm_ = nullptr; // pinned: overflow count + item-table check of the search in flight
    mutable bool cur_lm_ = false;      // the search call in flight takes the list-major path (decided once per call)
    mutable int last_scan_mode_ = 0;
    mutable long lm_overflows_ = 0; // statistics: queries redone because their candidate segment overflowed
    void search_listmajor_(int ni, const float* xq_pad, const idx_t* c_ids, const float* c_dis, int np, int k, float* dD,
                           idx_t* dI, int level, std::vector<uint32_t>* redo = nullptr) const;
    // ---- list-major search behind the f16 filter (ivf_lm_filter.hip; IVFFlat, IVFPQ): results bit-identical to the
    // query-major scan.  lmf_capable_: index type / shape; lmf_prepare_: (re)build what the sweeps need beside the lists
    // (IVFFlat: the fp16 shadow of the arena rows + max |y|^2; IVFPQ: the fp16 codebook + norm bounds) when a list or a
    // quantizer changed since the last call, and fill the filter fields of `p`; false = the stored values leave the
    // fp16 range (the search takes the query-major scan).
    virtual bool lmf_capable_() const { return false; }
    virtual bool lmf_prepare_(struct IvfLmParams& p) const { return false; }
    // flavour of the sweeps that serve this index (ivf_lmf_queries_per_item): 0 IVFFlat, 1 IVFPQ's codebook kernel, 2 the
    // pair-operand IVFFlat kernel (scalar quantizer, IVFPQ through its decoded residuals)
    virtual int lmf_sweep_kind_() const { return fused_kind_(); }
    // bytes freed by dropping the sweeps' own copies of the lists (rebuilt at the next list-major search)
    virtual size_t lmf_release_() { return 0; }
    virtual size_t lmf_shadow_bytes_() const { return 0; }
    // add() keeps a LIVE copy up to date instead of invalidating it (round 5; the reference appends into the layout it scans,
    // faiss/gpu/impl/IVFBase.cu:595-905): after a page has been appended, the 32-row blocks of the copy that hold new rows --
    // and every block of a list grow_lists_ relocated -- are rebuilt from the lists (launch_ivf_lmf_shadow /
    // launch_ivf_lmf_code_shadow with a first row per list).  d_first_row: device [nlist], 0xffffffff = list unchanged.
    virtual void lmf_patch_(const uint32_t* d_first_row) {}
    std::vector<uint8_t> moved_;        // lists relocated by grow_lists_ during the add page in flight
    std::vector<uint32_t> h_first_row_; // host image of the patch's first rows (alive until the page's synchronisation)
    DevBuf a_first_row_;
    mutable bool cur_lmf_ = false;        // the list-major search in flight runs the filter sweeps
    mutable bool cur_preassigned_ = false; // ... with the caller's coarse assignment (search_preassigned)
    mutable const uint32_t* cur_coarse_bad_ = nullptr; // ... with the coarse quantizer's deferred overflow flags (c_bad_) of the tile
    mutable int last_scan_arith_ = 0;     // oracle restatement of the last search: 0 query-major arithmetic, 1 f32 list-major
    mutable bool shadow_dirty_ = true;    // a list changed since the fp16 shadow was built
    mutable bool lmf_quant_dirty_ = true; // a quantizer changed since the fp16 codebook / norm bounds were built
    mutable DevBuf lm_prefixg_, lm_gmin_, lm_thrf_, lm_candpr_, lm_q16_, lm_qflags_, lm_xnb_, lm_pqgrid_, lm_scalar_, lm_pair16_, lm_pairxh_, lm_errf_, lm_an_, lm_rowbase_;
    // queries whose candidate segment overflowed (or that leave the fp16 range) are appended to `redo`
    void search_listmajor_filter_chunk_(int ni, int q0, const float* xq_pad, const idx_t* c_ids, const float* c_dis, int np, int k,
                                        float* dD, idx_t* dI, int64_t stride, int rt_g, int64_t gstride, int min_stride,
                                        std::vector<uint32_t>& redo) const;
    void search_listmajor_chunk_(int ni, const float* xq_pad, const idx_t* c_ids, const float* c_dis, int np, int k,
                                 float* dD, idx_t* dI, int level, int64_t stride, int min_p1, int RT, int64_t c1max) const;
    void upload_list_tables_();
    void ensure_arena_(int64_t rows);
    // make room for new_len[l] entries in every list (relocating the lists that outgrow their slack); est[l]
    // (nullable) = expected final length, used as the new capacity of a list that has to move
    void grow_lists_(const std::vector<uint32_t>& new_len, const std::vector<double>* est);
    void compact_(bool tight = false); // tight: no per-list slack beyond the granule (reclaimMemory)
    void add_core_(idx_t n, const float* x, const idx_t* xids, const idx_t* assign = nullptr);
    void search_core_(idx_t n, const float* x, idx_t k, float* distances, idx_t* labels, const idx_t* assign,
                      const float* centroid_dis, int nprobe_now, const IDSelector* sel) const;
    void search_core_body_(idx_t n, const float* x, idx_t k, float* distances, idx_t* labels, const idx_t* assign,
                           const float* centroid_dis, int nprobe_now) const;

   public:
    // when false, search() takes the unfused path (every distance as a key in HBM + select);
    // kept for cross-checking the fused kernel and for shapes that do not fit its LDS budget
    bool use_fused_scan = true;
    // which scan serves search(): 0 = automatic (list-major for batches of >= 2048 queries that probe every list
    // >= 8 times on average, without IDSelector, when the index type / dimension support it), 1 = query-major always
    // (ivf_fused.hip), 2 = list-major always (throws when unsupported): IVFFlat / IVFPQ behind the f16 filter of
    // ivf_lm_filter.hip -- results bit-identical to the query-major scan --, the scalar quantizer on the f32 matrix pipe;
    // 3 = list-major on the f32 matrix pipe for every type (round 3's scan, its own arithmetic: DESIGN.md 3.9).
    int scan_mode = 0;
    // The filter sweeps of the automatic mode keep their own copy of the lists (IVFFlat: an fp16 shadow, + 2 d bytes per
    // row on top of 4 d + 12; IVFPQ: the codes again in operand order, + M bytes per row): built at the first list-major
    // search, kept up to date by add().  false = never build it in mode 0 (query-major / f32 list-major serve the
    // calls, same results); a build that fails for lack of memory falls back the same way by itself.
    bool use_filter_shadow = true;
    // device bytes held for the database: the lists (rows / codes, ids, per-row terms) and the sweeps' copies
    void resident_bytes(size_t* lists, size_t* shadow) const;
    // what the last search() call used: 1 = query-major, 2 = list-major
    int last_scan_mode() const { return last_scan_mode_; }
    // ... and under which `arith` the oracle restates it: 0 = the query-major arithmetic (also what the list-major scan
    // behind the f16 filter returns, scan_mode 2 on IVFFlat / IVFPQ), 1 = the f32 list-major arithmetic (scan_mode 3, and
    // the scalar quantizer's list-major scan)
    int last_scan_arith() const { return last_scan_arith_; }
    long list_major_overflows() const { return lm_overflows_; }
    // the rule of scan_mode 0
    bool list_major_rule(idx_t n, int nprobe_now, idx_t k, bool has_selector) const;
    // tuning experiments of the filter path (faiss_amd_GpuIndexIVF_set_lmf_tuning; 0 = the built-in rule): rows of a list
    // per work item (a multiple of 256), 32-row blocks per granule (a power of two <= 8), candidate room per query
    int lmf_rows_per_item = 0, lmf_gran_blocks = 0, lmf_cand_cap = 0, lmf_min_stride = 0;
    int lmf_sample_shift = 0; // sweep 1 on the first rows_per_item >> shift rows of every item: 0 = the rule, -1 = all rows
    // test hook (faiss_amd_GpuIndexIVF_test_filter_dump): the ESTIMATES of the f16 filter sweeps for every row the n
    // queries probe, as keys (ordkey(estimate) << 32 | scan position) at keys_out[q * stride + scan position], and the
    // error band E_q the bound kernel derives for each query (band_out [n], written for queries whose probed lists hold
    // >= k granules; others keep the caller's value).  x: host, n <= 65536; stride >= rows any query probes.
    void test_filter_dump(idx_t n, const float* x, int nprobe_now, idx_t k, int64_t stride, unsigned long long* keys_out,
                          float* band_out) const;
};

class GpuIndexIVFFlat : public GpuIndexIVF {
   public:
    GpuIndexIVFFlat(std::shared_ptr<GpuResources> res, int dims, int nlist, int metric, GpuIndexFlat* coarse_quantizer = nullptr,
                    bool coarse_f16 = false, int indices_options = 3);
    // stored vector of id `key` (ids as given to add_with_ids / generated by add); faiss/gpu/GpuIndexIVFFlat.cu:370-390
    // offers reconstruct_n for contiguous ids, this is the same by-id lookup
    void reconstruct_n(idx_t i0, idx_t ni, float* recons) const override;
    void reconstruct(idx_t key, float* recons) const override { reconstruct_n(key, 1, recons); }

   protected:
    void fill_fused_(struct IvfFusedParams& p) const override;
    int fused_kind_() const override { return 0; }
    size_t ref_row_bytes_() const override { return (size_t)d * 4; }
    void append_(int n, const float* x_pad, const int64_t* d_labels, const int64_t* d_dest) override;
    void scan_(int nq, const float* xq_pad, int k, const int64_t* h_qoff) const override;
    void lists_changed_() override;
    bool lm_capable_() const override;
    void fill_lm_(struct IvfLmParams& p) const override;
    bool lmf_capable_() const override;
    bool lmf_prepare_(struct IvfLmParams& p) const override;
    mutable DevBuf arena_h_;          // fp16 shadow of the arena rows [arena_cap_rows_ + 128][dh_]
    size_t lmf_shadow_bytes_() const override { return arena_h_.cap; }
    void lmf_patch_(const uint32_t* d_first_row) override;
    void lmf_shadow_room_() const; // arena_h_ holds the blocks of arena_cap_rows_ rows (contents kept)
    size_t lmf_release_() override {
        const size_t b = arena_h_.cap;
        arena_h_.release();
        shadow_dirty_ = true;
        return b;
    }
    mutable float shadow_yn_max_ = 0.f;
    mutable bool shadow_in_range_ = true;
};

class GpuIndexIVFPQ : public GpuIndexIVF {
   public:
    GpuIndexIVFPQ(std::shared_ptr<GpuResources> res, int dims, int nlist, int M, int nbits, int metric,
                  GpuIndexFlat* coarse_quantizer = nullptr, bool coarse_f16 = false, int indices_options = 3);
    int M, nbits, dsub;
    int pq_niter = 25; // faiss::ClusteringParameters default used by ProductQuantizer::train
    // the M sub-quantizers are trained as one k-means (ivf_kernels.hip pq_train_*); false, or FAISS_AMD_PQ_TRAIN_LOOP=1 in
    // the environment: one Clustering per sub-space, the round-2 loop (same codebook bit for bit)
    bool pq_train_batched = true;
    void set_pq_centroids(const float* pq); // [M][256][dsub]
    std::vector<float> get_pq_centroids() const;
    // GpuIndexIVFPQ.h:98-113.  The term decomposition behind "precomputed codes" is always on here for L2 (one table per
    // query + a per-vector term) and has no meaning for inner product (the reference forces it off there,
    // GpuIndexIVFPQ.cu:228-241): the setter records the request, the getter reports what is IN FORCE -- true for L2,
    // false for inner product, whatever was asked for.  Likewise the lookup tables are always fp32 (on a power-of-two
    // grid): useFloat16LookupTables is accepted, getFloat16LookupTables() says false.
    void setPrecomputedCodes(bool enable) { precomputed_codes_ = enable; }
    bool getPrecomputedCodes() const { return metric_type == METRIC_L2; }
    bool getPrecomputedCodesRequested() const { return precomputed_codes_; }
    bool getFloat16LookupTables() const { return false; }
    int getNumSubQuantizers() const { return M; }
    int getBitsPerCode() const { return nbits; }
    int getCentroidsPerSubQuantizer() const { return 1 << nbits; }

   protected:
    void fill_fused_(struct IvfFusedParams& p) const override;
    int fused_kind_() const override { return 1; }
    int fused_M_() const override { return M; }
    bool precomputed_codes_ = false;
    DevBuf pq_;   // [M][256][dsub]
    DevBuf pq_t_; // [256][M][dsub]: the order the scan kernels build their lookup table in
    DevBuf zero_row_; // dpad zeros: the "centroid" with which the per-row term kernels yield |r^|^2 (arena_rn_)
    bool lm_capable_() const override;
    bool lmf_capable_() const override;
    bool lmf_prepare_(struct IvfLmParams& p) const override;
    mutable DevBuf pq16_;             // fp16 codebook [M][256][dsub]
    mutable DevBuf arena_cs_;         // operand-major copy of the codes for the filter sweeps (kernels.h IvfLmParams::arena_cs)
    size_t lmf_shadow_bytes_() const override { return arena_cs_.cap + pq16_.cap; }
    void lmf_patch_(const uint32_t* d_first_row) override;
    void lmf_shadow_room_() const; // arena_cs_ holds the blocks of arena_cap_rows_ rows (contents kept)
    void lmf_write_copy_(const uint32_t* d_first_row) const;
    bool lmf_two_copies_() const;
    bool lmf_codebook_capable_() const;
    bool lmf_decoded_() const;
    int lmf_sweep_kind_() const override { return lmf_decoded_() ? 2 : 1; }
    mutable DevBuf sq_one_, sq_nil_; // decoded mode: scale 1 / offset 0 for the pair-operand preparation

   public:
    // PQ64 over d = 128: the sweeps' codebook twice in LDS with different code -> bank maps, copy chosen per (row, sub-quantizer)
    // when the copy of the codes is written (ivf_lm_filter.hip lmf_code_choice_kernel).  false: round 4's one-copy sweeps.
    // Changing it drops the copy of the codes (rebuilt by the next list-major search).
    // PQ64 over d = 128: codebook gathers of the filter sweeps as one VALU instruction + one ds_read_b32 (ivf_lm_filter.hip FG,
    // round 6); off = the gathers as hipcc compiles them (A/B knob, results never change)
    bool lmf_fast_gather = true;
    bool lmf_two_copies = false; // measured in round 5 and not adopted: fewer LDS conflicts, more VALU, slower
    void set_lmf_two_copies(bool on) {
        std::lock_guard<std::mutex> g(mu_);
        if (on != lmf_two_copies) {
            lmf_two_copies = on;
            (void)lmf_release_();
        }
    }

   protected:
    size_t lmf_release_() override {
        const size_t b = arena_cs_.cap;
        arena_cs_.release();
        shadow_dirty_ = true;
        return b;
    }
    mutable float pq_yn_max_ = 0.f, cn_max_ = 0.f;
    mutable bool pq16_in_range_ = true;
    bool lm_pq_lds_capable_() const override { return ivf_lm_pq_lds_supported_(); }
    bool ivf_lm_pq_lds_supported_() const;
    void train_pq_batched_(idx_t nt, const float* res, std::vector<float>& pq);
    void fill_lm_(struct IvfLmParams& p) const override;
    bool extra_trained_() const override { return pq_.p != nullptr; }
    void lists_changed_() override;
    void train_residual_(idx_t n, const float* x_dev_pad) override;
    void append_(int n, const float* x_pad, const int64_t* d_labels, const int64_t* d_dest) override;
    void scan_(int nq, const float* xq_pad, int k, const int64_t* h_qoff) const override;
};

// faiss::gpu::GpuIndexIVFScalarQuantizer (faiss/gpu/GpuIndexIVFScalarQuantizer.h:27-131) over
// faiss::ScalarQuantizer (faiss/impl/ScalarQuantizer.h:25-120): every vector (or its residual to the list centroid)
// is stored as one small code per dimension.  qtype takes the reference's enum values for the types its GPU index
// supports (GpuScalarQuantizer.cuh:20-33): QT_8bit 0, QT_4bit 1, QT_8bit_uniform 2, QT_4bit_uniform 3, QT_fp16 4,
// QT_8bit_direct 5, QT_6bit 6.
class GpuIndexIVFScalarQuantizer : public GpuIndexIVF {
   public:
    GpuIndexIVFScalarQuantizer(std::shared_ptr<GpuResources> res, int dims, int nlist, int qtype, int metric,
                               bool encode_residual = true, GpuIndexFlat* coarse_quantizer = nullptr, bool coarse_f16 = false,
                               int indices_options = 3);
    int qtype;
    bool by_residual;
    size_t code_size; // bytes per vector as the reference counts them (sq.code_size)
    // ScalarQuantizer::rangestat / rangestat_arg (ScalarQuantizer.h:60-70); only RS_minmax (0) is trained here
    int rangestat = 0;
    float rangestat_arg = 0.f;
    // ScalarQuantizer::trained (ScalarQuantizer.h:72-73): {vmin, vdiff} (uniform types) or vmin[d] then vdiff[d]
    std::vector<float> trained;
    void set_trained(const float* t, size_t n);

   protected:
    void fill_fused_(struct IvfFusedParams& p) const override;
    int fused_kind_() const override { return 2; }
    size_t ref_row_bytes_() const override { return code_size; }
    int sq_chunk_bytes_() const override;
    bool extra_trained_() const override { return !needs_training_() || !trained.empty(); }
    void train_residual_(idx_t n, const float* x_dev_pad) override;
    void append_(int n, const float* x_pad, const int64_t* d_labels, const int64_t* d_dest) override;
    void scan_(int nq, const float* xq_pad, int k, const int64_t* h_qoff) const override;
    // list-major scan (ivf_listmajor.hip, kind 2): every code type, d <= 128
    void lists_changed_() override;
    bool lm_capable_() const override;
    void fill_lm_(struct IvfLmParams& p) const override;
    // behind the f16 filter (round 5): the IVFFlat sweeps over an fp16 copy of the centred codes, d <= 512, every code type
    bool lmf_capable_() const override;
    bool lmf_prepare_(struct IvfLmParams& p) const override;
    void lmf_patch_(const uint32_t* d_first_row) override;
    size_t lmf_shadow_bytes_() const override { return arena_h_.cap; }
    size_t lmf_release_() override {
        const size_t b = arena_h_.cap;
        arena_h_.release();
        shadow_dirty_ = true;
        return b;
    }

   private:
    mutable DevBuf arena_h_;                // fp16 copy of the centred codes, operand-major 32-row blocks (IvfLmParams::arena_h)
    mutable float sq_rn_max_ = 0.f, sq_cn_max_ = 0.f; // max |s o code'|^2 (L2) / bound of max |code'|^2 over the stored rows
    mutable bool shadow_in_range_ = true;
    float sq_bn_ = 0.f;                     // |b'|^2
    void lmf_shadow_room_() const;
    void lmf_write_copy_(const uint32_t* d_first_row, bool merge) const;
    int dsq_;      // d rounded up to 16
    int ct_;       // SqCodeType of the scan kernel
    float levels_; // 255 / 15 / 63 (code range of the type), 0 for the types without a trained range
    bool needs_training_() const { return levels_ > 0.f; }
    DevBuf vmin_, vdiff_; // [d] (uniform types: replicated) -- the encoder's view of `trained`
    DevBuf sq_s_, sq_b_;  // [dsq_] scale / offset per dimension of the decoder: x^ = fmaf(code, s, b)
    DevBuf sq_bm_;        // [dsq_] list-major scan: offset of the centred codes, fmaf(mid, s, b)
    DevBuf sq_zero_;      // [dsq_] zeros (list-major scan: the centroid of a search without residual encoding)
    void upload_tables_();
    void row_norms_all_(); // arena_rn_ = |s o code|^2 of every arena row (list-major scan, L2)
};

// ------------------------------------------------------------------ IndexShards
// Database-sharded meta index: add() splits rows evenly over the shards, search() runs every
// shard on its own host thread and k-way merges the partial results on the host
// (faiss/IndexShards.cpp:135-265, faiss/utils/Heap.cpp:166-240).
class IndexShards : public Index {
   public:
    IndexShards(int d, bool threaded, bool successive_ids);
    ~IndexShards() override;
    bool threaded, successive_ids;
    bool own_indices = false;
    void add_shard(Index* idx);
    int count() const { return (int)shards_.size(); }
    Index* at(int i) { return shards_[i]; }
    void train(idx_t n, const float* x) override;
    void add(idx_t n, const float* x) override;
    void add_with_ids(idx_t n, const float* x, const idx_t* xids) override;
    // params are handed to every shard (faiss/IndexShards.cpp:196-265); a selector sees the shard-local labels
    void search(idx_t n, const float* x, idx_t k, float* distances, idx_t* labels,
                const SearchParameters* params = nullptr) const override;
    void reset() override;

   private:
    std::vector<Index*> shards_;
    void sync_();
};

// ------------------------------------------------------------------ IndexReplicas
// Every replica holds the whole database; add() goes to all of them, search() splits the QUERIES evenly over
// the replicas, one host thread each (faiss/IndexReplicas.cpp:91-175).  What index_cpu_to_gpu_multiple builds
// by default (GpuMultipleClonerOptions::shard = false, faiss/gpu/GpuClonerOptions.h:57-59).
class IndexReplicas : public Index {
   public:
    IndexReplicas(int d, bool threaded);
    ~IndexReplicas() override;
    bool threaded;
    bool own_indices = false;
    void add_replica(Index* idx);
    int count() const { return (int)replicas_.size(); }
    Index* at(int i) { return replicas_[i]; }
    void train(idx_t n, const float* x) override;
    void add(idx_t n, const float* x) override;
    void add_with_ids(idx_t n, const float* x, const idx_t* xids) override;
    // search parameters are refused, like the reference (faiss/IndexReplicas.cpp:129-130)
    void search(idx_t n, const float* x, idx_t k, float* distances, idx_t* labels,
                const SearchParameters* params = nullptr) const override;
    void reconstruct(idx_t key, float* recons) const override;
    void reset() override;

   private:
    std::vector<Index*> replicas_;
    void sync_();
};

// faiss::gpu::GpuParameterSpace::set_index_parameter (faiss/gpu/GpuAutoTune.cpp:81-114): "nprobe" on IVF indexes,
// recursively through IndexReplicas / IndexShards; "use_precomputed_table" is accepted on IVFPQ (the per-vector term is
// always on here).  Throws on a parameter the index does not have.
void set_index_parameter(Index* index, const std::string& name, double val);

// merge nshard sorted partial results [s][nq][k] into [nq][k] under (distance, label) order;
// base[s] is added to shard s's labels (successive_ids translation), may be null.
void merge_knn_results(int metric, idx_t nq, idx_t k, int nshard, const float* all_d, const idx_t* all_i,
                       const idx_t* base, float* D, idx_t* I);

// Brute-force k-nearest-neighbour on raw row-major fp32 arrays, host or device (the float32 / row-major subset
// of faiss::gpu::bfKnn, faiss/gpu/GpuDistance.h:32-152 and GpuDistance.cu:bfKnn).  Same kernels, same tie rule and
// the same bits as GpuIndexFlat::search on an index holding `vectors`.
// ScalarQuantizer::train with RS_meanstd (1) / RS_quantiles (2) / RS_optim (3) on the host (index.cpp): rows [n][d] dense, k = 2^bits
void sq_train_rangestat_host(int rangestat, float rangestat_arg, int64_t n, int d, int k, bool uniform, const float* rows,
                             std::vector<float>& trained);
void bfKnn(std::shared_ptr<GpuResources> res, int metric, const float* vectors, idx_t num_vectors, const float* queries,
           idx_t num_queries, int dims, idx_t k, float* out_distances, idx_t* out_indices);

// The whole operator surface of faiss::gpu::bfKnn (GpuDistanceParams, faiss/gpu/GpuDistance.h:32-152): f32 / f16 / bf16
// inputs, row or column major, int64 or int32 indices, optional distances, k = -1 for all pairwise distances; and
// bfKnn_tiling (GpuDistance.cu:430-570): inputs beyond the given device-memory limits are processed in (query chunk) x
// (vector chunk) tiles whose partial results are merged.  All-fp16 vectors run as an fp16-storage index.
struct DistanceParams {
    int metric = METRIC_L2;
    float metricArg = 0.f;
    int k = 0, dims = 0;
    const void* vectors = nullptr;
    int vectorType = 1; // 1 = f32, 2 = f16, 3 = bf16
    bool vectorsRowMajor = true;
    idx_t numVectors = 0;
    const float* vectorNorms = nullptr; // accepted, unused (norms are recomputed in the kernels' own summation order)
    const void* queries = nullptr;
    int queryType = 1;
    bool queriesRowMajor = true;
    idx_t numQueries = 0;
    float* outDistances = nullptr;
    bool ignoreOutDistances = false;
    int outIndicesType = 1; // 1 = int64, 2 = int32
    void* outIndices = nullptr;
    int device = -1;
};
void bfKnn(std::shared_ptr<GpuResources> res, const DistanceParams& args);
void bfKnn_tiling(std::shared_ptr<GpuResources> res, const DistanceParams& args, size_t vectorsMemoryLimit,
                  size_t queriesMemoryLimit);

// device-side variant used by the one-process-per-GPU sharded search (faiss_amd/distributed.py):
// all pointers are device pointers on res's device; work is ordered on res's stream and the
// call returns after the stream has drained.
void merge_knn_results_device(GpuResources& res, int metric, int nq, int k, int nshard, const float* all_d,
                              const idx_t* all_i, const idx_t* base_host, float* D, idx_t* I);

// ------------------------------------------------------------------ Clustering (k-means)
struct Clustering : ClusteringParameters {
    int d, k;
    std::vector<float> centroids; // [k][d]
    std::vector<float> obj;       // objective (sum of distances) per iteration
    Clustering(int d_, int k_) : d(d_), k(k_) {}
    // `index` is the assignment engine (reset / add(k centroids) / search k=1),
    // exactly how the reference drives a GPU flat index (faiss/Clustering.cpp:255-357).
    // With a GpuIndexFlat as the engine the whole loop runs on the device (training set uploaded once, assignment,
    // counting sort by cluster and centroid update as kernels; only the objective and the cluster sizes come back
    // per iteration) and x may be a host or a device pointer; results are bit-identical to the host loop.
    // ldx: row stride of x in floats (0 = d; other strides for device data only)
    // `centroids` non-empty on entry = initial centroids (a multiple of d floats, at most k of them: the rest is drawn
    // from the training set), faiss/Clustering.cpp:330-345; with frozen_centroids they stay as given.
    void train(idx_t n, const float* x, Index& index, int64_t ldx = 0);
    bool last_train_on_device = false;
    // the refill of empty clusters (faiss/Clustering.cpp:180-232 split_clusters) on `centroids`, for callers that run the
    // iterations themselves (the product quantizer trains its M sub-spaces as one k-means): hassign = cluster sizes
    void split_empty_clusters(std::mt19937_64& rng, idx_t nx, std::vector<idx_t>& hassign) { split_clusters_(rng, nx, hassign, 0); }

   private:
    void train_once_(idx_t n, const float* x, Index& index, int64_t ldx, uint64_t run_seed, const std::vector<float>& init);
    void train_host_(idx_t n, const float* x, Index& index, uint64_t run_seed, const std::vector<float>& init);
    void train_device_(idx_t n, const float* x, int64_t ldx, class GpuIndexFlat& flat, uint64_t run_seed,
                       const std::vector<float>& init);
    // post_process_centroids (faiss/Clustering.cpp:236-253) on the host copy; first_free: centroids [0, first_free) frozen
    void post_process_(int first_free);
    void split_clusters_(std::mt19937_64& rng, idx_t nx, std::vector<idx_t>& hassign, int first_free = 0);
};

} // namespace faiss_amd
