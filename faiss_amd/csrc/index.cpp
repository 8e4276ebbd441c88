// faiss_amd/csrc/index.cpp -- host logic of the MI355X backend (see index.h for the reference
// interfaces each class mirrors).  Every distance / selection / scan is a HIP kernel from
// kernels.h; there is no CPU compute fallback anywhere in this file.
#include "index.h"
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <queue>
#include <random>
#include <thread>
#include "kernels.h"

namespace faiss_amd {

// ====================================================================== memory helpers
// Accounting of every device allocation of the library (all of them are DevBuf's), per device: what
// StandardGpuResources::getMemoryInfo reports and setLogMemoryAllocations prints (faiss/gpu/StandardGpuResources.cpp:327,676).
namespace {
constexpr int kMaxDevices = 64;
std::atomic<size_t> g_dev_bytes[kMaxDevices], g_dev_allocs[kMaxDevices], g_dev_peak[kMaxDevices];
std::atomic<int> g_log_allocs[kMaxDevices];
int current_device_() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        dev = 0;
    }
    return dev >= 0 && dev < kMaxDevices ? dev : 0;
}
void account_(int dev, size_t bytes, bool alloc, const void* ptr) {
    if (alloc) {
        const size_t now = g_dev_bytes[dev].fetch_add(bytes) + bytes;
        g_dev_allocs[dev].fetch_add(1);
        size_t peak = g_dev_peak[dev].load();
        while (now > peak && !g_dev_peak[dev].compare_exchange_weak(peak, now)) {}
    } else {
        g_dev_bytes[dev].fetch_sub(bytes);
        g_dev_allocs[dev].fetch_sub(1);
    }
    if (g_log_allocs[dev].load(std::memory_order_relaxed))
        fprintf(stderr, "faiss_amd: device %d %s %zu bytes at %p (%zu bytes in %zu allocations)\n", dev, alloc ? "alloc" : "free", bytes,
                ptr, g_dev_bytes[dev].load(), g_dev_allocs[dev].load());
}
} // namespace
void device_memory_info(int device, size_t* allocations, size_t* bytes, size_t* peak_bytes) {
    FA_THROW_IF_NOT(device >= 0 && device < kMaxDevices);
    if (allocations) *allocations = g_dev_allocs[device].load();
    if (bytes) *bytes = g_dev_bytes[device].load();
    if (peak_bytes) *peak_bytes = g_dev_peak[device].load();
}
void set_log_memory_allocations(int device, bool on) {
    FA_THROW_IF_NOT(device >= 0 && device < kMaxDevices);
    g_log_allocs[device].store(on ? 1 : 0);
}
DevBuf::~DevBuf() {
    release();
}
void DevBuf::release() {
    if (p) {
        (void)hipFree(p);
        account_(dev, cap, false, p);
        p = nullptr;
        cap = 0;
    }
}
void DevBuf::ensure(size_t bytes, size_t keep_bytes, hipStream_t stream) {
    if (bytes <= cap) return;
    size_t ncap = std::max(bytes, cap + cap / 2);
    ncap = round_up(ncap, 256);
    if (p && keep_bytes == 0) release(); // nothing to keep: the peak must not hold the old and the new buffer together
    void* np = nullptr;
    hipError_t me = hipMalloc(&np, ncap);
    if (me == hipErrorOutOfMemory && ncap > round_up(bytes, 256)) {
        (void)hipGetLastError();
        ncap = round_up(bytes, 256); // the geometric head-room is a convenience, not a requirement
        me = hipMalloc(&np, ncap);
    }
    if (me == hipErrorOutOfMemory) {
        (void)hipGetLastError(); // (not sticky: the next call must not see it)
        throw DeviceOutOfMemory("out of device memory allocating " + std::to_string(ncap) + " bytes");
    }
    HIP_CHECK(me);
    if (p && keep_bytes) {
        HIP_CHECK(hipMemcpyAsync(np, p, keep_bytes, hipMemcpyDeviceToDevice, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
    }
    if (p) release();
    p = np;
    cap = ncap;
    dev = current_device_();
    account_(dev, cap, true, p);
}

bool is_device_pointer(const void* p) {
    if (!p) return false;
    hipPointerAttribute_t a;
    hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) {
        (void)hipGetLastError(); // plain malloc'ed host memory: not an error for us
        return false;
    }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

// ====================================================================== resources
GpuResources::GpuResources(int device_) : device(device_) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) {
        FA_THROW_MSG("no HIP device available: this backend has no CPU fallback");
    }
    FA_THROW_IF_NOT_MSG(device >= 0 && device < n, "invalid device");
    HIP_CHECK(hipSetDevice(device));
    HIP_CHECK(hipStreamCreateWithFlags(&own_stream_, hipStreamNonBlocking));
    stream = own_stream_;
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, device));
    num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
}
GpuResources::~GpuResources() {
    (void)hipSetDevice(device);
    for (auto& s : spans) {
        (void)hipEventDestroy(s.a);
        (void)hipEventDestroy(s.b);
    }
    if (pager.events) {
        for (int s = 0; s < 2; s++) {
            (void)hipEventDestroy(pager.q_ready[s]);
            (void)hipEventDestroy(pager.r_ready[s]);
            (void)hipEventDestroy(pager.r_copied[s]);
        }
    }
    for (int s = 0; s < 2; s++) {
        if (pager.pin_q[s]) (void)hipHostFree(pager.pin_q[s]);
        if (pager.pin_d[s]) (void)hipHostFree(pager.pin_d[s]);
        if (pager.pin_i[s]) (void)hipHostFree(pager.pin_i[s]);
    }
    if (pager.copy_stream) (void)hipStreamDestroy(pager.copy_stream);
    if (own_stream_) (void)hipStreamDestroy(own_stream_);
}
void GpuResources::set_default_stream(hipStream_t s) {
    set_device();
    collect(); // (timing events were recorded on the old stream)
    sync();
    stream = s ? s : own_stream_;
}
void GpuResources::set_device() const {
    HIP_CHECK(hipSetDevice(device));
}
void GpuResources::sync() const {
    HIP_CHECK(hipStreamSynchronize(stream));
}
void GpuResources::begin_span(const char* name) const {
    Span s;
    s.name = name;
    HIP_CHECK(hipEventCreate(&s.a));
    HIP_CHECK(hipEventCreate(&s.b));
    HIP_CHECK(hipEventRecord(s.a, stream));
    spans.push_back(s);
}
void GpuResources::end_span() const {
    HIP_CHECK(hipEventRecord(spans.back().b, stream));
}
void GpuResources::collect() const {
    sync();
    for (auto& s : spans) {
        float ms = 0.f;
        HIP_CHECK(hipEventElapsedTime(&ms, s.a, s.b));
        auto& t = totals[s.name];
        t.first += ms;
        t.second += 1;
        (void)hipEventDestroy(s.a);
        (void)hipEventDestroy(s.b);
    }
    spans.clear();
}
void GpuResources::reset_profile() const {
    collect();
    totals.clear();
}

// ====================================================================== interruption
static std::mutex g_interrupt_mu;
static InterruptFn g_interrupt_fn = nullptr;
static void* g_interrupt_user = nullptr;
void set_interrupt_callback(InterruptFn fn, void* user) {
    std::lock_guard<std::mutex> g(g_interrupt_mu);
    g_interrupt_fn = fn;
    g_interrupt_user = user;
}
void check_interrupt() {
    InterruptFn fn;
    void* user;
    {
        std::lock_guard<std::mutex> g(g_interrupt_mu);
        fn = g_interrupt_fn;
        user = g_interrupt_user;
    }
    if (fn && fn(user)) FA_THROW_MSG("computation interrupted");
}

void set_index_parameter(Index* index, const std::string& name, double val) {
    FA_THROW_IF_NOT_MSG(index, "null index");
    if (auto* rep = dynamic_cast<IndexReplicas*>(index)) {
        for (int i = 0; i < rep->count(); i++) set_index_parameter(rep->at(i), name, val);
        return;
    }
    if (auto* sh = dynamic_cast<IndexShards*>(index)) {
        for (int i = 0; i < sh->count(); i++) set_index_parameter(sh->at(i), name, val);
        return;
    }
    if (name == "nprobe") {
        if (auto* ivf = dynamic_cast<GpuIndexIVF*>(index)) {
            FA_THROW_IF_NOT_MSG(val >= 1 && val <= kMaxSelectionK, "nprobe must be in [1, 2048]");
            ivf->nprobe = (int)val;
            return;
        }
    }
    if (name == "use_precomputed_table" && dynamic_cast<GpuIndexIVFPQ*>(index)) return;
    FA_THROW_MSG("ParameterSpace: parameter '" + name + "' does not apply to this index");
}

// ====================================================================== Index base
void Index::add_with_ids(idx_t, const float*, const idx_t*) {
    FA_THROW_MSG("add_with_ids not implemented for this type of index");
}
void Index::assign(idx_t n, const float* x, idx_t* labels, idx_t k) const {
    std::vector<float> dis((size_t)n * k);
    search(n, x, k, dis.data(), labels);
}
void Index::reconstruct(idx_t, float*) const {
    FA_THROW_MSG("reconstruct not implemented for this type of index");
}
void Index::reconstruct_n(idx_t i0, idx_t ni, float* recons) const {
    for (idx_t i = 0; i < ni; i++) reconstruct(i0 + i, recons + i * d);
}
void Index::compute_residual(const float* x, float* residual, idx_t key) const {
    reconstruct(key, residual);
    for (int i = 0; i < d; i++) residual[i] = x[i] - residual[i];
}

void Index::compute_residual_n(idx_t n, const float* xs, float* residuals, const idx_t* keys) const {
    for (idx_t i = 0; i < n; i++) compute_residual(xs + i * d, residuals + i * d, keys[i]);
}
void Index::reconstruct_batch(idx_t n, const idx_t* keys, float* recons) const {
    for (idx_t i = 0; i < n; i++) reconstruct(keys[i], recons + i * d);
}

// stage n x d floats (host or device, dense rows) into a padded device buffer
static void stage_padded(const GpuResources& res, const float* x, int64_t n, int d, int dpad, DevBuf& raw,
                         float* dst) {
    if (n == 0) return;
    const float* src = x;
    if (!is_device_pointer(x)) {
        raw.ensure((size_t)n * d * sizeof(float));
        HIP_CHECK(hipMemcpyAsync(raw.p, x, (size_t)n * d * sizeof(float), hipMemcpyHostToDevice,
                                 res.stream));
        src = raw.as<float>();
    }
    launch_pad_rows(src, d, n, d, dst, dpad, dpad, res.stream);
}

static void copy_out(const GpuResources& res, void* dst, const void* dsrc, size_t bytes) {
    if (bytes == 0) return;
    HIP_CHECK(hipMemcpyAsync(dst, dsrc, bytes,
                             is_device_pointer(dst) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                             res.stream));
}

// ====================================================================== paged host search
// Pages of a host-resident query batch through pinned double buffers: while the kernels of page p run on the
// resources' stream, the copy stream moves page p+1 to the device and the results of page p-1 back, and the host
// thread does the pageable <-> pinned memcpys of both.  `compute(ni, dq, dD, dI)` searches one page whose queries
// (dense [ni][d]), distances and labels are DEVICE buffers; it enqueues on R.stream (it may synchronise).
// Reference: GpuIndex::searchFromCpuPaged_ (faiss/gpu/GpuIndex.cu:554-774).
template <typename Compute>
static void paged_host_search(const GpuResources& R, idx_t n, const float* x, int d, idx_t k, float* distances,
                              idx_t* labels, idx_t page, Compute compute) {
    GpuResources::Pager& P = R.pager;
    std::lock_guard<std::mutex> pager_lock(P.mu);
    if (!P.copy_stream) HIP_CHECK(hipStreamCreateWithFlags(&P.copy_stream, hipStreamNonBlocking));
    if (!P.events) {
        for (int s = 0; s < 2; s++) {
            HIP_CHECK(hipEventCreateWithFlags(&P.q_ready[s], hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&P.r_ready[s], hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&P.r_copied[s], hipEventDisableTiming));
        }
        P.events = true;
    }
    const size_t qbytes = (size_t)page * d * 4, rbytes = (size_t)page * k * 8;
    if (qbytes > P.pin_q_cap) {
        for (int s = 0; s < 2; s++) {
            if (P.pin_q[s]) (void)hipHostFree(P.pin_q[s]);
            HIP_CHECK(hipHostMalloc(&P.pin_q[s], qbytes, hipHostMallocDefault));
        }
        P.pin_q_cap = qbytes;
    }
    if (rbytes > P.pin_r_cap) {
        for (int s = 0; s < 2; s++) {
            if (P.pin_d[s]) (void)hipHostFree(P.pin_d[s]);
            if (P.pin_i[s]) (void)hipHostFree(P.pin_i[s]);
            HIP_CHECK(hipHostMalloc(&P.pin_d[s], rbytes / 2, hipHostMallocDefault));
            HIP_CHECK(hipHostMalloc(&P.pin_i[s], rbytes, hipHostMallocDefault));
        }
        P.pin_r_cap = rbytes;
    }
    DevBuf dq[2], dD[2], dI[2];
    for (int s = 0; s < 2; s++) {
        dq[s].ensure(qbytes);
        dD[s].ensure((size_t)page * k * 4);
        dI[s].ensure((size_t)page * k * 8);
    }
    const idx_t npages = (n + page - 1) / page;
    auto rows = [&](idx_t p) { return std::min(page, n - p * page); };
    auto issue_h2d = [&](idx_t p) {
        const int s = (int)(p & 1);
        // slot s was last read by the H2D of page p-2: wait for that copy itself, not for the kernels that consumed it
        if (p >= 2) HIP_CHECK(hipEventSynchronize(P.q_ready[s]));
        memcpy(P.pin_q[s], x + (size_t)p * page * d, (size_t)rows(p) * d * 4);
        HIP_CHECK(hipMemcpyAsync(dq[s].p, P.pin_q[s], (size_t)rows(p) * d * 4, hipMemcpyHostToDevice, P.copy_stream));
        HIP_CHECK(hipEventRecord(P.q_ready[s], P.copy_stream));
    };
    auto drain = [&](idx_t p) {
        const int s = (int)(p & 1);
        HIP_CHECK(hipEventSynchronize(P.r_copied[s]));
        memcpy(distances + (size_t)p * page * k, P.pin_d[s], (size_t)rows(p) * k * 4);
        memcpy(labels + (size_t)p * page * k, P.pin_i[s], (size_t)rows(p) * k * 8);
    };
    issue_h2d(0);
    for (idx_t p = 0; p < npages; p++) {
        const int s = (int)(p & 1);
        const idx_t ni = rows(p);
        if (p + 1 < npages) issue_h2d(p + 1);
        HIP_CHECK(hipStreamWaitEvent(R.stream, P.q_ready[s], 0));
        if (p >= 2) HIP_CHECK(hipStreamWaitEvent(R.stream, P.r_copied[s], 0)); // result slot s free again
        compute(ni, dq[s].as<float>(), dD[s].as<float>(), dI[s].as<idx_t>());
        HIP_CHECK(hipEventRecord(P.r_ready[s], R.stream));
        HIP_CHECK(hipStreamWaitEvent(P.copy_stream, P.r_ready[s], 0));
        HIP_CHECK(hipMemcpyAsync(P.pin_d[s], dD[s].p, (size_t)ni * k * 4, hipMemcpyDeviceToHost, P.copy_stream));
        HIP_CHECK(hipMemcpyAsync(P.pin_i[s], dI[s].p, (size_t)ni * k * 8, hipMemcpyDeviceToHost, P.copy_stream));
        HIP_CHECK(hipEventRecord(P.r_copied[s], P.copy_stream));
        if (p >= 1) drain(p - 1); // its D2H had the kernels of page p to complete under
    }
    drain(npages - 1);
    R.sync();
    R.paged_searches++;
}
// page size of a paged search: at least four pages, whole thousands of queries, never more than `tile`
static idx_t paged_page_size(const GpuResources& R, idx_t n, idx_t tile) {
    if (R.paged_page_queries > 0) return std::min<idx_t>(R.paged_page_queries, tile);
    idx_t page = std::max<idx_t>(4096, (n / 4 + 1023) / 1024 * 1024);
    return std::min<idx_t>(std::min<idx_t>(page, 65536), tile);
}
static bool use_paged_path(const GpuResources& R, idx_t n, int d, const float* x, const float* distances,
                           const idx_t* labels) {
    return (size_t)n * d * 4 >= R.paged_min_bytes && !is_device_pointer(x) && !is_device_pointer(distances) &&
           !is_device_pointer(labels);
}

// ====================================================================== GpuIndexFlat
GpuIndexFlat::GpuIndexFlat(std::shared_ptr<GpuResources> res, int dims, int metric, bool use_float16)
        : Index(dims, metric), res_(std::move(res)), use_float16_(use_float16) {
    FA_THROW_IF_NOT_MSG(dims > 0, "dimension must be positive");
    // L2 / inner product on the matrix pipes; the extra metrics of the reference's flat index (faiss/gpu/impl/
    // GeneralDistance.cuh: L1, Linf, Lp, Canberra, BrayCurtis, JensenShannon, Jaccard) on a plain brute-force pass
    FA_THROW_IF_NOT_MSG(metric_supported(0, metric), "unsupported metric type");
    dpad_ = (int)round_up(dims, 8);
    dh_ = (int)round_up(dims, kFilterSlab);
    is_trained = true;
    res_->set_device();
    scal_.ensure(16);
    HIP_CHECK(hipMemset(scal_.p, 0, 16));
}
GpuIndexFlat::~GpuIndexFlat() {
    (void)hipSetDevice(res_->device);
    if (h_novf_) (void)hipHostFree(h_novf_);
}

const float* GpuIndexFlat::device_vectors() const {
    FA_THROW_IF_NOT_MSG(!use_float16_, "fp32 rows are not resident with fp16 storage");
    return xb_.as<float>();
}
size_t GpuIndexFlat::resident_bytes() const {
    const size_t n = (size_t)ntotal;
    return (use_float16_ ? 0 : n * dpad_ * 4) + n * 4 + (n + kFilterTileRows) * ((size_t)dh_ * 2 + 4);
}
void GpuIndexFlat::rows_to_f32(float* dst) const {
    if (ntotal == 0) return;
    res_->set_device();
    if (!use_float16_) {
        HIP_CHECK(hipMemcpyAsync(dst, xb_.p, (size_t)ntotal * dpad_ * 4, hipMemcpyDeviceToDevice, res_->stream));
    } else {
        launch_f16_rows_to_f32(xbh_.as<_Float16>(), dh_, ntotal, dpad_, dst, res_->stream);
    }
    res_->sync();
}
const float* GpuIndexFlat::rows_f32_(DevBuf& tmp) const {
    if (!use_float16_) return xb_.as<float>();
    tmp.ensure(std::max<size_t>((size_t)ntotal * dpad_ * 4, 256));
    launch_f16_rows_to_f32(xbh_.as<_Float16>(), dh_, ntotal, dpad_, tmp.as<float>(), res_->stream);
    return tmp.as<float>();
}

void GpuIndexFlat::reset() {
    std::lock_guard<std::mutex> g(mu_);
    xbo_rows_ = -1;
    ntotal = 0;
    yn_max_ = 0.f;
    db_f16_ok_ = true;
    res_->set_device();
    HIP_CHECK(hipMemsetAsync(scal_.p, 0, 16, res_->stream));
}

void GpuIndexFlat::add(idx_t n, const float* x) {
    if (n == 0) return;
    FA_THROW_IF_NOT_MSG(x, "null input");
    FA_THROW_IF_NOT_MSG(ntotal + n < ((idx_t)1 << 31), "at most 2^31-1 vectors per device index");
    std::lock_guard<std::mutex> g(mu_);
    res_->set_device();
    xbo_rows_ = -1; // (the operand-major copy of a small database is rebuilt by the next search that wants it)
    const size_t row = (size_t)dpad_ * sizeof(float);
    if (!use_float16_) xb_.ensure((size_t)(ntotal + n) * row, (size_t)ntotal * row, res_->stream);
    xbn_.ensure((size_t)(ntotal + n) * sizeof(float), (size_t)ntotal * sizeof(float), res_->stream);
    // one tile of padding rows behind the last one: the filter kernel reads whole 64-row tiles
    xbh_.ensure((size_t)(ntotal + n + kFilterTileRows) * dh_ * 2, (size_t)ntotal * dh_ * 2, res_->stream);
    xbhn_.ensure((size_t)(ntotal + n + kFilterTileRows) * sizeof(float), (size_t)ntotal * sizeof(float), res_->stream);
    // page the upload so the raw staging buffer stays bounded (reference: GpuIndex.cu:197-217)
    const idx_t page = std::max<idx_t>(1, ((idx_t)256 << 20) / ((idx_t)d * 4));
    DevBuf f16_stage; // fp16 storage: the padded fp32 page lives only until its fp16 copy and norms exist
    if (use_float16_) f16_stage.ensure((size_t)std::min(page, n) * row);
    for (idx_t i0 = 0; i0 < n; i0 += page) {
        idx_t ni = std::min(page, n - i0);
        float* dst = use_float16_ ? f16_stage.as<float>() : xb_.as<float>() + (size_t)(ntotal + i0) * dpad_;
        stage_padded(*res_, x + (size_t)i0 * d, ni, d, dpad_, q_raw_, dst);
        // (the index holds fp16 VALUES: norms, range statistics and the filter's bias are those of the rounded rows)
        if (use_float16_) launch_round_f16_inplace(dst, (int64_t)ni * dpad_, res_->stream);
        launch_l2_norms(dst, dpad_, ni, dpad_, xbn_.as<float>() + ntotal + i0, res_->stream);
        // fp16 shadow copy + |y|^2/2 for the filter kernel, range / norm statistics
        launch_convert_f16(dst, dpad_, ni, d, xbh_.as<char>() + (size_t)(ntotal + i0) * dh_ * 2, dh_,
                           scal_.as<unsigned>(), nullptr, res_->stream);
        // (the -inf padding after the last row is rewritten by every page; only the final one survives)
        launch_half_norms(xbn_.as<float>() + ntotal + i0, ni, kFilterTileRows, metric_type,
                          xbhn_.as<float>() + ntotal + i0, res_->stream);
        launch_max_f32(xbn_.as<float>() + ntotal + i0, ni, scal_.as<unsigned>() + 1, res_->stream);
        res_->sync(); // q_raw_ is reused by the next page
    }
    // the tile of padding rows behind the last row: zeros (their bias is -inf, their coordinates must not
    // turn that into a NaN next to real rows)
    HIP_CHECK(hipMemsetAsync(xbh_.as<char>() + (size_t)(ntotal + n) * dh_ * 2, 0, (size_t)kFilterTileRows * dh_ * 2,
                             res_->stream));
    {
        unsigned bits[2];
        HIP_CHECK(hipMemcpy(bits, scal_.p, 8, hipMemcpyDeviceToHost));
        float amax, ynm;
        memcpy(&amax, &bits[0], 4);
        memcpy(&ynm, &bits[1], 4);
        db_f16_ok_ = amax <= 65000.f && ynm <= FLT_MAX; // NaN/inf are reported as +inf bits
        yn_max_ = ynm;
    }
    ntotal += n;
}

void GpuIndexFlat::reconstruct_n(idx_t i0, idx_t ni, float* recons) const {
    FA_THROW_IF_NOT_MSG(i0 >= 0 && ni >= 0 && i0 + ni <= ntotal, "index out of range");
    if (ni == 0) return;
    std::lock_guard<std::mutex> g(mu_);
    res_->set_device();
    DevBuf tmp;
    const float* src = xb_.as<float>() + (size_t)i0 * dpad_;
    if (use_float16_) {
        tmp.ensure((size_t)ni * dpad_ * 4);
        launch_f16_rows_to_f32(xbh_.as<_Float16>() + (size_t)i0 * dh_, dh_, ni, dpad_, tmp.as<float>(), res_->stream);
        src = tmp.as<float>();
    }
    HIP_CHECK(hipMemcpy2DAsync(recons, (size_t)d * 4, src, (size_t)dpad_ * 4, (size_t)d * 4, (size_t)ni,
                               is_device_pointer(recons) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                               res_->stream));
    res_->sync();
}
void GpuIndexFlat::reconstruct(idx_t key, float* recons) const {
    reconstruct_n(key, 1, recons);
}

// x (null: none), keys and out may each live on the host or on the device
static void rows_by_key(const GpuResources& R, const float* x, const idx_t* keys, idx_t n, int d, const float* rows,
                        int64_t ld_rows, idx_t nrows, float* out, const _Float16* rows16 = nullptr, int64_t ld_rows16 = 0) {
    if (n == 0) return;
    DevBuf bx, bk, bo;
    const float* dx = x;
    if (x && !is_device_pointer(x)) {
        bx.ensure((size_t)n * d * 4);
        HIP_CHECK(hipMemcpyAsync(bx.p, x, (size_t)n * d * 4, hipMemcpyHostToDevice, R.stream));
        dx = bx.as<float>();
    }
    const idx_t* dk = keys;
    if (!is_device_pointer(keys)) {
        bk.ensure((size_t)n * 8);
        HIP_CHECK(hipMemcpyAsync(bk.p, keys, (size_t)n * 8, hipMemcpyHostToDevice, R.stream));
        dk = bk.as<idx_t>();
    }
    float* dout = out;
    if (!is_device_pointer(out)) {
        bo.ensure((size_t)n * d * 4);
        dout = bo.as<float>();
    }
    launch_rows_by_key(dx, d, dk, n, d, rows, ld_rows, nrows, dout, d, R.stream, rows16, ld_rows16);
    if (dout != out) HIP_CHECK(hipMemcpyAsync(out, dout, (size_t)n * d * 4, hipMemcpyDeviceToHost, R.stream));
    R.sync();
}
void GpuIndexFlat::compute_residual_n(idx_t n, const float* xs, float* residuals, const idx_t* keys) const {
    FA_THROW_IF_NOT_MSG(n >= 0, "negative count");
    if (n == 0) return;
    FA_THROW_IF_NOT_MSG(xs && residuals && keys, "null argument");
    std::lock_guard<std::mutex> g(mu_);
    res_->set_device();
    rows_by_key(*res_, xs, keys, n, d, use_float16_ ? nullptr : xb_.as<float>(), dpad_, ntotal, residuals,
                xbh_.as<_Float16>(), dh_);
}
void GpuIndexFlat::compute_residual(const float* x, float* residual, idx_t key) const {
    compute_residual_n(1, x, residual, &key);
}
void GpuIndexFlat::reconstruct_batch(idx_t n, const idx_t* keys, float* recons) const {
    FA_THROW_IF_NOT_MSG(n >= 0, "negative count");
    if (n == 0) return;
    FA_THROW_IF_NOT_MSG(keys && recons, "null argument");
    std::lock_guard<std::mutex> g(mu_);
    res_->set_device();
    rows_by_key(*res_, nullptr, keys, n, d, use_float16_ ? nullptr : xb_.as<float>(), dpad_, ntotal, recons,
                xbh_.as<_Float16>(), dh_);
}

void bfKnn(std::shared_ptr<GpuResources> res, int metric, const float* vectors, idx_t num_vectors, const float* queries,
           idx_t num_queries, int dims, idx_t k, float* out_distances, idx_t* out_indices) {
    FA_THROW_IF_NOT_MSG(res, "null resources");
    FA_THROW_IF_NOT_MSG(dims >= 1 && num_vectors >= 0 && num_queries >= 0, "bad sizes");
    FA_THROW_IF_NOT_MSG(metric == METRIC_L2 || metric == METRIC_INNER_PRODUCT, "bfKnn: metric must be L2 or inner product");
    if (num_queries == 0) return;
    FA_THROW_IF_NOT_MSG((vectors || num_vectors == 0) && queries && out_distances && out_indices, "null argument");
    // the kernels read 16-byte aligned, zero-padded rows with their norms next to them: a private padded copy
    // is the index add() path; the arrays the caller passed are never written
    GpuIndexFlat tmp(res, dims, metric);
    if (num_vectors) tmp.add(num_vectors, vectors);
    tmp.search(num_queries, queries, k, out_distances, out_indices);
}

static size_t dtype_size(int type) {
    FA_THROW_IF_NOT_MSG(type >= 1 && type <= 3, "unknown DistanceDataType");
    return type == 1 ? 4 : 2;
}
// fp32 row-major device copy [n][d] of a matrix of any supported type / layout (host or device source); when the
// source already is that (device, f32, row major) it is returned as it is
static const float* as_f32_rows(const GpuResources& R, const void* src, int type, bool row_major, idx_t n, int d,
                                DevBuf& raw, DevBuf& out) {
    const size_t bytes = (size_t)n * d * dtype_size(type);
    const void* dsrc = src;
    if (!is_device_pointer(src)) {
        raw.ensure(std::max<size_t>(bytes, 16));
        HIP_CHECK(hipMemcpyAsync(raw.p, src, bytes, hipMemcpyHostToDevice, R.stream));
        dsrc = raw.p;
    }
    if (type == 1 && row_major) return (const float*)dsrc;
    out.ensure(std::max<size_t>((size_t)n * d * 4, 16));
    launch_convert_matrix(dsrc, type, row_major, n, d, out.as<float>(), R.stream);
    return out.as<float>();
}

void bfKnn(std::shared_ptr<GpuResources> res, const DistanceParams& a) {
    FA_THROW_IF_NOT_MSG(res, "null resources");
    FA_THROW_IF_NOT_MSG(a.device == -1 || a.device == res->device, "args.device differs from the device of the resources");
    FA_THROW_IF_NOT_MSG(metric_supported(0, a.metric), "bfKnn: unsupported metric");
    FA_THROW_IF_NOT_MSG(a.k != -1 || !is_general_metric(a.metric), "bfKnn: k = -1 (all pairwise distances) is L2 / inner product only");
    FA_THROW_IF_NOT_MSG(a.dims >= 1 && a.numVectors >= 0 && a.numQueries >= 0, "bad sizes");
    FA_THROW_IF_NOT_MSG(a.k == -1 || (a.k >= 1 && a.k <= kMaxSelectionK), "k must be in [1, 2048], or -1 for all pairwise distances");
    if (a.numQueries == 0) return;
    FA_THROW_IF_NOT_MSG((a.vectors || a.numVectors == 0) && a.queries, "bfKnn: vectors / queries must be provided (passed null)");
    res->set_device();
    const GpuResources& R = *res;
    DevBuf vraw, vf32, qraw, qf32;
    const float* v = a.numVectors ? as_f32_rows(R, a.vectors, a.vectorType, a.vectorsRowMajor, a.numVectors, a.dims, vraw, vf32) : nullptr;
    const float* q = as_f32_rows(R, a.queries, a.queryType, a.queriesRowMajor, a.numQueries, a.dims, qraw, qf32);
    // fp16 vectors AND queries: an fp16-storage index holds exactly those values at half the bytes
    GpuIndexFlat tmp(res, a.dims, a.metric, a.vectorType == 2 && a.queryType == 2);
    tmp.metric_arg = a.metricArg;
    if (a.numVectors) tmp.add(a.numVectors, v);
    if (a.k == -1) {
        FA_THROW_IF_NOT_MSG(a.outDistances, "bfKnn: outDistances must be provided for k = -1");
        tmp.pairwise_distances(a.numQueries, q, a.outDistances);
        return;
    }
    FA_THROW_IF_NOT_MSG(a.outIndices, "bfKnn: outIndices must be provided (passed null)");
    FA_THROW_IF_NOT_MSG(a.ignoreOutDistances || a.outDistances, "bfKnn: outDistances must be provided (passed null)");
    FA_THROW_IF_NOT_MSG(a.outIndicesType == 1 || a.outIndicesType == 2, "unknown IndicesDataType");
    DevBuf dd, di;
    float* D = a.outDistances;
    if (a.ignoreOutDistances || !D) {
        dd.ensure((size_t)a.numQueries * a.k * 4);
        D = dd.as<float>();
    }
    if (a.outIndicesType == 1) {
        tmp.search(a.numQueries, q, a.k, D, (idx_t*)a.outIndices);
        return;
    }
    di.ensure((size_t)a.numQueries * a.k * 8);
    tmp.search(a.numQueries, q, a.k, D, di.as<idx_t>());
    DevBuf d32;
    int32_t* out32 = (int32_t*)a.outIndices;
    const bool out_dev = is_device_pointer(a.outIndices);
    if (!out_dev) {
        d32.ensure((size_t)a.numQueries * a.k * 4);
        out32 = d32.as<int32_t>();
    }
    launch_i64_to_i32(di.as<idx_t>(), a.numQueries * a.k, out32, R.stream);
    if (!out_dev)
        HIP_CHECK(hipMemcpyAsync(a.outIndices, out32, (size_t)a.numQueries * a.k * 4, hipMemcpyDeviceToHost, R.stream));
    R.sync();
}

void bfKnn_tiling(std::shared_ptr<GpuResources> res, const DistanceParams& a, size_t vectorsMemoryLimit,
                  size_t queriesMemoryLimit) {
    if (vectorsMemoryLimit == 0 && queriesMemoryLimit == 0) {
        bfKnn(res, a);
        return;
    }
    // reference: faiss/gpu/GpuDistance.cu:430-570 (same argument checks)
    FA_THROW_IF_NOT_MSG(a.k > 0, "bfKnn_tiling: tiling is only supported for k > 0");
    FA_THROW_IF_NOT_MSG(a.numQueries > 0 && a.queries && a.vectors, "bfKnn_tiling: vectors and queries must be provided");
    FA_THROW_IF_NOT_MSG(a.outIndices, "bfKnn: outIndices must be provided (passed null)");
    const size_t qsz = dtype_size(a.queryType), vsz = dtype_size(a.vectorType);
    const size_t lsz = a.outIndicesType == 1 ? 8 : 4;
    idx_t qshard = a.numQueries, vshard = a.numVectors;
    if (queriesMemoryLimit > 0) {
        FA_THROW_IF_NOT_MSG(!is_device_pointer(a.queries), "bfKnn_tiling: queries should be in CPU memory when queriesMemoryLimit > 0");
        FA_THROW_IF_NOT_MSG(a.queriesRowMajor, "bfKnn_tiling: tiling queries is only supported in row major mode");
        qshard = (idx_t)(queriesMemoryLimit / ((size_t)a.k * (qsz + lsz) + (size_t)a.dims * qsz));
        FA_THROW_IF_NOT_MSG(qshard > 0, "bfKnn_tiling: queriesMemoryLimit is too low");
    }
    if (vectorsMemoryLimit > 0) {
        FA_THROW_IF_NOT_MSG(!is_device_pointer(a.vectors), "bfKnn_tiling: vectors should be in CPU memory when vectorsMemoryLimit > 0");
        FA_THROW_IF_NOT_MSG(a.vectorsRowMajor, "bfKnn_tiling: tiling vectors is only supported in row major mode");
        vshard = (idx_t)(vectorsMemoryLimit / ((size_t)a.dims * vsz));
        FA_THROW_IF_NOT_MSG(vshard > 0, "bfKnn_tiling: vectorsMemoryLimit is too low");
    }
    const int nvs = (int)std::max<idx_t>(1, (a.numVectors + vshard - 1) / std::max<idx_t>(vshard, 1));
    for (idx_t q0 = 0; q0 < a.numQueries; q0 += qshard) {
        const idx_t nq = std::min(qshard, a.numQueries - q0);
        std::vector<float> pd((size_t)nvs * nq * a.k);
        std::vector<idx_t> pi((size_t)nvs * nq * a.k);
        std::vector<idx_t> base(nvs);
        for (int s = 0; s < nvs; s++) {
            const idx_t v0 = (idx_t)s * vshard, nv = std::min(vshard, a.numVectors - v0);
            DistanceParams t = a;
            t.queries = (const char*)a.queries + (a.queriesRowMajor ? (size_t)q0 * a.dims * qsz : (size_t)q0 * qsz);
            t.numQueries = nq;
            t.vectors = (const char*)a.vectors + (size_t)v0 * a.dims * vsz;
            t.numVectors = nv;
            t.vectorNorms = nullptr;
            t.outDistances = pd.data() + (size_t)s * nq * a.k;
            t.ignoreOutDistances = false;
            t.outIndicesType = 1;
            t.outIndices = pi.data() + (size_t)s * nq * a.k;
            FA_THROW_IF_NOT_MSG(a.queriesRowMajor || qshard == a.numQueries, "column-major queries cannot be tiled");
            bfKnn(res, t);
            base[s] = v0;
        }
        // partial top-k of the vector chunks -> top-k (ties to the lower id: equal to one untiled search)
        std::vector<float> D((size_t)nq * a.k);
        std::vector<idx_t> I((size_t)nq * a.k);
        merge_knn_results(order_metric(a.metric), nq, a.k, nvs, pd.data(), pi.data(), base.data(), D.data(), I.data());
        if (!a.ignoreOutDistances && a.outDistances) {
            FA_THROW_IF_NOT_MSG(!is_device_pointer(a.outDistances), "bfKnn_tiling: outputs of a tiled search live in CPU memory");
            memcpy(a.outDistances + (size_t)q0 * a.k, D.data(), D.size() * 4);
        }
        FA_THROW_IF_NOT_MSG(!is_device_pointer(a.outIndices), "bfKnn_tiling: outputs of a tiled search live in CPU memory");
        if (a.outIndicesType == 1) {
            memcpy((idx_t*)a.outIndices + (size_t)q0 * a.k, I.data(), I.size() * 8);
        } else {
            int32_t* o = (int32_t*)a.outIndices + (size_t)q0 * a.k;
            for (size_t i = 0; i < I.size(); i++) o[i] = (int32_t)I[i];
        }
    }
}

// choose the database split count: blocks = nsplit * ngroups should fill whole rounds of CUs
static void choose_splits(int nb, int ngroups, int num_cus, int& nsplit, int& rows_per_split, int split_cap = 64) {
    const int TR = kFlatTileRows;
    // splits of at least 2048 rows for large databases; small ones (the IVF coarse quantizer: nlist
    // centroids) go down to 4 tiles per split so that the grid still covers the chip
    const int max_split = std::max(1, nb >= 65536 ? nb / 2048 : nb / (4 * TR));
    int best = 1;
    if (max_split >= 8) {
        double best_eff = -1.0;
        for (int s = 8; s <= std::min(max_split, split_cap); s += 8) {
            long total = (long)s * ngroups;
            long rounds = (total + num_cus - 1) / num_cus;
            double eff = (double)total / (double)(rounds * num_cus);
            if (eff > best_eff + 1e-9) {
                best_eff = eff;
                best = s;
            }
        }
    } else {
        best = max_split;
    }
    nsplit = best;
    rows_per_split = (int)round_up(div_up(nb, nsplit), TR);
    // trailing splits that end up empty publish zero counts inside the kernel
}

static int reservoir_capacity(int k) {
    int cap = 64;
    while (cap < 4 * k) cap <<= 1;
    cap = std::max(cap, 64);
    if (cap < k + 32) cap = (int)round_up(k + 32, 64);
    return cap;
}

bool GpuIndexFlat::filter_applicable_(int k) const {
    return use_filter_kernel && !use_simple_kernel && db_f16_ok_ && ntotal >= filter_min_rows && k <= 1024 &&
           d >= 32 && dh_ <= 8 * kFilterSlab;
}

// split count, sampling stride of the maxima pass and segment capacity of the collect pass for a
// tile of n queries
void GpuIndexFlat::plan_filter_(int n, int k, int& geom, int& nsplit, int& tstride, int& cap, int& gcap) const {
    std::string knobs;
    for (const char* name : {"FAISS_AMD_FILTER_GEOM", "FAISS_AMD_FILTER_NSPLIT", "FAISS_AMD_FILTER_TSTRIDE"}) {
        const char* e = experiment_env(name);
        knobs += e ? e : "-";
        knobs += ';';
    }
    auto& pc = plan_cache_;
    if (pc.n == n && pc.k == k && pc.ntotal == ntotal && pc.knobs == knobs) {
        geom = pc.geom, nsplit = pc.nsplit, tstride = pc.tstride, cap = pc.cap, gcap = pc.gcap;
        return;
    }
    // large batches of d <= 128: 8 waves x 128 queries per workgroup, one workgroup per CU;
    // otherwise 4 waves x 64 queries, two workgroups per CU.  The 1024-query workgroups pay off only when they are
    // (nearly) full: measured nq = 2560 (2.5 groups) 1.10 ms against 0.94 ms with the small geometry, nq = 1280 0.77
    // against 0.53, nq = 5120 (5 full groups) 1.60 against 1.71 (profiles/r02_g_flat_batch_sizes.txt)
    const double fill2 = (double)n / ((double)div_up((size_t)n, 1024) * 1024.0);
    geom = (dh_ == kFilterSlab && n >= 2048 && fill2 >= 0.92) ? 2 : 0;
    if (const char* e = experiment_env("FAISS_AMD_FILTER_GEOM")) { // timing experiments only
        if (atoi(e) == 0 || dh_ == kFilterSlab) geom = atoi(e) ? 2 : 0;
    }
    const int qpb = flat_filter_queries_per_block(geom), cps = flat_filter_chunks_per_split(geom);
    const int ngroups = (int)div_up(n, qpb);
    const int total_tiles = (int)div_up(ntotal, kFilterTileRows);
    // S = cps * nsplit chunk maxima per query must exceed k comfortably (S >= 2.5 k keeps the expected
    // number of rows above the k-th largest maximum below ~1.3 k); nsplit * ngroups workgroups should keep
    // every resident workgroup slot busy until the end
    const int smin = (int)round_up(std::max<size_t>(8, div_up((size_t)(5 * k), 2 * cps)), 8);
    const int smax = (int)std::max<size_t>(smin, std::min<size_t>(256, (size_t)total_tiles / 4 / 8 * 8));
    const int slots = (geom == 2 ? 1 : 2) * res_->num_cus;
    // cost of a workgroup of the last query group relative to a full one: its wavefronts without a query
    // skip the MFMAs, which leaves the matrix pipes to the others (measured: 2 of 8 waves ~ 0.4)
    const int waves = geom == 2 ? 8 : 4, qpw = qpb / waves;
    const int last_waves = (int)div_up(n - (ngroups - 1) * qpb, qpw);
    const double last_cost = std::max(0.4, (double)last_waves / waves);
    int best = smin;
    double best_score = 1e30;
    for (int s = smin; s <= smax; s += 8) {
        // makespan of the launch under the dispatcher's greedy placement (workgroup b: group (b >> 3) % ngroups
        // when s is a multiple of 8, flat_filter_kernel), in units of one sweep of the database
        std::priority_queue<double, std::vector<double>, std::greater<double>> free_at;
        for (int i = 0; i < slots; ++i) free_at.push(0.0);
        double makespan = 0.0;
        const long total = (long)s * ngroups;
        for (long b = 0; b < total; ++b) {
            const int grp = (int)((b >> 3) % ngroups);
            const double t = free_at.top() + (grp == ngroups - 1 ? last_cost : 1.0) / s;
            free_at.pop();
            free_at.push(t);
            makespan = std::max(makespan, t);
        }
        // slight preference for fewer splits (measured at nq = 10k, nb = 1M: 48 splits / 1.9 rounds 3.21 ms,
        // 96 / 3.75 3.10 ms, 128 / 5.0 3.01 ms, 256 / 10.0 3.12 ms)
        const double score = makespan * (1.0 + 2e-4 * s);
        if (score < best_score) {
            best_score = score;
            best = s;
        }
    }
    nsplit = best;
    if (const char* e = experiment_env("FAISS_AMD_FILTER_NSPLIT")) nsplit = atoi(e); // timing experiments only
    const int tiles_per_split = total_tiles / nsplit;
    // sample of the maxima pass: every 8th tile when a split holds >= 96 of them (same-box A/B at the bench shape, round 4,
    // profiles/r04_d_flat_tstride_ab.txt + r04_e: stride 2 / 4 / 8 / 16 = 3.29 / 2.92 / 2.85 / 3.05 ms and 2.98 / 2.90 / 3.05 ms
    // for 4 / 8 / 16 on a second box: the halved maxima pass outweighs the doubled candidates, a quarter of it does not)
    tstride = tiles_per_split >= 96 ? 8 : tiles_per_split >= 32 ? 4 : tiles_per_split >= 16 ? 2 : 1;
    if (const char* e = experiment_env("FAISS_AMD_FILTER_TSTRIDE")) tstride = std::max(1, atoi(e)); // timing experiments only
    // expected rows above the threshold: S * -ln(1 - k/S) in the sample, tstride times that overall; the
    // re-rank kernel gathers at most 4096 of them per query, so large k samples more tiles
    const double S = (double)cps * nsplit;
    const double in_sample = S * -std::log(1.0 - std::min(0.95, (double)k / S));
    while (tstride > 1 && in_sample * tstride > 2800.0) tstride >>= 1;
    const double expect = in_sample * tstride / nsplit;
    cap = 32;
    while (cap < 4.0 * expect + 16.0) cap <<= 1;
    // LDS gather buffer of the re-rank kernel: twice the expectation + slack (more workgroups per CU than
    // with the full 4096 entries); a query with more candidates than that takes the exact path
    int kp = 1;
    while (kp < k) kp <<= 1;
    gcap = 1024;
    while (gcap < 4096 && (gcap < 2.0 * in_sample * tstride + 512.0 || gcap < kp)) gcap <<= 1;
    pc.n = n, pc.k = k, pc.ntotal = ntotal, pc.knobs = knobs;
    pc.geom = geom, pc.nsplit = nsplit, pc.tstride = tstride, pc.cap = cap, pc.gcap = gcap;
}

void GpuIndexFlat::search_tile_general_(int n, const float* xq_pad, int k, float* dD, idx_t* dI) const {
    const GpuResources& R = *res_;
    const int nb = (int)ntotal;
    SelectParams sp{};
    sp.metric = order_metric(metric_type);
    sp.nq = n;
    sp.k = k;
    sp.mode = 0;
    sp.id_base = 0;
    sp.out_dis = dD;
    sp.out_ids = dI;
    sp.nseg = 1;
    sp.seg_stride = 0;
    one_cnt_.ensure((size_t)n * 4);
    sp.seg_cnt = one_cnt_.as<uint32_t>();
    if (nb == 0) {
        HIP_CHECK(hipMemsetAsync(one_cnt_.p, 0, (size_t)n * 4, R.stream));
        launch_select_k(sp, R.stream);
        return;
    }
    DevBuf widened; // fp16 storage: a widened temporary copy of the rows
    const float* xb_rows = rows_f32_(widened);
    all_keys_.ensure((size_t)n * nb * 8);
    std::vector<uint32_t> cnt(n, (uint32_t)nb);
    HIP_CHECK(hipMemcpyAsync(one_cnt_.p, cnt.data(), (size_t)n * 4, hipMemcpyHostToDevice, R.stream));
    HIP_CHECK(hipStreamSynchronize(R.stream));
    {
        SpanGuard sg(&R, "flat_general_kernel");
        launch_flat_general(metric_type, metric_arg, xq_pad, dpad_, n, xb_rows, dpad_, nb, d,
                            all_keys_.as<unsigned long long>(), R.stream);
    }
    sp.keys = all_keys_.as<unsigned long long>();
    sp.q_stride = nb;
    {
        SpanGuard sg(&R, "select_k_kernel");
        launch_select_k(sp, R.stream);
    }
    if (use_float16_) R.sync(); // `widened` is released on return
}

void GpuIndexFlat::search_tile_(int n, const float* xq_pad, int k, float* dD, idx_t* dI, uint32_t* defer_bad) const {
    last_used_filter = false;
    last_filter_overflow = 0;
    const bool small_fused_now = !is_general_metric(metric_type) && filter_applicable_(k) && use_small_fused && !sel_active_ &&
                                 !use_float16_ && flat_small_fused_supported(metric_type, (int)ntotal, d, dh_, k);
    if (defer_bad && !small_fused_now) { // (every other path serves all its queries before it returns)
        HIP_CHECK(hipMemsetAsync(defer_bad, 0, (size_t)n * 4, res_->stream));
        defer_bad = nullptr;
    }
    if (is_general_metric(metric_type)) {
        search_tile_general_(n, xq_pad, k, dD, dI);
        return;
    }
    if (!filter_applicable_(k)) {
        search_tile_exact_(n, xq_pad, k, dD, dI);
        return;
    }
    last_used_filter = true;
    const GpuResources& R = *res_;
    const int nb = (int)ntotal;
    // ---- queries whose segments overflowed (or left the fp16 range) go through the exact fp32 scan
    auto redo_overflow = [&]() {
        if (!h_novf_) HIP_CHECK(hipHostMalloc((void**)&h_novf_, 64, hipHostMallocDefault));
        HIP_CHECK(hipMemcpyAsync(h_novf_, scal_.as<unsigned>() + 2, 4, hipMemcpyDeviceToHost, R.stream));
        R.sync();
        const unsigned novf = *h_novf_;
        last_filter_overflow = (int)novf;
        if (novf > 0) {
            ovf_q_.ensure((size_t)novf * dpad_ * 4);
            ovf_d_.ensure((size_t)novf * k * 4);
            ovf_i_.ensure((size_t)novf * k * 8);
            launch_gather_rows(xq_pad, dpad_, dpad_, ovf_list_.as<uint32_t>(), (int)novf, ovf_q_.as<float>(), R.stream);
            // the exact scan tiles its own scratch; reuse of res_keys_ is ordered on the stream
            const int tile = 16384;
            for (int i0 = 0; i0 < (int)novf; i0 += tile) {
                const int ni = std::min(tile, (int)novf - i0);
                search_tile_exact_(ni, ovf_q_.as<float>() + (size_t)i0 * dpad_, k, ovf_d_.as<float>() + (size_t)i0 * k,
                                   ovf_i_.as<idx_t>() + (size_t)i0 * k);
            }
            launch_scatter_results(ovf_d_.as<float>(), ovf_i_.as<idx_t>(), k, ovf_list_.as<uint32_t>(), (int)novf, dD, dI,
                                   R.stream);
        }
    };
    // ---- small databases (the coarse quantizer of an IVF index: nlist <= 4096 centroids, k = nprobe <= 64): maxima pass, threshold,
    // collect pass, exact re-rank and ordering in ONE launch (flat_filter.hip flat_small_fused_kernel, round 6) -- same bits
    if (small_fused_now) {
        qh_.ensure((size_t)n * dh_ * 2);
        flags_.ensure((size_t)n * 4);
        q_norm_.ensure((size_t)n * 4);
        ovf_list_.ensure((size_t)n * 4);
        {
            SpanGuard sg(&R, "convert_f16_query");
            launch_prep_queries(xq_pad, dpad_, n, d, dpad_, qh_.p, dh_, flags_.as<uint32_t>(), q_norm_.as<float>(),
                                scal_.as<unsigned>() + 2, R.stream);
        }
        if (xbo_rows_ != ntotal) { // (operand-major copy of the fp16 rows: rebuilt behind add() / reset(), <= 1 MB)
            xbo_.ensure((size_t)div_up(nb, 32) * 8192);
            launch_flat_operand_major(xbh_.p, dh_, nb, xbo_.p, R.stream);
            xbo_rows_ = ntotal;
        }
        FlatSmallParams sp{};
        sp.metric = metric_type, sp.nq = n, sp.nb = nb, sp.d = d, sp.dpad = dpad_, sp.k = k;
        sp.xbo = xbo_.p;
        sp.xqh = qh_.as<_Float16>(), sp.ldqh = dh_;
        sp.xq = xq_pad, sp.ldq = dpad_;
        sp.xqn = q_norm_.as<float>();
        sp.flags = flags_.as<uint32_t>();
        sp.xbh = xbh_.as<_Float16>(), sp.ldbh = dh_;
        sp.xbhn = xbhn_.as<float>();
        sp.xb = xb_.as<float>(), sp.ldb = dpad_;
        sp.xbn = xbn_.as<float>();
        sp.yn_max = yn_max_;
        sp.out_dis = dD, sp.out_ids = dI;
        sp.ovf_list = ovf_list_.as<uint32_t>();
        sp.ovf_cnt = scal_.as<unsigned>() + 2;
        sp.bad_out = defer_bad;
        {
            SpanGuard sg(&R, "flat_small_fused_kernel");
            launch_flat_small_fused(sp, R.stream);
        }
        if (!defer_bad) redo_overflow();
        return;
    }
    FlatFilterParams fp{};
    fp.metric = metric_type;
    int gcap = 4096;
    plan_filter_(n, k, fp.geom, fp.nsplit, fp.tstride, fp.cap, gcap);
    fp.cps = flat_filter_chunks_per_split(fp.geom);
    fp.ngroups = (int)div_up(n, flat_filter_queries_per_block(fp.geom));
    // ---- fp16 queries (+ per-query range flags), exact norms
    qh_.ensure((size_t)n * dh_ * 2);
    flags_.ensure((size_t)n * 4);
    thr_.ensure((size_t)n * 4);
    maxes_.ensure((size_t)n * fp.nsplit * fp.cps * 4);
    q_norm_.ensure((size_t)n * 4);
    ovf_list_.ensure((size_t)n * 4);
    res_keys_.ensure((size_t)n * fp.nsplit * fp.cap * 8);
    res_cnt_.ensure((size_t)n * fp.nsplit * 4);
    {
        SpanGuard sg(&R, "convert_f16_query");
        launch_prep_queries(xq_pad, dpad_, n, d, dpad_, qh_.p, dh_, flags_.as<uint32_t>(), q_norm_.as<float>(),
                            scal_.as<unsigned>() + 2, R.stream);
    }
    fp.xqh = qh_.as<_Float16>();
    fp.xqn = q_norm_.as<float>();
    fp.xbh = xbh_.as<_Float16>();
    fp.xbhn = sel_active_ ? sel_xbhn_.as<float>() : xbhn_.as<float>();
    fp.ldqh = dh_;
    fp.ldbh = dh_;
    fp.nq = n;
    fp.nb = nb;
    fp.d = d;
    fp.dh = dh_;
    fp.k = k;
    fp.yn_max = yn_max_;
    fp.maxes = maxes_.as<float>();
    fp.thr = thr_.as<float>();
    fp.res_keys = res_keys_.as<unsigned long long>();
    fp.res_cnt = res_cnt_.as<uint32_t>();
    fp.flags = flags_.as<uint32_t>();
    fp.dump = nullptr;
    fp.exact_inputs = use_float16_ ? 1 : 0;
    {
        // chunk maxima over a 1/tstride sample of the tiles -> per-query threshold
        SpanGuard sg(&R, "flat_filter_kernel_max");
        launch_flat_filter(fp, 0, R.stream);
    }
    {
        SpanGuard sg(&R, "flat_tighten_kernel");
        launch_flat_tighten(fp, R.stream);
    }
    {
        // every row above the threshold -> (query, split) segments
        SpanGuard sg(&R, "flat_filter_kernel");
        launch_flat_filter(fp, 1, R.stream);
    }
    FlatRerankParams rp{};
    rp.metric = metric_type;
    rp.nq = n;
    rp.k = k;
    rp.kp = 1;
    while (rp.kp < k) rp.kp <<= 1;
    rp.d = d;
    rp.dpad = dpad_;
    rp.nsplit = fp.nsplit;
    rp.cap = fp.cap;
    rp.gcap = gcap;
    rp.res_keys = fp.res_keys;
    rp.res_cnt = fp.res_cnt;
    rp.flags = fp.flags;
    rp.xq = xq_pad;
    rp.xqn = q_norm_.as<float>();
    rp.xb = use_float16_ ? nullptr : xb_.as<float>();
    rp.xb16 = xbh_.as<_Float16>();
    rp.xbn = xbn_.as<float>();
    rp.ldq = dpad_;
    rp.ldb = dpad_;
    rp.ldb16 = dh_;
    rp.exact_inputs = use_float16_ ? 1 : 0;
    rp.yn_max = yn_max_;
    rp.id_base = 0;
    rp.out_dis = dD;
    rp.out_ids = dI;
    rp.ovf_list = ovf_list_.as<uint32_t>();
    rp.ovf_cnt = scal_.as<unsigned>() + 2;
    {
        SpanGuard sg(&R, "flat_rerank_kernel");
        launch_flat_rerank(rp, R.stream);
    }
    redo_overflow();
}

void GpuIndexFlat::filter_scores(idx_t n, const float* x, float* scores, float* err_bound) const {
    FA_THROW_IF_NOT_MSG(db_f16_ok_ && ntotal > 0 && n > 0 && !is_general_metric(metric_type), "filter not applicable");
    std::lock_guard<std::mutex> g(mu_);
    res_->set_device();
    const GpuResources& R = *res_;
    q_pad_.ensure((size_t)n * dpad_ * 4);
    stage_padded(R, x, n, d, dpad_, q_raw_, q_pad_.as<float>());
    if (use_float16_) launch_round_f16_inplace(q_pad_.as<float>(), (int64_t)n * dpad_, R.stream);
    qh_.ensure((size_t)n * dh_ * 2);
    flags_.ensure((size_t)n * 4);
    q_norm_.ensure((size_t)n * 4);
    launch_convert_f16(q_pad_.as<float>(), dpad_, n, d, qh_.p, dh_, nullptr, flags_.as<uint32_t>(), R.stream);
    launch_l2_norms(q_pad_.as<float>(), dpad_, n, dpad_, q_norm_.as<float>(), R.stream);
    DevBuf dump;
    dump.ensure((size_t)n * ntotal * 4);
    FlatFilterParams fp{};
    fp.metric = metric_type;
    fp.xqh = qh_.as<_Float16>();
    fp.xqn = q_norm_.as<float>();
    fp.xbh = xbh_.as<_Float16>();
    fp.xbhn = xbhn_.as<float>();
    fp.ldqh = fp.ldbh = dh_;
    fp.nq = (int)n;
    fp.nb = (int)ntotal;
    fp.d = d;
    fp.dh = dh_;
    fp.geom = 0;
    fp.cps = flat_filter_chunks_per_split(fp.geom);
    fp.ngroups = (int)div_up(n, flat_filter_queries_per_block(fp.geom));
    fp.nsplit = 8;
    fp.tstride = 1;
    fp.k = 1;
    fp.cap = 32;
    fp.yn_max = yn_max_;
    fp.flags = flags_.as<uint32_t>();
    fp.dump = dump.as<float>();
    fp.exact_inputs = use_float16_ ? 1 : 0;
    launch_flat_filter(fp, 2, R.stream);
    copy_out(R, scores, dump.p, (size_t)n * ntotal * 4);
    std::vector<float> xn(n);
    HIP_CHECK(hipMemcpyAsync(xn.data(), q_norm_.p, (size_t)n * 4, hipMemcpyDeviceToHost, R.stream));
    R.sync();
    for (idx_t i = 0; i < n; i++) err_bound[i] = flat_filter_err_bound(metric_type, d, xn[i], yn_max_, use_float16_);
}

void GpuIndexFlat::search_tile_exact_(int n, const float* xq_pad, int k, float* dD, idx_t* dI) const {
    const GpuResources& R = *res_;
    const int nb = (int)ntotal;
    // fp16 storage: this path (small databases, k > 1024, overflow fallback) runs on a widened temporary copy
    DevBuf widened;
    const float* xb_rows = nb > 0 ? rows_f32_(widened) : nullptr;
    if (metric_type == METRIC_L2) {
        q_norm_.ensure((size_t)n * 4);
        SpanGuard sg(&R, "l2_norms_query");
        launch_l2_norms(xq_pad, dpad_, n, dpad_, q_norm_.as<float>(), R.stream);
    }
    SelectParams sp{};
    sp.metric = metric_type;
    sp.nq = n;
    sp.k = k;
    sp.mode = 0;
    sp.id_base = 0;
    sp.out_dis = dD;
    sp.out_ids = dI;
    if (nb == 0) {
        one_cnt_.ensure((size_t)n * 4);
        HIP_CHECK(hipMemsetAsync(one_cnt_.p, 0, (size_t)n * 4, R.stream));
        sp.keys = nullptr;
        sp.q_stride = 0;
        sp.nseg = 1;
        sp.seg_stride = 0;
        sp.seg_cnt = one_cnt_.as<uint32_t>();
        launch_select_k(sp, R.stream);
        return;
    }
    if (k == 1 && !use_simple_kernel && !sel_active_ && flat_assign_small_supported(nb, dpad_)) {
        SpanGuard sg(&R, "flat_assign_small_kernel");
        launch_flat_assign_small(metric_type, xq_pad, dpad_, n, xb_rows, xbn_.as<float>(), dpad_, nb, dpad_, dD, dI, R.stream);
        if (use_float16_) R.sync(); // `widened` is released on return
        return;
    }
    if (use_simple_kernel) {
        all_keys_.ensure((size_t)n * nb * 8);
        one_cnt_.ensure((size_t)n * 4);
        std::vector<uint32_t> cnt(n, (uint32_t)nb);
        HIP_CHECK(hipMemcpyAsync(one_cnt_.p, cnt.data(), (size_t)n * 4, hipMemcpyHostToDevice, R.stream));
        HIP_CHECK(hipStreamSynchronize(R.stream));
        {
            SpanGuard sg(&R, "flat_simple_kernel");
            launch_flat_simple(metric_type, xq_pad, q_norm_.as<float>(), dpad_, n, xb_rows,
                               xbn_.as<float>(), dpad_, nb, dpad_, all_keys_.as<unsigned long long>(),
                               R.stream);
        }
        sp.keys = all_keys_.as<unsigned long long>();
        sp.q_stride = nb;
        sp.nseg = 1;
        sp.seg_stride = 0;
        sp.seg_cnt = one_cnt_.as<uint32_t>();
        {
            SpanGuard sg(&R, "select_k_kernel");
            launch_select_k(sp, R.stream);
        }
        if (use_float16_) R.sync(); // `widened` is released on return
        return;
    }
    FlatScanParams fp{};
    fp.metric = metric_type;
    fp.xq = xq_pad;
    fp.xqn = q_norm_.as<float>();
    fp.xb = xb_rows;
    fp.xbn = xbn_.as<float>();
    if (sel_active_) { // IDSelector: excluded rows carry +inf (L2 norms) / -inf (IP start value)
        FA_THROW_IF_NOT_MSG(!use_simple_kernel, "IDSelector: not available on the scalar cross-check kernel");
        if (metric_type == METRIC_L2) fp.xbn = sel_xbn_.as<float>();
        else fp.ip_bias = sel_xbhn_.as<float>();
    }
    fp.ldq = dpad_;
    fp.ldb = dpad_;
    fp.nq = n;
    fp.nb = nb;
    fp.dpad = dpad_;
    fp.ngroups = (int)div_up(n, kFlatQueriesPerBlock);
    choose_splits(nb, fp.ngroups, R.num_cus, fp.nsplit, fp.rows_per_split);
    fp.k = k;
    fp.cap = reservoir_capacity(k);
    res_keys_.ensure((size_t)n * fp.nsplit * fp.cap * 8);
    res_cnt_.ensure((size_t)n * fp.nsplit * 4);
    fp.res_keys = res_keys_.as<unsigned long long>();
    fp.res_cnt = res_cnt_.as<uint32_t>();
    fp.dump = nullptr;
    {
        SpanGuard sg(&R, "flat_scan_kernel");
        launch_flat_scan(fp, R.stream);
    }
    sp.keys = fp.res_keys;
    sp.q_stride = (int64_t)fp.nsplit * fp.cap;
    sp.nseg = fp.nsplit;
    sp.seg_stride = fp.cap;
    sp.seg_cnt = fp.res_cnt;
    {
        SpanGuard sg(&R, "select_k_kernel");
        launch_select_k(sp, R.stream);
    }
    if (use_float16_) R.sync(); // `widened` is released on return
}

// queries per tile so that the reservoirs stay within the scratch budget
static int flat_query_tile(const GpuResources& R, int k, bool simple, idx_t nb) {
    size_t per_q;
    // (simple: every distance of a query as a key in memory -- the scalar cross-check kernel and the extra metrics)
    if (simple) per_q = (size_t)std::max<idx_t>(nb, 1) * 8;
    else per_q = (size_t)64 * std::max(reservoir_capacity(k), 2 * (k + 32)) * 8;
    size_t t = R.temp_budget_bytes / per_q;
    t = std::max<size_t>(t, 256);
    t = std::min<size_t>(t, simple ? (size_t)65280 : (size_t)1 << 20); // (simple: one grid row per query)
    return (int)(t / 256 * 256);
}

void GpuIndexFlat::search_device(int n, const float* xq_pad, int k, float* dD, idx_t* dI, uint32_t* defer_bad) const {
    // (a caller-owned coarse quantizer may serve several IVF indexes: the scratch of a search belongs to one call at a time)
    std::lock_guard<std::mutex> g(mu_);
    const int tile = flat_query_tile(*res_, k, use_simple_kernel || is_general_metric(metric_type), ntotal);
    for (int i0 = 0; i0 < n; i0 += tile) {
        int ni = std::min(tile, n - i0);
        const float* q = xq_pad + (size_t)i0 * dpad_;
        if (use_float16_) {
            // fp16 storage: the queries are converted too (FlatIndex::query, faiss/gpu/impl/FlatIndex.cu:112-135); the caller's
            // copy stays fp32 (an IVF index takes its residuals from the unrounded query)
            q_pad_.ensure((size_t)ni * dpad_ * 4);
            HIP_CHECK(hipMemcpyAsync(q_pad_.p, q, (size_t)ni * dpad_ * 4, hipMemcpyDeviceToDevice, res_->stream));
            launch_round_f16_inplace(q_pad_.as<float>(), (int64_t)ni * dpad_, res_->stream);
            q = q_pad_.as<float>();
        }
        search_tile_(ni, q, k, dD + (size_t)i0 * k, dI + (size_t)i0 * k, defer_bad ? defer_bad + i0 : nullptr);
    }
}

// IDSelector of a flat search: labels are row numbers, so the selector becomes one bit per row, and the bit becomes a
// start value no threshold admits: the filter kernel's accumulators start from xbhn (-|y|^2/2 or 0), -inf there keeps an
// excluded row out of the chunk maxima and out of the candidates; the exact scan adds xbn (L2: +inf) / ip_bias (IP: -inf).
// Every kernel then computes, for the rows that remain, exactly what it computes without a selector.
void GpuIndexFlat::prepare_selector_(const IDSelector& sel) const {
    const GpuResources& R = *res_;
    FA_THROW_IF_NOT_MSG(!use_simple_kernel, "IDSelector: not available on the scalar cross-check kernel");
    FA_THROW_IF_NOT_MSG(!is_general_metric(metric_type), "IDSelector: not available with the extra metrics");
    SelProgram prog{};
    sel.compile(prog, R.device, R.stream);
    const size_t words = div_up((size_t)std::max<idx_t>(ntotal, 1), 64);
    sel_mask_.ensure(words * 8);
    launch_selector_mask(nullptr, ntotal, 0, prog, sel_mask_.as<uint64_t>(), nullptr, R.stream);
    sel_xbhn_.ensure((size_t)(ntotal + kFilterTileRows) * 4);
    launch_mask_bias(xbhn_.as<float>(), sel_mask_.as<uint32_t>(), ntotal, kFilterTileRows, -INFINITY, -INFINITY,
                     sel_xbhn_.as<float>(), R.stream);
    if (metric_type == METRIC_L2) {
        sel_xbn_.ensure((size_t)std::max<idx_t>(ntotal, 1) * 4);
        launch_mask_bias(xbn_.as<float>(), sel_mask_.as<uint32_t>(), ntotal, 0, INFINITY, INFINITY, sel_xbn_.as<float>(),
                         R.stream);
    }
    // (everything stays on the stream: no read-back, the searches that follow are ordered behind the mask)
}

void GpuIndexFlat::search(idx_t n, const float* x, idx_t k, float* distances, idx_t* labels,
                          const SearchParameters* params) const {
    FA_THROW_IF_NOT_MSG(k >= 1 && k <= kMaxSelectionK, "k must be in [1, 2048]");
    if (n == 0) return;
    FA_THROW_IF_NOT_MSG(x && distances && labels, "null argument");
    std::lock_guard<std::mutex> g(mu_);
    res_->set_device();
    const GpuResources& R = *res_;
    struct SelScope { // the masked arrays serve this call only
        bool& on;
        ~SelScope() { on = false; }
    } sel_scope{sel_active_};
    if (params && params->sel && ntotal > 0) {
        prepare_selector_(*params->sel);
        sel_active_ = true;
    }
    if (use_paged_path(R, n, d, x, distances, labels)) {
        const idx_t page = paged_page_size(R, n, flat_query_tile(R, (int)k, use_simple_kernel || is_general_metric(metric_type), ntotal));
        paged_host_search(R, n, x, d, k, distances, labels, page, [&](idx_t ni, const float* dq, float* dD, idx_t* dI) {
            search_body_(ni, dq, k, dD, dI);
        });
        return;
    }
    search_body_(n, x, k, distances, labels);
}
void GpuIndexFlat::search_body_(idx_t n, const float* x, idx_t k, float* distances, idx_t* labels) const {
    const GpuResources& R = *res_;
    const bool out_dev_d = is_device_pointer(distances), out_dev_i = is_device_pointer(labels);
    const idx_t tile = flat_query_tile(R, (int)k, use_simple_kernel || is_general_metric(metric_type), ntotal);
    for (idx_t i0 = 0; i0 < n; i0 += tile) {
        check_interrupt();
        const int ni = (int)std::min(tile, n - i0);
        q_pad_.ensure((size_t)ni * dpad_ * 4);
        stage_padded(R, x + (size_t)i0 * d, ni, d, dpad_, q_raw_, q_pad_.as<float>());
        // fp16 storage: the queries are converted too (FlatIndex::query, faiss/gpu/impl/FlatIndex.cu:112-135)
        if (use_float16_) launch_round_f16_inplace(q_pad_.as<float>(), (int64_t)ni * dpad_, R.stream);
        float* dD = out_dev_d ? distances + (size_t)i0 * k : nullptr;
        idx_t* dI = out_dev_i ? labels + (size_t)i0 * k : nullptr;
        if (!dD) {
            out_d_.ensure((size_t)ni * k * 4);
            dD = out_d_.as<float>();
        }
        if (!dI) {
            out_i_.ensure((size_t)ni * k * 8);
            dI = out_i_.as<idx_t>();
        }
        search_tile_(ni, q_pad_.as<float>(), (int)k, dD, dI);
        if (!out_dev_d) copy_out(R, distances + (size_t)i0 * k, dD, (size_t)ni * k * 4);
        if (!out_dev_i) copy_out(R, labels + (size_t)i0 * k, dI, (size_t)ni * k * 8);
        R.sync();
    }
}

void GpuIndexFlat::pairwise_distances(idx_t n, const float* x, float* out) const {
    if (n == 0 || ntotal == 0) return;
    FA_THROW_IF_NOT_MSG(!is_general_metric(metric_type), "pairwise distances: L2 / inner product only");
    std::lock_guard<std::mutex> g(mu_);
    res_->set_device();
    const GpuResources& R = *res_;
    q_pad_.ensure((size_t)n * dpad_ * 4);
    stage_padded(R, x, n, d, dpad_, q_raw_, q_pad_.as<float>());
    if (use_float16_) launch_round_f16_inplace(q_pad_.as<float>(), (int64_t)n * dpad_, R.stream);
    q_norm_.ensure((size_t)n * 4);
    launch_l2_norms(q_pad_.as<float>(), dpad_, n, dpad_, q_norm_.as<float>(), R.stream);
    DevBuf dump;
    float* dptr = out;
    const bool dev_out = is_device_pointer(out);
    if (!dev_out) {
        dump.ensure((size_t)n * ntotal * 4);
        dptr = dump.as<float>();
    }
    DevBuf widened;
    FlatScanParams fp{};
    fp.metric = metric_type;
    fp.xq = q_pad_.as<float>();
    fp.xqn = q_norm_.as<float>();
    fp.xb = rows_f32_(widened);
    fp.xbn = xbn_.as<float>();
    fp.ldq = fp.ldb = dpad_;
    fp.nq = (int)n;
    fp.nb = (int)ntotal;
    fp.dpad = dpad_;
    fp.ngroups = (int)div_up(n, kFlatQueriesPerBlock);
    choose_splits(fp.nb, fp.ngroups, R.num_cus, fp.nsplit, fp.rows_per_split);
    fp.k = 1;
    fp.cap = 64;
    fp.dump = dptr;
    launch_flat_scan(fp, R.stream);
    if (!dev_out) copy_out(R, out, dptr, (size_t)n * ntotal * 4);
    R.sync();
}

// ====================================================================== Clustering
// Lloyd iterations exactly as the reference organises them (faiss/Clustering.cpp:255-357):
// assignment through index.search(k=1), centroid update and empty-cluster splitting on the
// host.  Random choices use our own generator, so centroids are not bit-identical to faiss
// (its own GPU tests only compare the objective, faiss/gpu/test/test_gpu_basics.py:117-133).
// split big clusters into empty ones (faiss/Clustering.cpp split_clusters); centroids on the host
void Clustering::split_clusters_(std::mt19937_64& rng, idx_t nx, std::vector<idx_t>& hassign, int first_free) {
    const float EPS = 1.f / 1024.f;
    // (frozen centroids are neither refilled nor split: faiss/Clustering.cpp:180-232 starts at k_frozen)
    for (int ci = first_free; ci < k; ci++) {
        if (hassign[ci] != 0) continue;
        int cj = first_free;
        for (long tries = 0;; ++tries) {
            double p = (hassign[cj] - 1.0) / (double)(nx - k);
            double r = (double)(rng() >> 11) * (1.0 / 9007199254740992.0);
            if (r < p) break;
            if (tries > 64L * k) {
                // (only reachable with frozen centroids owning nearly all points: take the largest free cluster)
                for (int c = first_free; c < k; c++)
                    if (hassign[c] > hassign[cj]) cj = c;
                break;
            }
            cj = cj + 1 < k ? cj + 1 : first_free;
        }
        memcpy(&centroids[(size_t)ci * d], &centroids[(size_t)cj * d], sizeof(float) * d);
        for (int j = 0; j < d; j++) {
            if (j % 2 == 0) {
                centroids[(size_t)ci * d + j] *= 1 + EPS;
                centroids[(size_t)cj * d + j] *= 1 - EPS;
            } else {
                centroids[(size_t)ci * d + j] *= 1 - EPS;
                centroids[(size_t)cj * d + j] *= 1 + EPS;
            }
        }
        hassign[ci] = hassign[cj] / 2;
        hassign[cj] -= hassign[ci];
    }
}

void Clustering::post_process_(int first_free) {
    if (spherical) {
        // fvec_renorm_L2 (faiss/utils/distances.cpp): x /= sqrt(|x|^2), zero vectors stay
        for (int c = first_free; c < k; c++) {
            float* v = &centroids[(size_t)c * d];
            float n2 = 0.f;
            for (int j = 0; j < d; j++) n2 += v[j] * v[j];
            if (n2 > 0.f) {
                const float inv = 1.f / sqrtf(n2);
                for (int j = 0; j < d; j++) v[j] *= inv;
            }
        }
    }
    if (int_centroids)
        for (size_t i = (size_t)first_free * d; i < (size_t)k * d; i++) centroids[i] = roundf(centroids[i]);
}

// faiss::Clustering::train_encoded's outer structure (faiss/Clustering.cpp:255-420): nredo runs from different random
// starts, the centroids of the run with the best final objective are kept and end up in the index
void Clustering::train(idx_t nx, const float* x_in, Index& index, int64_t ldx) {
    FA_THROW_IF_NOT_MSG(nx >= k, "need at least as many training points as clusters");
    FA_THROW_IF_NOT_MSG(nredo >= 1 && niter >= 0, "nredo must be positive");
    FA_THROW_IF_NOT_MSG(centroids.size() % (size_t)d == 0 && centroids.size() <= (size_t)k * d,
                        "initial centroids: a multiple of d floats, at most k vectors");
    const std::vector<float> init = centroids;
    if (nredo == 1) {
        train_once_(nx, x_in, index, ldx, (uint64_t)seed, init);
        return;
    }
    const bool similarity = order_metric(index.metric_type) == METRIC_INNER_PRODUCT;
    std::vector<float> best_c, best_obj;
    float best = 0.f;
    for (int redo = 0; redo < nredo; redo++) {
        train_once_(nx, x_in, index, ldx, (uint64_t)seed + 1 + (uint64_t)redo, init);
        const float o = obj.empty() ? 0.f : obj.back();
        if (verbose) printf("k-means run %d of %d: objective %g\n", redo + 1, nredo, o);
        if (redo == 0 || (similarity ? o > best : o < best)) {
            best = o;
            best_c = centroids;
            best_obj = obj;
        }
    }
    centroids = best_c;
    obj = best_obj;
    index.reset();
    index.add(k, centroids.data());
}

void Clustering::train_once_(idx_t nx, const float* x_in, Index& index, int64_t ldx, uint64_t run_seed,
                             const std::vector<float>& init) {
    last_train_on_device = false;
    if (ldx == 0) ldx = d;
    if (auto* flat = dynamic_cast<GpuIndexFlat*>(&index)) {
        if (!flat->getUseFloat16() && flat->d == d && nx < ((idx_t)1 << 31)) {
            train_device_(nx, x_in, ldx, *flat, run_seed, init);
            return;
        }
    }
    FA_THROW_IF_NOT_MSG(!is_device_pointer(x_in) && ldx == d,
                        "k-means through a generic assignment index takes dense host training data");
    train_host_(nx, x_in, index, run_seed, init);
}

void Clustering::train_host_(idx_t nx, const float* x_in, Index& index, uint64_t run_seed, const std::vector<float>& init) {
    std::mt19937_64 rng(run_seed);
    // ---- subsample (faiss/Clustering.cpp subsample_training_set)
    std::vector<float> sub;
    const float* x = x_in;
    if (nx > (idx_t)k * max_points_per_centroid) {
        idx_t ns = (idx_t)k * max_points_per_centroid;
        std::vector<idx_t> perm(nx);
        std::iota(perm.begin(), perm.end(), 0);
        for (idx_t i = 0; i < ns; i++) {
            idx_t j = i + (idx_t)(rng() % (uint64_t)(nx - i));
            std::swap(perm[i], perm[j]);
        }
        sub.resize((size_t)ns * d);
        for (idx_t i = 0; i < ns; i++)
            memcpy(&sub[(size_t)i * d], x_in + (size_t)perm[i] * d, sizeof(float) * d);
        x = sub.data();
        nx = ns;
    }
    // ---- init: the given centroids, then distinct random points for the rest
    const int n_init = (int)(init.size() / (size_t)d);
    const int first_free = frozen_centroids ? n_init : 0;
    centroids.assign((size_t)k * d, 0.f);
    if (n_init) memcpy(centroids.data(), init.data(), init.size() * sizeof(float));
    {
        std::vector<idx_t> perm(nx);
        std::iota(perm.begin(), perm.end(), 0);
        for (int i = 0; i < k - n_init; i++) {
            idx_t j = i + (idx_t)(rng() % (uint64_t)(nx - i));
            std::swap(perm[i], perm[j]);
            memcpy(&centroids[(size_t)(n_init + i) * d], x + (size_t)perm[i] * d, sizeof(float) * d);
        }
    }
    post_process_(first_free);
    std::vector<idx_t> assign(nx);
    std::vector<float> dis(nx);
    std::vector<double> sums((size_t)k * d);
    std::vector<idx_t> hassign(k);
    obj.clear();
    for (int it = 0; it < niter; it++) {
        check_interrupt();
        index.reset();
        index.add(k, centroids.data());
        index.search(nx, x, 1, dis.data(), assign.data());
        double o = 0;
        for (idx_t i = 0; i < nx; i++) o += dis[i];
        obj.push_back((float)o);
        // ---- update (faiss/Clustering.cpp:307-324 compute_centroids)
        std::fill(sums.begin(), sums.end(), 0.0);
        std::fill(hassign.begin(), hassign.end(), 0);
        for (idx_t i = 0; i < nx; i++) {
            idx_t c = assign[i];
            if (c < 0) continue;
            hassign[c]++;
            const float* xi = x + (size_t)i * d;
            double* s = &sums[(size_t)c * d];
            for (int j = 0; j < d; j++) s[j] += xi[j];
        }
        for (int c = first_free; c < k; c++) {
            if (hassign[c] == 0) continue;
            for (int j = 0; j < d; j++)
                centroids[(size_t)c * d + j] = (float)(sums[(size_t)c * d + j] / (double)hassign[c]);
        }
        split_clusters_(rng, nx, hassign, first_free);
        post_process_(first_free);
        if (verbose) printf("  k-means iteration %d objective %g\n", it, o);
    }
    index.reset();
    index.add(k, centroids.data());
}

// The same loop with everything but the random choices on the device.  The generator is drawn from in the same
// order as above (subsample, initial points, splits) and the update adds in the same order, so both paths give
// the same centroids bit for bit (tests/test_gpu_parity.py::test_kmeans_device_matches_host_loop).
void Clustering::train_device_(idx_t nx, const float* x_in, int64_t ldx, GpuIndexFlat& flat, uint64_t run_seed,
                               const std::vector<float>& init) {
    last_train_on_device = true;
    auto res = flat.resources();
    const GpuResources& R = *res;
    R.set_device();
    const int dp = flat.dpad();
    const bool x_dev = is_device_pointer(x_in);
    FA_THROW_IF_NOT_MSG(x_dev || ldx == d, "strided training rows must live on the device");
    std::mt19937_64 rng(run_seed);
    const int n_init = (int)(init.size() / (size_t)d);
    const int first_free = frozen_centroids ? n_init : 0;
    const bool post = spherical || int_centroids;
    DevBuf xd, raw, sel, tmp;
    // ---- subsample (faiss/Clustering.cpp subsample_training_set)
    if (nx > (idx_t)k * max_points_per_centroid) {
        const idx_t ns = (idx_t)k * max_points_per_centroid;
        std::vector<uint32_t> perm((size_t)nx);
        std::iota(perm.begin(), perm.end(), 0u);
        for (idx_t i = 0; i < ns; i++) {
            idx_t j = i + (idx_t)(rng() % (uint64_t)(nx - i));
            std::swap(perm[i], perm[j]);
        }
        xd.ensure((size_t)ns * dp * 4);
        if (x_dev) {
            sel.ensure((size_t)ns * 4);
            HIP_CHECK(hipMemcpyAsync(sel.p, perm.data(), (size_t)ns * 4, hipMemcpyHostToDevice, R.stream));
            tmp.ensure((size_t)ns * d * 4);
            launch_gather_rows(x_in, ldx, d, sel.as<uint32_t>(), (int)ns, tmp.as<float>(), R.stream);
            launch_pad_rows(tmp.as<float>(), d, ns, d, xd.as<float>(), dp, dp, R.stream);
            R.sync();
        } else {
            std::vector<float> sub((size_t)ns * d);
            for (idx_t i = 0; i < ns; i++) memcpy(&sub[(size_t)i * d], x_in + (size_t)perm[i] * d, sizeof(float) * d);
            stage_padded(R, sub.data(), ns, d, dp, raw, xd.as<float>());
            R.sync();
        }
        nx = ns;
    } else {
        xd.ensure((size_t)nx * dp * 4);
        if (x_dev) launch_pad_rows(x_in, ldx, nx, d, xd.as<float>(), dp, dp, R.stream);
        else stage_padded(R, x_in, nx, d, dp, raw, xd.as<float>());
        R.sync();
    }
    raw.release();
    tmp.release();
    // ---- init: the given centroids, then distinct random points for the rest
    DevBuf cen, frozen; // [k][d] dense
    cen.ensure((size_t)k * d * 4);
    centroids.resize((size_t)k * d);
    {
        std::vector<uint32_t> perm((size_t)nx);
        std::iota(perm.begin(), perm.end(), 0u);
        for (int i = 0; i < k - n_init; i++) {
            idx_t j = i + (idx_t)(rng() % (uint64_t)(nx - i));
            std::swap(perm[i], perm[j]);
        }
        if (n_init) HIP_CHECK(hipMemcpyAsync(cen.p, init.data(), init.size() * 4, hipMemcpyHostToDevice, R.stream));
        if (k > n_init) {
            sel.ensure((size_t)(k - n_init) * 4);
            HIP_CHECK(hipMemcpyAsync(sel.p, perm.data(), (size_t)(k - n_init) * 4, hipMemcpyHostToDevice, R.stream));
            launch_gather_rows(xd.as<float>(), dp, d, sel.as<uint32_t>(), k - n_init, cen.as<float>() + (size_t)n_init * d,
                               R.stream);
        }
        R.sync();
        if (post) {
            HIP_CHECK(hipMemcpy(centroids.data(), cen.p, (size_t)k * d * 4, hipMemcpyDeviceToHost));
            post_process_(first_free);
            HIP_CHECK(hipMemcpy(cen.p, centroids.data(), (size_t)k * d * 4, hipMemcpyHostToDevice));
        }
        if (first_free) {
            frozen.ensure((size_t)first_free * d * 4);
            HIP_CHECK(hipMemcpy(frozen.p, cen.p, (size_t)first_free * d * 4, hipMemcpyDeviceToDevice));
        }
    }
    // a chunk per wavefront of the rank kernel (which walks its chunk 64 points at a time): about 512 of them
    int chunk = 256;
    while (div_up(nx, chunk) > 512) chunk *= 2;
    if (const char* e = experiment_env("FAISS_AMD_KMEANS_CHUNK")) chunk = std::max(64, atoi(e)); // timing experiments only
    const int nchunks = (int)div_up(nx, chunk);
    DevBuf dis, lab, hist, cnt, zero, start, dest, order;
    dis.ensure((size_t)nx * 4);
    lab.ensure((size_t)nx * 8);
    dest.ensure((size_t)nx * 8);
    order.ensure((size_t)nx * 4);
    hist.ensure((size_t)nchunks * k * 4);
    cnt.ensure((size_t)k * 4);
    zero.ensure((size_t)k * 4);
    start.ensure((size_t)(k + 1) * 8);
    HIP_CHECK(hipMemsetAsync(zero.p, 0, (size_t)k * 4, R.stream));
    std::vector<float> hdis((size_t)nx);
    std::vector<uint32_t> hcnt((size_t)k);
    std::vector<idx_t> hassign(k);
    obj.clear();
    for (int it = 0; it < niter; it++) {
        check_interrupt();
        flat.reset();
        flat.add(k, cen.as<float>());
        flat.search_device((int)nx, xd.as<float>(), 1, dis.as<float>(), lab.as<idx_t>());
        HIP_CHECK(hipMemcpyAsync(hdis.data(), dis.p, (size_t)nx * 4, hipMemcpyDeviceToHost, R.stream));
        // ---- update (faiss/Clustering.cpp:307-324 compute_centroids)
        HIP_CHECK(hipMemsetAsync(hist.p, 0, (size_t)nchunks * k * 4, R.stream));
        launch_ivf_histogram(lab.as<int64_t>(), nx, k, chunk, hist.as<uint32_t>(), R.stream);
        launch_ivf_chunk_scan(hist.as<uint32_t>(), nchunks, k, zero.as<uint32_t>(), cnt.as<uint32_t>(), R.stream);
        HIP_CHECK(hipMemcpyAsync(hcnt.data(), cnt.p, (size_t)k * 4, hipMemcpyDeviceToHost, R.stream));
        launch_exclusive_scan(cnt.as<uint32_t>(), k, start.as<int64_t>(), R.stream);
        launch_ivf_rank(lab.as<int64_t>(), nx, k, chunk, hist.as<uint32_t>(), start.as<int64_t>(), dest.as<int64_t>(),
                        R.stream);
        launch_invert_dest(dest.as<int64_t>(), nx, order.as<uint32_t>(), R.stream);
        launch_kmeans_update(xd.as<float>(), dp, d, order.as<uint32_t>(), start.as<int64_t>(), cnt.as<uint32_t>(), k,
                             cen.as<float>(), R.stream);
        // frozen centroids: what the update wrote over them is undone
        if (first_free)
            HIP_CHECK(hipMemcpyAsync(cen.p, frozen.p, (size_t)first_free * d * 4, hipMemcpyDeviceToDevice, R.stream));
        R.sync();
        double o = 0;
        for (idx_t i = 0; i < nx; i++) o += hdis[i];
        obj.push_back((float)o);
        bool any_empty = false;
        for (int c = 0; c < k; c++) {
            hassign[c] = hcnt[c];
            any_empty |= c >= first_free && hcnt[c] == 0;
        }
        if (any_empty || post) {
            HIP_CHECK(hipMemcpy(centroids.data(), cen.p, (size_t)k * d * 4, hipMemcpyDeviceToHost));
            if (any_empty) split_clusters_(rng, nx, hassign, first_free);
            post_process_(first_free);
            HIP_CHECK(hipMemcpy(cen.p, centroids.data(), (size_t)k * d * 4, hipMemcpyHostToDevice));
        }
        if (verbose) printf("  k-means iteration %d objective %g\n", it, o);
    }
    HIP_CHECK(hipMemcpy(centroids.data(), cen.p, (size_t)k * d * 4, hipMemcpyDeviceToHost));
    flat.reset();
    flat.add(k, cen.as<float>());
}

// ====================================================================== GpuIndexIVF
GpuIndexIVF::GpuIndexIVF(std::shared_ptr<GpuResources> res, int dims, int metric, int nlist_, GpuIndexFlat* coarse_quantizer,
                         bool coarse_f16, int indices_options_)
        : Index(dims, metric), nlist(nlist_), indices_options(indices_options_), res_(std::move(res)) {
    FA_THROW_IF_NOT_MSG(nlist > 0, "nlist must be positive");
    FA_THROW_IF_NOT_MSG(metric_supported(1, metric), "unsupported metric type (reference: faiss/gpu/GpuIndexIVF.cu:35-37)");
    FA_THROW_IF_NOT_MSG(indices_options >= 0 && indices_options <= 3, "invalid indicesOptions");
    dpad_ = (int)round_up(dims, 8);
    if (coarse_quantizer) {
        // the caller's quantizer (faiss/gpu/GpuIndexIVF.cu:41-70 verifyIVFSettings_): same dimension, same device, not owned
        FA_THROW_IF_NOT_MSG(coarse_quantizer->d == dims, "the coarse quantizer's dimension differs from the index's");
        FA_THROW_IF_NOT_MSG(coarse_quantizer->device() == res_->device, "the coarse quantizer lives on another device");
        FA_THROW_IF_NOT_MSG(coarse_quantizer->metric_type == METRIC_L2 || coarse_quantizer->metric_type == METRIC_INNER_PRODUCT,
                            "the coarse quantizer must be an L2 / inner-product flat index");
        FA_THROW_IF_NOT_MSG(coarse_quantizer->ntotal == 0 || coarse_quantizer->ntotal == nlist,
                            "the coarse quantizer must be empty or hold exactly nlist centroids");
        quantizer = coarse_quantizer;
        own_fields = false;
    } else {
        quantizer = new GpuIndexFlat(res_, dims, metric, coarse_f16);
        // the coarse quantizer holds only nlist rows: let it take the fp16 filter + exact re-rank path from
        // 2048 centroids on (bit-identical results; the fp32 scan's per-split reservoirs bootstrap poorly
        // on a few thousand rows)
        quantizer->filter_min_rows = 2048;
    }
    is_trained = false;
    list_len_.assign(nlist, 0);
    list_cap_.assign(nlist, 0);
    list_start_.assign(nlist, 0);
}
GpuIndexIVF::~GpuIndexIVF() {
    (void)hipSetDevice(res_->device);
    if (h_lm_) (void)hipHostFree(h_lm_);
    if (own_fields) delete quantizer;
}

void GpuIndexIVF::update_is_trained_() {
    is_trained = quantizer->ntotal == nlist && extra_trained_();
    cent_dirty_ = true;
}
// end of a derived constructor: a caller-owned quantizer that already holds its centroids makes the coarse level trained
// (faiss/gpu/GpuIndexIVF.cu:60-70: is_trained = quantizer->is_trained && quantizer->ntotal == nlist)
void GpuIndexIVF::adopt_quantizer_() {
    if (quantizer->ntotal != nlist) return;
    res_->set_device();
    update_is_trained_();
    ensure_arena_(64);
    upload_list_tables_();
}
const float* GpuIndexIVF::centroids_dev_() const {
    if (!quantizer->getUseFloat16()) return centroids_dev_();
    if (cent_dirty_) {
        FA_THROW_IF_NOT_MSG(quantizer->ntotal == nlist, "the coarse quantizer does not hold nlist centroids");
        cent_f32_.ensure((size_t)nlist * dpad_ * 4);
        quantizer->rows_to_f32(cent_f32_.as<float>());
        cent_dirty_ = false;
    }
    return cent_f32_.as<float>();
}

void GpuIndexIVF::upload_list_tables_() {
    d_list_len_.ensure((size_t)nlist * 4);
    d_list_start_.ensure((size_t)nlist * 8);
    HIP_CHECK(hipMemcpyAsync(d_list_len_.p, list_len_.data(), (size_t)nlist * 4, hipMemcpyHostToDevice,
                             res_->stream));
    HIP_CHECK(hipMemcpyAsync(d_list_start_.p, list_start_.data(), (size_t)nlist * 8, hipMemcpyHostToDevice,
                             res_->stream));
    res_->sync();
}

// the arena buffers hold at least `rows` rows (geometric growth; contents of the rows handed out so far kept)
void GpuIndexIVF::ensure_arena_(int64_t rows) {
    rows = std::max<int64_t>(rows, 64);
    if (rows <= arena_cap_rows_ && arena_.p) return;
    int64_t ncap = std::max<int64_t>(rows, arena_cap_rows_ + arena_cap_rows_ / 4);
    ncap = (int64_t)round_up((size_t)ncap, 64);
    const size_t keep = (size_t)arena_rows_;
    // DevBuf::ensure grows to max(bytes, 1.5 cap): ask for exactly ncap rows of each array (+ one tile of rows nobody
    // owns behind the last: the list-major scan fetches whole 64-row tiles without clamping at the end of a list)
    const size_t pad = 128;
    arena_.ensure(((size_t)ncap + pad) * code_bytes_, keep * code_bytes_, res_->stream);
    arena_ids_.ensure((size_t)ncap * 8, keep * 8, res_->stream);
    if (use_t2_) arena_t2_.ensure((size_t)ncap * 4, keep * 4, res_->stream);
    if (use_rn_) arena_rn_.ensure(((size_t)ncap + pad) * 4, keep * 4, res_->stream);
    arena_cap_rows_ = ncap;
}

void GpuIndexIVF::resident_bytes(size_t* lists, size_t* shadow) const {
    std::lock_guard<std::mutex> g(mu_);
    if (lists) *lists = arena_.cap + arena_ids_.cap + arena_t2_.cap + arena_rn_.cap;
    if (shadow) *shadow = lmf_shadow_bytes_();
}
void GpuIndexIVF::arena_stats(int64_t* used, int64_t* holes, int64_t* allocated) const {
    std::lock_guard<std::mutex> g(mu_);
    if (used) *used = arena_rows_;
    if (holes) *holes = hole_rows_;
    if (allocated) *allocated = arena_cap_rows_;
}

void GpuIndexIVF::reset() {
    std::lock_guard<std::mutex> g(mu_);
    res_->set_device();
    shadow_dirty_ = true;
    ntotal = 0;
    nstored_ = 0;
    arena_rows_ = 0;
    hole_rows_ = 0;
    std::fill(list_len_.begin(), list_len_.end(), 0);
    std::fill(list_cap_.begin(), list_cap_.end(), 0);
    std::fill(list_start_.begin(), list_start_.end(), 0);
    ensure_arena_(64);
    upload_list_tables_();
}

void GpuIndexIVF::set_centroids(const float* centroids) {
    FA_THROW_IF_NOT_MSG(centroids, "null centroids");
    std::lock_guard<std::mutex> g(mu_);
    res_->set_device();
    quantizer->reset();
    quantizer->add(nlist, centroids);
    update_is_trained_();
    lmf_quant_dirty_ = true;
    ensure_arena_(64);
    upload_list_tables_();
    if (nstored_ > 0 && is_trained) lists_changed_();
}

void GpuIndexIVF::train(idx_t n, const float* x) {
    if (is_trained && quantizer->ntotal == nlist) return; // reference: GpuIndexIVF.cu trainQuantizer_
    lmf_quant_dirty_ = true;
    FA_THROW_IF_NOT_MSG(n > 0 && x, "empty training set");
    res_->set_device();
    if (quantizer->ntotal != nlist) {
        Clustering clus(d, nlist);
        static_cast<ClusteringParameters&>(clus) = cp;
        clus.niter = cp_niter;
        clus.seed = cp_seed;
        clus.verbose = verbose || cp.verbose;
        // the quantizer itself is the assignment index, exactly like GpuIndexIVF::trainQuantizer_
        // (faiss/gpu/GpuIndexIVF.cu:508-538)
        clus.train(n, x, *quantizer);
        FA_THROW_IF_NOT(quantizer->ntotal == nlist);
    }
    {
        // residual training (IVFPQ) works on padded device copies
        std::lock_guard<std::mutex> g(mu_);
        cent_dirty_ = true;
        if (!extra_trained_()) {
            DevBuf xpad;
            xpad.ensure((size_t)n * dpad_ * 4);
            stage_padded(*res_, x, n, d, dpad_, q_raw_, xpad.as<float>());
            res_->sync();
            train_residual_(n, xpad.as<float>());
        }
        update_is_trained_();
        ensure_arena_(64);
        upload_list_tables_();
    }
}

void GpuIndexIVF::add(idx_t n, const float* x) {
    // ids are generated sequentially when absent, from the number of vectors add() has been given so far
    // (reference: faiss/gpu/GpuIndex.cu:137-144; ntotal counts attempted vectors, GpuIndexIVF.cu:293-298)
    add_core_(n, x, nullptr);
}
void GpuIndexIVF::add_with_ids(idx_t n, const float* x, const idx_t* xids) {
    FA_THROW_IF_NOT_MSG(n == 0 || xids, "null ids");
    add_core_(n, x, xids);
}

// Lists that outgrow their slack move to fresh rows at the end of the arena (geometric capacity); the abandoned
// range becomes a hole that compact_() reclaims once holes and slack exceed 60 % of the stored rows.  Per call O(nlist) host work
// + the moved bytes (amortised O(1) per added vector), instead of round 1's rebuild of the whole arena.
void GpuIndexIVF::grow_lists_(const std::vector<uint32_t>& new_len, const std::vector<double>* est) {
    std::vector<IvfMoveJob> jobs;
    int64_t extra = 0;
    const int64_t G = granule_;
    for (int l = 0; l < nlist; l++) {
        if (new_len[l] <= list_cap_[l]) continue;
        double want = (double)new_len[l];
        if (list_cap_[l] > 0) want = std::max(want, 1.5 * (double)list_cap_[l]);
        if (est) want = std::max(want, 1.03 * (*est)[l]);
        const int64_t ncap = (int64_t)round_up((size_t)std::ceil(want), (size_t)G);
        FA_THROW_IF_NOT_MSG(ncap < ((int64_t)1 << 32), "inverted list too long");
        if (list_len_[l] > 0) jobs.push_back({list_start_[l], arena_rows_ + extra, (int64_t)round_up(list_len_[l], (size_t)G)});
        if (!moved_.empty()) moved_[l] = 1;
        hole_rows_ += list_cap_[l];
        list_start_[l] = arena_rows_ + extra;
        list_cap_[l] = (uint32_t)ncap;
        extra += ncap;
    }
    if (extra == 0) return;
    const GpuResources& R = *res_;
    if (moved_.empty()) shadow_dirty_ = true; // (add_core_ patches a live copy of the lists: it passes moved_)
    ensure_arena_(arena_rows_ + extra); // (keeps rows [0, arena_rows_): the jobs' sources)
    arena_rows_ += extra;
    if (!jobs.empty()) {
        a_jobs_.ensure(jobs.size() * sizeof(IvfMoveJob));
        HIP_CHECK(hipMemcpyAsync(a_jobs_.p, jobs.data(), jobs.size() * sizeof(IvfMoveJob), hipMemcpyHostToDevice,
                                 R.stream));
        const IvfMoveJob* dj = a_jobs_.as<IvfMoveJob>();
        launch_ivf_move(arena_.as<uint8_t>(), arena_.as<uint8_t>(), dj, (int)jobs.size(), (int)code_bytes_, R.stream);
        launch_ivf_move(arena_ids_.as<uint8_t>(), arena_ids_.as<uint8_t>(), dj, (int)jobs.size(), 8, R.stream);
        if (use_t2_)
            launch_ivf_move(arena_t2_.as<uint8_t>(), arena_t2_.as<uint8_t>(), dj, (int)jobs.size(), 4, R.stream);
        if (use_rn_)
            launch_ivf_move(arena_rn_.as<uint8_t>(), arena_rn_.as<uint8_t>(), dj, (int)jobs.size(), 4, R.stream);
    }
    HIP_CHECK(hipMemcpyAsync(d_list_start_.p, list_start_.data(), (size_t)nlist * 8, hipMemcpyHostToDevice, R.stream));
    R.sync(); // jobs / list_start_ host vectors are read by the copies above
}

void GpuIndexIVF::reserveMemory(size_t numVecs) {
    std::lock_guard<std::mutex> g(mu_);
    res_->set_device();
    // the vectors + one granule of rounding per list + the growth margin a relocated list takes (1.5 x)
    const int64_t want = (int64_t)((double)numVecs * 1.25) + (int64_t)nlist * granule_ + 64;
    ensure_arena_(std::max<int64_t>(want, arena_rows_));
    res_->sync();
}
size_t GpuIndexIVF::reclaimMemory() {
    std::lock_guard<std::mutex> g(mu_);
    res_->set_device();
    const size_t row_bytes = code_bytes_ + 8 + (use_t2_ ? 4 : 0) + (use_rn_ ? 4 : 0);
    // (+ the 128 rows of padding behind the codes and the norms, ensure_arena_)
    const size_t pad_bytes = 128 * (code_bytes_ + (use_rn_ ? 4 : 0));
    size_t before = (size_t)arena_cap_rows_ * row_bytes + (arena_.p ? pad_bytes : 0);
    // add-path scratch and the per-search scratch of the list-major scans (key segments sized up to the temp budget,
    // plan tables, granule minima): all of it is re-grown on demand
    for (DevBuf* b : {&a_xpad_, &a_lab_, &a_dis_, &a_dest_, &a_ids_, &a_hist_, &a_newlen_, &a_jobs_, &lm_prefix_, &lm_p0_,
                      &lm_cnt_, &lm_bucket_, &lm_bstart_, &lm_pairs_, &lm_items_, &lm_bounds_, &lm_thr_, &lm_keys_, &lm_ovf_,
                      &lm_qn_, &lm_prefixg_, &lm_gmin_, &lm_thrf_, &lm_candpr_, &lm_q16_, &lm_qflags_, &lm_xnb_, &lm_pqgrid_, &lm_pair16_, &lm_pairxh_, &lm_errf_, &lm_an_, &lm_rowbase_,
                      &part_keys_, &part_cnt_, &keys_}) {
        before += b->cap;
        b->release();
    }
    // the filter sweeps' copies of the lists (fp16 shadow / operand-major codes): derived data, rebuilt on demand
    const size_t shadow = lmf_release_();
    before += shadow;
    if (arena_.p) compact_(true);
    const size_t after = (size_t)arena_cap_rows_ * row_bytes + (arena_.p ? pad_bytes : 0);
    return before > after ? before - after : 0;
}
void GpuIndexIVF::updateQuantizer() {
    std::lock_guard<std::mutex> g(mu_);
    res_->set_device();
    FA_THROW_IF_NOT_MSG(quantizer->ntotal == 0 || quantizer->ntotal == nlist, "the coarse quantizer must hold nlist centroids");
    lmf_quant_dirty_ = true;
    update_is_trained_();
    if (quantizer->ntotal == nlist) {
        ensure_arena_(64);
        upload_list_tables_();
        if (nstored_ > 0 && is_trained) lists_changed_();
    }
}

// rebuild the arena without holes (lists in id order, 1/8 slack each)
void GpuIndexIVF::compact_(bool tight) {
    const GpuResources& R = *res_;
    shadow_dirty_ = true;
    const int64_t G = granule_;
    std::vector<IvfMoveJob> jobs;
    std::vector<int64_t> nstart(nlist);
    std::vector<uint32_t> ncap(nlist);
    int64_t acc = 0;
    for (int l = 0; l < nlist; l++) {
        nstart[l] = acc;
        ncap[l] = list_len_[l] ? (uint32_t)round_up((size_t)list_len_[l] + (tight ? 0 : list_len_[l] / 8), (size_t)G) : 0u;
        if (list_len_[l]) jobs.push_back({list_start_[l], acc, (int64_t)round_up(list_len_[l], (size_t)G)});
        acc += ncap[l];
    }
    DevBuf na, ni, nt, nr;
    const int64_t rows = std::max<int64_t>(acc, 64);
    na.ensure(((size_t)rows + 128) * code_bytes_); // (+ one tile: see ensure_arena_)
    ni.ensure((size_t)rows * 8);
    if (use_t2_) nt.ensure((size_t)rows * 4);
    if (use_rn_) nr.ensure(((size_t)rows + 128) * 4);
    if (!jobs.empty()) {
        a_jobs_.ensure(jobs.size() * sizeof(IvfMoveJob));
        HIP_CHECK(hipMemcpyAsync(a_jobs_.p, jobs.data(), jobs.size() * sizeof(IvfMoveJob), hipMemcpyHostToDevice,
                                 R.stream));
        const IvfMoveJob* dj = a_jobs_.as<IvfMoveJob>();
        launch_ivf_move(arena_.as<uint8_t>(), na.as<uint8_t>(), dj, (int)jobs.size(), (int)code_bytes_, R.stream);
        launch_ivf_move(arena_ids_.as<uint8_t>(), ni.as<uint8_t>(), dj, (int)jobs.size(), 8, R.stream);
        if (use_t2_) launch_ivf_move(arena_t2_.as<uint8_t>(), nt.as<uint8_t>(), dj, (int)jobs.size(), 4, R.stream);
        if (use_rn_) launch_ivf_move(arena_rn_.as<uint8_t>(), nr.as<uint8_t>(), dj, (int)jobs.size(), 4, R.stream);
    }
    R.sync();
    std::swap(arena_.p, na.p);
    std::swap(arena_.cap, na.cap);
    std::swap(arena_ids_.p, ni.p);
    std::swap(arena_ids_.cap, ni.cap);
    if (use_t2_) {
        std::swap(arena_t2_.p, nt.p);
        std::swap(arena_t2_.cap, nt.cap);
    }
    if (use_rn_) {
        std::swap(arena_rn_.p, nr.p);
        std::swap(arena_rn_.cap, nr.cap);
    }
    list_start_ = nstart;
    list_cap_ = ncap;
    arena_rows_ = acc;
    arena_cap_rows_ = rows;
    hole_rows_ = 0;
    upload_list_tables_();
}

// add path, page by page like the reference (faiss/gpu/GpuIndex.cu:197-254 addPaged_): per page the vectors are
// staged once, assigned to their lists (k = 1 search on the quantizer), ranked inside their lists by a stable
// counting sort on the device (ivf_kernels.hip) and encoded / scattered into the lists' slack.  The host sees
// nlist list lengths per page, never a per-vector array.
void GpuIndexIVF::add_core(idx_t n, const float* x, const idx_t* xids, const idx_t* precomputed_idx) {
    FA_THROW_IF_NOT_MSG(n == 0 || precomputed_idx, "precomputed IVF assignments must not be null");
    add_core_(n, x, xids, precomputed_idx);
}
void GpuIndexIVF::add_core_(idx_t n, const float* x, const idx_t* xids, const idx_t* assign) {
    FA_THROW_IF_NOT_MSG(is_trained, "index must be trained before adding vectors");
    if (n == 0) return;
    FA_THROW_IF_NOT_MSG(x, "null argument");
    std::lock_guard<std::mutex> g(mu_);
    res_->set_device();
    const GpuResources& R = *res_;
    const idx_t page = std::max<idx_t>(1, std::min<idx_t>(((idx_t)512 << 20) / ((idx_t)dpad_ * 4), 1 << 20));
    // the filter sweeps' copy of the lists: a live one is kept up to date page by page (lmf_patch_), otherwise it is (re)built
    // as a whole by the first list-major search that wants it
    const bool patch = use_filter_shadow && !shadow_dirty_ && lmf_shadow_bytes_() > 0 && lmf_capable_();
    if (!patch) shadow_dirty_ = true;
    struct MovedScope {
        std::vector<uint8_t>& m;
        ~MovedScope() { m.clear(); }
    } moved_scope{moved_};
    const int chunk = 2048;
    const idx_t pn = std::min(page, n);
    a_xpad_.ensure((size_t)pn * dpad_ * 4);
    a_lab_.ensure((size_t)pn * 8);
    a_dis_.ensure((size_t)pn * 4);
    a_dest_.ensure((size_t)pn * 8);
    a_ids_.ensure((size_t)pn * 8);
    a_newlen_.ensure((size_t)nlist * 4);
    a_hist_.ensure(div_up(pn, chunk) * (size_t)nlist * 4);
    const idx_t id_base = ntotal;
    const std::vector<uint32_t> len0(list_len_);
    std::vector<uint32_t> new_len(nlist);
    std::vector<double> est(nlist);
    for (idx_t i0 = 0; i0 < n; i0 += page) {
        check_interrupt();
        const int ni = (int)std::min(page, n - i0);
        const int nchunks = (int)div_up(ni, chunk);
        stage_padded(R, x + (size_t)i0 * d, ni, d, dpad_, q_raw_, a_xpad_.as<float>());
        // ---- coarse assignment (NaN vectors get label -1 and are skipped, like the reference), or the caller's
        if (assign) {
            HIP_CHECK(hipMemcpyAsync(a_lab_.p, assign + i0, (size_t)ni * 8, hipMemcpyDefault, R.stream));
            launch_ivf_sanitize_assign(a_lab_.as<idx_t>(), ni, nlist, R.stream);
        } else {
            quantizer->search_device(ni, a_xpad_.as<float>(), 1, a_dis_.as<float>(), a_lab_.as<idx_t>());
        }
        if (xids) {
            HIP_CHECK(hipMemcpyAsync(a_ids_.p, xids + i0, (size_t)ni * 8, hipMemcpyDefault, R.stream));
        } else {
            launch_iota_i64(a_ids_.as<int64_t>(), ni, id_base + i0, R.stream);
        }
        // ---- list lengths after this page
        HIP_CHECK(hipMemsetAsync(a_hist_.p, 0, (size_t)nchunks * nlist * 4, R.stream));
        launch_ivf_histogram(a_lab_.as<int64_t>(), ni, nlist, chunk, a_hist_.as<uint32_t>(), R.stream);
        launch_ivf_chunk_scan(a_hist_.as<uint32_t>(), nchunks, nlist, d_list_len_.as<uint32_t>(), a_newlen_.as<uint32_t>(),
                              R.stream);
        HIP_CHECK(hipMemcpyAsync(new_len.data(), a_newlen_.p, (size_t)nlist * 4, hipMemcpyDeviceToHost, R.stream));
        R.sync();
        // ---- room: a list that has to move takes the length this call is expected to leave it with
        const double scale = (double)n / (double)(i0 + ni);
        idx_t added = 0;
        for (int l = 0; l < nlist; l++) {
            est[l] = (double)len0[l] + ((double)new_len[l] - (double)len0[l]) * scale;
            added += (idx_t)new_len[l] - (idx_t)list_len_[l];
        }
        // a live copy of the lists is patched per page; from the moment the lists start to change until lmf_patch_ has
        // returned the copy counts as dirty, so that ANY exception in between (HIP error, OOM, ...) leaves a copy that the
        // next list-major search rebuilds instead of one that silently misses relocated / extended blocks
        const bool live = patch && !shadow_dirty_;
        if (live) {
            moved_.assign((size_t)nlist, 0);
            shadow_dirty_ = true;
        } else {
            moved_.clear();
        }
        grow_lists_(new_len, &est);
        // ---- destination rows (insertion order kept inside every list), then encode / scatter
        launch_ivf_rank(a_lab_.as<int64_t>(), ni, nlist, chunk, a_hist_.as<uint32_t>(), d_list_start_.as<int64_t>(),
                        a_dest_.as<int64_t>(), R.stream);
        append_(ni, a_xpad_.as<float>(), a_lab_.as<int64_t>(), a_dest_.as<int64_t>());
        if (indices_options == 1) // INDICES_IVF: the label of an entry is (list << 32 | offset), user ids are not kept at all
            launch_ivf_pair_ids(a_lab_.as<int64_t>(), a_dest_.as<int64_t>(), ni, d_list_start_.as<int64_t>(), arena_ids_.as<int64_t>(),
                                R.stream);
        else
            launch_scatter_i64(a_ids_.as<int64_t>(), a_dest_.as<int64_t>(), ni, arena_ids_.as<int64_t>(), R.stream);
        HIP_CHECK(hipMemcpyAsync(d_list_len_.p, a_newlen_.p, (size_t)nlist * 4, hipMemcpyDeviceToDevice, R.stream));
        if (live) {
            h_first_row_.resize((size_t)nlist);
            for (int l = 0; l < nlist; l++)
                h_first_row_[l] = moved_[l] ? 0u : new_len[l] != list_len_[l] ? (list_len_[l] & ~31u) : 0xffffffffu;
            a_first_row_.ensure((size_t)nlist * 4);
            HIP_CHECK(hipMemcpyAsync(a_first_row_.p, h_first_row_.data(), (size_t)nlist * 4, hipMemcpyHostToDevice, R.stream));
            try {
                shadow_dirty_ = false; // (lmf_shadow_room_ keeps the contents of a clean copy when it grows the buffer)
                lmf_patch_(a_first_row_.as<uint32_t>());
            } catch (const DeviceOutOfMemory&) {
                // no room to grow the copy: drop it (the next list-major search rebuilds it, or falls back to query-major)
                (void)lmf_release_();
                shadow_dirty_ = true;
            } catch (...) {
                shadow_dirty_ = true;
                throw;
            }
        }
        list_len_ = new_len;
        nstored_ += added;
        ntotal += ni;
        R.sync(); // the staging buffers are reused by the next page
    }
    // holes + slack beyond 60 % of the stored rows: rebuild without holes (amortised: the stored rows grew by a
    // constant factor since the last rebuild)
    if (arena_rows_ > ((int64_t)1 << 16) && (double)arena_rows_ > 1.6 * (double)nstored_ + 2.0 * granule_ * nlist) compact_();
}

void GpuIndexIVF::set_lists(const uint32_t* list_sizes, const uint8_t* codes, const idx_t* ids) {
    FA_THROW_IF_NOT_MSG(is_trained, "copy the quantizers before the lists");
    FA_THROW_IF_NOT_MSG(list_sizes, "null list sizes");
    std::lock_guard<std::mutex> g(mu_);
    res_->set_device();
    const GpuResources& R = *res_;
    const int64_t G = granule_;
    std::vector<int64_t> src_start(nlist), nstart(nlist);
    std::vector<uint32_t> ncap(nlist);
    int64_t acc = 0, rows = 0;
    for (int l = 0; l < nlist; l++) {
        src_start[l] = acc;
        nstart[l] = rows;
        ncap[l] = (uint32_t)round_up(list_sizes[l], (size_t)G);
        acc += list_sizes[l];
        rows += ncap[l];
    }
    FA_THROW_IF_NOT_MSG(acc == 0 || (codes && (ids || indices_options == 1)), "null codes / ids");
    shadow_dirty_ = true;
    // nothing of the index changes before every device copy has been issued successfully
    arena_rows_ = 0; // (old contents are dropped: nothing to keep when the buffers grow)
    hole_rows_ = 0;
    ensure_arena_(rows);
    const size_t src_row = ref_row_bytes_();
    if (acc > 0) {
        // ids: list by list into the lists' row ranges
        DevBuf tmp_ids, tmp_codes, dsrc;
        tmp_ids.ensure((size_t)acc * 8);
        std::vector<idx_t> pair_ids;
        if (indices_options == 1) { // INDICES_IVF: (list << 32 | offset) instead of the caller's ids
            pair_ids.resize((size_t)acc);
            for (int l = 0; l < nlist; l++)
                for (uint32_t j = 0; j < list_sizes[l]; j++) pair_ids[(size_t)src_start[l] + j] = ((idx_t)l << 32) | (idx_t)j;
            ids = pair_ids.data();
        }
        HIP_CHECK(hipMemcpyAsync(tmp_ids.p, ids, (size_t)acc * 8, hipMemcpyDefault, R.stream));
        if (!pair_ids.empty()) R.sync(); // (the host vector goes out of scope below)
        std::vector<IvfMoveJob> jobs;
        for (int l = 0; l < nlist; l++)
            if (list_sizes[l]) jobs.push_back({src_start[l], nstart[l], (int64_t)list_sizes[l]});
        a_jobs_.ensure(jobs.size() * sizeof(IvfMoveJob));
        HIP_CHECK(hipMemcpyAsync(a_jobs_.p, jobs.data(), jobs.size() * sizeof(IvfMoveJob), hipMemcpyHostToDevice,
                                 R.stream));
        launch_ivf_move(tmp_ids.as<uint8_t>(), arena_ids_.as<uint8_t>(), a_jobs_.as<IvfMoveJob>(), (int)jobs.size(), 8,
                        R.stream);
        // `codes` are the reference's list payloads: d floats (IVFFlat) or M bytes (IVFPQ) per entry
        tmp_codes.ensure((size_t)acc * code_bytes_);
        if (fused_kind_() != 1) {
            HIP_CHECK(hipMemsetAsync(tmp_codes.p, 0, (size_t)acc * code_bytes_, R.stream));
            HIP_CHECK(hipMemcpy2DAsync(tmp_codes.p, code_bytes_, codes, src_row, src_row, (size_t)acc, hipMemcpyDefault,
                                       R.stream));
            if (fused_kind_() == 2) {
                // plain rows -> 64-row chunk-major blocks (kernels.h sq_code_offset)
                dsrc.ensure((size_t)nlist * 8 * 2 + (size_t)nlist * 4);
                int64_t* d_src = dsrc.as<int64_t>();
                int64_t* d_dst = d_src + nlist;
                uint32_t* d_len = (uint32_t*)(d_dst + nlist);
                HIP_CHECK(hipMemcpyAsync(d_src, src_start.data(), (size_t)nlist * 8, hipMemcpyHostToDevice, R.stream));
                HIP_CHECK(hipMemcpyAsync(d_dst, nstart.data(), (size_t)nlist * 8, hipMemcpyHostToDevice, R.stream));
                HIP_CHECK(hipMemcpyAsync(d_len, list_sizes, (size_t)nlist * 4, hipMemcpyHostToDevice, R.stream));
                launch_ivfsq_pack_lists(tmp_codes.as<uint8_t>(), d_src, d_dst, d_len, nlist, (int)code_bytes_, sq_chunk_bytes_(),
                                        arena_.as<uint8_t>(), R.stream);
            } else {
                launch_ivf_move(tmp_codes.as<uint8_t>(), arena_.as<uint8_t>(), a_jobs_.as<IvfMoveJob>(), (int)jobs.size(),
                                (int)code_bytes_, R.stream);
            }
        } else {
            HIP_CHECK(hipMemcpyAsync(tmp_codes.p, codes, (size_t)acc * code_bytes_, hipMemcpyDefault, R.stream));
            dsrc.ensure((size_t)nlist * 8 * 2 + (size_t)nlist * 4);
            int64_t* d_src = dsrc.as<int64_t>();
            int64_t* d_dst = d_src + nlist;
            uint32_t* d_len = (uint32_t*)(d_dst + nlist);
            HIP_CHECK(hipMemcpyAsync(d_src, src_start.data(), (size_t)nlist * 8, hipMemcpyHostToDevice, R.stream));
            HIP_CHECK(hipMemcpyAsync(d_dst, nstart.data(), (size_t)nlist * 8, hipMemcpyHostToDevice, R.stream));
            HIP_CHECK(hipMemcpyAsync(d_len, list_sizes, (size_t)nlist * 4, hipMemcpyHostToDevice, R.stream));
            launch_ivfpq_pack_lists(tmp_codes.as<uint8_t>(), d_src, d_dst, d_len, nlist, (int)code_bytes_, arena_.as<uint8_t>(),
                                    R.stream);
        }
        R.sync();
    }
    for (int l = 0; l < nlist; l++) list_len_[l] = list_sizes[l];
    list_cap_ = ncap;
    list_start_ = nstart;
    arena_rows_ = rows;
    ntotal = acc;
    nstored_ = acc;
    upload_list_tables_();
    lists_changed_();
}

std::vector<idx_t> GpuIndexIVF::getListIndices(idx_t list) const {
    FA_THROW_IF_NOT(list >= 0 && list < nlist);
    std::lock_guard<std::mutex> g(mu_);
    res_->set_device();
    std::vector<idx_t> out(list_len_[list]);
    if (!out.empty())
        HIP_CHECK(hipMemcpy(out.data(), arena_ids_.as<idx_t>() + list_start_[list], out.size() * 8,
                            hipMemcpyDeviceToHost));
    return out;
}
std::vector<uint8_t> GpuIndexIVF::getListVectorData(idx_t list) const {
    FA_THROW_IF_NOT(list >= 0 && list < nlist);
    std::lock_guard<std::mutex> g(mu_);
    res_->set_device();
    const size_t dst_row = ref_row_bytes_();
    std::vector<uint8_t> out((size_t)list_len_[list] * dst_row);
    if (out.empty()) return out;
    if (fused_kind_() == 2) {
        // chunk-major blocks -> plain rows -> the reference's code_size-byte entries
        DevBuf tmp;
        tmp.ensure((size_t)list_len_[list] * code_bytes_);
        launch_ivfsq_unpack_rows(arena_.as<uint8_t>(), list_start_[list], list_len_[list], (int)code_bytes_, sq_chunk_bytes_(),
                                 tmp.as<uint8_t>(), res_->stream);
        HIP_CHECK(hipMemcpy2DAsync(out.data(), dst_row, tmp.p, code_bytes_, dst_row, list_len_[list], hipMemcpyDeviceToHost,
                                   res_->stream));
        res_->sync();
    } else if (fused_kind_() != 1) {
        HIP_CHECK(hipMemcpy2D(out.data(), dst_row, arena_.as<uint8_t>() + list_start_[list] * code_bytes_, code_bytes_,
                              dst_row, list_len_[list], hipMemcpyDeviceToHost));
    } else {
        // rotated block layout -> the reference's plain [entry][M] codes
        DevBuf tmp;
        tmp.ensure(out.size());
        launch_ivfpq_unpack_list(arena_.as<uint8_t>(), list_start_[list], list_len_[list], (int)code_bytes_,
                                 tmp.as<uint8_t>(), res_->stream);
        HIP_CHECK(hipMemcpyAsync(out.data(), tmp.p, out.size(), hipMemcpyDeviceToHost, res_->stream));
        res_->sync();
    }
    return out;
}

void GpuIndexIVF::search(idx_t n, const float* x, idx_t k, float* distances, idx_t* labels,
                         const SearchParameters* params) const {
    // (faiss/gpu/GpuIndexIVF.cu:358-402: nprobe of SearchParametersIVF when that is what the caller passed)
    const SearchParametersIVF* ivf = dynamic_cast<const SearchParametersIVF*>(params);
    search_core_(n, x, k, distances, labels, nullptr, nullptr, ivf && ivf->nprobe > 0 ? ivf->nprobe : nprobe,
                 params ? params->sel : nullptr);
}
void GpuIndexIVF::search_preassigned(idx_t n, const float* x, idx_t k, const idx_t* assign, const float* centroid_dis,
                                     float* distances, idx_t* labels, const SearchParameters* params) const {
    FA_THROW_IF_NOT_MSG(n == 0 || (assign && centroid_dis), "search_preassigned: null assign / centroid_dis");
    const SearchParametersIVF* ivf = dynamic_cast<const SearchParametersIVF*>(params);
    search_core_(n, x, k, distances, labels, assign, centroid_dis, ivf && ivf->nprobe > 0 ? ivf->nprobe : nprobe,
                 params ? params->sel : nullptr);
}
void GpuIndexIVF::search_core_(idx_t n, const float* x, idx_t k, float* distances, idx_t* labels, const idx_t* assign,
                               const float* centroid_dis, int nprobe_now, const IDSelector* sel) const {
    FA_THROW_IF_NOT_MSG(is_trained, "index not trained");
    FA_THROW_IF_NOT_MSG(k >= 1 && k <= kMaxSelectionK, "k must be in [1, 2048]");
    FA_THROW_IF_NOT_MSG(nprobe_now >= 1 && nprobe_now <= kMaxSelectionK, "nprobe must be in [1, 2048]");
    if (n == 0) return;
    FA_THROW_IF_NOT_MSG(x && distances && labels, "null argument");
    std::lock_guard<std::mutex> g(mu_);
    res_->set_device();
    const GpuResources& R = *res_;
    // IDSelector: one bit per arena row from the stored ids (one pass over 8 bytes per row, once per call); the scan
    // kernels consult it for the rows that would otherwise become candidates
    struct SelScope {
        const uint32_t*& m;
        ~SelScope() { m = nullptr; }
    } sel_scope{cur_sel_mask_};
    if (sel && arena_rows_ > 0) {
        SelProgram prog{};
        sel->compile(prog, R.device, R.stream);
        sel_mask_.ensure(div_up((size_t)arena_rows_, 64) * 8);
        launch_selector_mask(arena_ids_.as<int64_t>(), arena_rows_, 0, prog, sel_mask_.as<uint64_t>(), nullptr, R.stream);
        cur_sel_mask_ = sel_mask_.as<uint32_t>();
    }
    // which scan serves this call: decided once, on the whole batch, so that the pages / tiles of one call agree
    FA_THROW_IF_NOT_MSG(scan_mode >= 0 && scan_mode <= 3,
                        "scan_mode must be 0 (auto), 1 (query-major), 2 (list-major) or 3 (list-major, f32 matrix pipe)");
    if (scan_mode >= 2) {
        FA_THROW_IF_NOT_MSG(lm_capable_() || (scan_mode == 2 && lmf_capable_()),
                            "list-major scan: index type / dimension not supported (IVFFlat behind the f16 filter: d <= 512; "
                            "IVFPQ, scalar quantizer, the f32 scan: d <= 128)");
        // (the filter sweeps test the selector's row bits themselves; the f32 scans do not)
        FA_THROW_IF_NOT_MSG(!sel || (scan_mode == 2 && lmf_capable_()),
                            "list-major scan: IDSelector searches take the query-major scan (or the filter path of IVFFlat / IVFPQ)");
        cur_lm_ = true;
    } else {
        cur_lm_ = scan_mode == 0 && list_major_rule(n, nprobe_now, k, sel != nullptr);
    }
    // IVFFlat / IVFPQ: the list-major scan runs behind the f16 filter (results = the query-major scan's, bit for bit)
    // unless the f32 scan of round 3 is asked for.  Stored values outside the fp16 range: the query-major scan serves
    // the call (same bits, no filter).
    cur_preassigned_ = assign != nullptr;
    cur_lmf_ = cur_lm_ && scan_mode != 3 && lmf_capable_();
    if (cur_lmf_ && scan_mode == 0 && !use_filter_shadow) cur_lmf_ = cur_lm_ = false; // (the caller opted out of the copies)
    if (cur_lmf_ && nstored_ > 0) {
        IvfLmParams probe{};
        try {
            if (!lmf_prepare_(probe)) cur_lmf_ = cur_lm_ = false;
        } catch (const DeviceOutOfMemory&) {
            // the sweeps' copy of the lists (IVFFlat: + 50 % of the rows, IVFPQ: the codes again) does not fit: the
            // query-major scan returns the same bits without it.  Only an explicit scan_mode 2 reports the failure.
            if (scan_mode == 2) throw;
            (void)const_cast<GpuIndexIVF*>(this)->lmf_release_();
            cur_lmf_ = cur_lm_ = false;
        }
    }
    last_scan_mode_ = cur_lm_ ? 2 : 1;
    last_scan_arith_ = cur_lm_ && !cur_lmf_ ? 1 : 0;
    if (use_paged_path(R, n, d, x, distances, labels)) {
        const idx_t page = paged_page_size(R, n, 65536);
        idx_t done = 0; // pages come in order
        paged_host_search(R, n, x, d, k, distances, labels, page, [&](idx_t ni, const float* dq, float* dD, idx_t* dI) {
            search_core_body_(ni, dq, k, dD, dI, assign ? assign + (size_t)done * nprobe_now : nullptr,
                              centroid_dis ? centroid_dis + (size_t)done * nprobe_now : nullptr, nprobe_now);
            done += ni;
        });
        return;
    }
    search_core_body_(n, x, k, distances, labels, assign, centroid_dis, nprobe_now);
}
void GpuIndexIVF::search_core_body_(idx_t n, const float* x, idx_t k, float* distances, idx_t* labels, const idx_t* assign,
                                    const float* centroid_dis, int nprobe_now) const {
    const GpuResources& R = *res_;
    // preassigned arrays are [n][nprobe] whatever nlist is (surplus columns hold -1)
    const int np = assign ? nprobe_now : std::min(nprobe_now, nlist);
    const bool out_dev_d = is_device_pointer(distances), out_dev_i = is_device_pointer(labels);
    // query tile bounded by the worst-case candidate volume (reference: IVFUtils.cu:46-127)
    uint32_t max_len = 1;
    for (auto l : list_len_) max_len = std::max(max_len, l);
    FA_THROW_IF_NOT_MSG((uint64_t)np * max_len < ((uint64_t)1 << 32), "nprobe * longest list exceeds 2^32 scan positions");
    size_t per_q = (size_t)np * max_len * 8;
    idx_t tile = (idx_t)std::max<size_t>(1, R.temp_budget_bytes / per_q);
    tile = std::min<idx_t>(tile, 16384);
    int fused_cap = 0, fused_kp = 0, fused_nlut = 1;
    bool fused;
    int sq_npc = np; // scalar quantizer: probes per workgroup the LDS table rows allow
    if (fused_kind_() == 2) {
        // the scan exists only as the fused kernel; with residual encoding (L2) every probe of a workgroup needs its
        // own table row, so a large nprobe x d is split over several workgroups per query
        const int dsq = (int)round_up(d, 16);
        IvfFusedParams probe{};
        fill_fused_(probe);
        fused = false;
        for (;;) {
            const int rows = sq_table_rows(metric_type, probe.sq_by_residual != 0, sq_npc);
            if (ivf_fused_supported(2, rows, dsq, (int)k, np, &fused_cap, &fused_kp, &fused_nlut) &&
                (size_t)rows * dsq * 4 <= 64 * 1024) {
                fused = true;
                break;
            }
            if (sq_npc == 1) break;
            sq_npc = (sq_npc + 1) / 2;
        }
        FA_THROW_IF_NOT_MSG(fused, "scalar-quantizer search: k / d outside the range of the scan kernel");
    } else {
        fused = use_fused_scan && ivf_fused_supported(fused_kind_(), fused_M_(), dpad_, (int)k, np, &fused_cap, &fused_kp,
                                                     &fused_nlut);
    }
    if (fused || cur_lm_) tile = 65536; // no per-candidate scratch here: the tile only bounds the staging buffers
    for (idx_t i0 = 0; i0 < n; i0 += tile) {
        check_interrupt();
        const int ni = (int)std::min(tile, n - i0);
        float* dD = out_dev_d ? distances + (size_t)i0 * k : nullptr;
        idx_t* dI = out_dev_i ? labels + (size_t)i0 * k : nullptr;
        if (!dD) {
            out_d_.ensure((size_t)ni * k * 4);
            dD = out_d_.as<float>();
        }
        if (!dI) {
            out_i_.ensure((size_t)ni * k * 8);
            dI = out_i_.as<idx_t>();
        }
        if (nstored_ == 0) {
            // trained but empty (e.g. an IndexShards shard that received no rows): nothing to scan
            launch_fill_knn(dD, dI, (int64_t)ni * k, metric_type, R.stream);
            if (!out_dev_d) copy_out(R, distances + (size_t)i0 * k, dD, (size_t)ni * k * 4);
            if (!out_dev_i) copy_out(R, labels + (size_t)i0 * k, dI, (size_t)ni * k * 8);
            R.sync();
            continue;
        }
        q_pad_.ensure((size_t)ni * dpad_ * 4);
        stage_padded(R, x + (size_t)i0 * d, ni, d, dpad_, q_raw_, q_pad_.as<float>());
        // ---- coarse quantizer: nprobe nearest centroids (reference: IVFBase.cu:509-593)
        c_dis_.ensure((size_t)ni * np * 4);
        c_ids_.ensure((size_t)ni * np * 8);
        cur_coarse_bad_ = nullptr;
        if (assign) {
            HIP_CHECK(hipMemcpyAsync(c_ids_.p, assign + (size_t)i0 * np, (size_t)ni * np * 8, hipMemcpyDefault, R.stream));
            HIP_CHECK(hipMemcpyAsync(c_dis_.p, centroid_dis + (size_t)i0 * np, (size_t)ni * np * 4, hipMemcpyDefault,
                                     R.stream));
            launch_ivf_sanitize_assign(c_ids_.as<idx_t>(), (int64_t)ni * np, nlist, R.stream);
        } else {
            // (behind the filter sweeps the quantizer's own overflow -- a query outside the fp16 range, a flood of near-ties -- joins
            // the redo set of the list-major search instead of costing every search a read-back right here)
            if (cur_lm_ && cur_lmf_) c_bad_.ensure((size_t)ni * 4);
            cur_coarse_bad_ = cur_lm_ && cur_lmf_ ? c_bad_.as<uint32_t>() : nullptr;
            quantizer->search_device(ni, q_pad_.as<float>(), np, c_dis_.as<float>(), c_ids_.as<idx_t>(),
                                     const_cast<uint32_t*>(cur_coarse_bad_));
        }
        // the query-major scan of the `nn` queries staged in q_pad_ / c_ids_ / c_dis_ -> gD / gI (device)
        auto query_major = [&](int nn, float* gD, idx_t* gI) {
        const int ni = nn; // (shadows the tile's count: the redo of a list-major search runs this on a few queries)
        float* const dD = gD;
        idx_t* const dI = gI;
        if (fused) {
            // ---- table build + list scan + k-selection in one launch, nothing but results leaves LDS
            IvfFusedParams fp{};
            fp.metric = metric_type;
            fp.kind = fused_kind_();
            fp.nq = ni;
            fp.nprobe = np;
            fp.d = d;
            fp.dpad = dpad_;
            fp.xq = q_pad_.as<float>();
            fp.ldq = dpad_;
            fp.coarse_ids = c_ids_.as<idx_t>();
            fp.coarse_dis = c_dis_.as<float>();
            fp.list_len = d_list_len_.as<uint32_t>();
            fp.list_start = d_list_start_.as<int64_t>();
            fp.arena_ids = arena_ids_.as<int64_t>();
            fp.sel_mask = cur_sel_mask_;
            fp.k = (int)k;
            fp.kp = fused_kp;
            fp.cap = fused_cap;
            fp.nlut = fused_nlut;
            // enough workgroups to fill the chip twice over; probes are split only for small batches
            const int want = 4 * R.num_cus;
            int G = ni >= want ? 1 : std::min<int>(np, (int)div_up(want, ni));
            fp.npc = (int)div_up(np, G);
            if (fp.kind == 2) fp.npc = std::min(fp.npc, sq_npc);
            fp.G = (int)div_up(np, fp.npc);
            fp.out_dis = dD;
            fp.out_ids = dI;
            // large batches: the final selection of a query runs in its own launch (ivf_finish_kernel)
            static const char* defer_env = experiment_env("FAISS_AMD_IVF_DEFER"); // timing experiments only: 0 / 1
            fp.defer_finish = fp.G == 1 && ni >= want && (!defer_env || atoi(defer_env) != 0) ? 1 : 0;
            if (defer_env && atoi(defer_env) == 1 && fp.G == 1) fp.defer_finish = 1;
            if (fp.G > 1 || fp.defer_finish) {
                const size_t per_q = fp.defer_finish ? (size_t)fp.cap : (size_t)fp.G * k;
                part_keys_.ensure((size_t)ni * per_q * 8);
                part_cnt_.ensure((size_t)ni * fp.G * 4);
                prefix_.ensure((size_t)ni * (np + 1) * 4);
                fp.part_keys = part_keys_.as<unsigned long long>();
                fp.part_cnt = part_cnt_.as<uint32_t>();
                fp.prefix_out = prefix_.as<uint32_t>();
            }
            fill_fused_(fp);
            if (fp.kind == 2) fp.M = sq_table_rows(metric_type, fp.sq_by_residual != 0, fp.npc);
            probe_len_.ensure((size_t)ni * np * 4);
            probe_start_.ensure((size_t)ni * np * 8);
            launch_ivf_probe_info(fp.coarse_ids, (int64_t)ni * np, d_list_len_.as<uint32_t>(),
                                  d_list_start_.as<int64_t>(), probe_len_.as<uint32_t>(), probe_start_.as<int64_t>(), R.stream);
            fp.probe_len = probe_len_.as<uint32_t>();
            fp.probe_start = probe_start_.as<int64_t>();
            static const bool phases = experiment_env("FAISS_AMD_IVF_PHASES") != nullptr; // diagnostics only
            DevBuf ticks;
            if (phases && fp.kind == 1) {
                ticks.ensure(64);
                HIP_CHECK(hipMemsetAsync(ticks.p, 0, 64, R.stream));
                fp.phase_ticks = ticks.as<unsigned long long>();
            }
            {
                SpanGuard sg(&R, fp.kind == 1 ? "ivfpq_fused_kernel" : fp.kind == 2 ? "ivfsq_fused_kernel" : "ivfflat_fused_kernel");
                launch_ivf_fused(fp, R.stream);
            }
            if (fp.phase_ticks) {
                unsigned long long h[5];
                HIP_CHECK(hipMemcpyAsync(h, ticks.p, sizeof(h), hipMemcpyDeviceToHost, R.stream));
                R.sync();
                const double w = (double)std::max<unsigned long long>(h[4], 1);
                fprintf(stderr, "ivfpq_fused_kernel phases, mean shader-clock ticks per workgroup (%llu workgroups): probes+query "
                        "%.0f, table %.0f, scan %.0f, finish %.0f\n", h[4], h[0] / w, h[1] / w, h[2] / w, h[3] / w);
            }
            if (fp.defer_finish) {
                SpanGuard sg(&R, "ivf_finish_kernel");
                launch_ivf_finish(fp, R.stream);
            } else if (fp.G > 1) {
                SelectParams sp{};
                sp.metric = metric_type;
                sp.nq = ni;
                sp.k = (int)k;
                sp.keys = fp.part_keys;
                sp.q_stride = (int64_t)fp.G * k;
                sp.nseg = fp.G;
                sp.seg_stride = k;
                sp.seg_cnt = fp.part_cnt;
                sp.mode = 1;
                sp.nprobe = np;
                sp.ivf_prefix = fp.prefix_out;
                sp.coarse_ids = c_ids_.as<idx_t>();
                sp.list_start = d_list_start_.as<int64_t>();
                sp.arena_ids = arena_ids_.as<int64_t>();
                sp.out_dis = dD;
                sp.out_ids = dI;
                SpanGuard sg(&R, "select_k_kernel");
                launch_select_k(sp, R.stream);
            }
            return;
        }
        // ---- per-(query, probe) offsets
        prefix_.ensure((size_t)ni * (np + 1) * 4);
        totals_.ensure((size_t)ni * 4);
        launch_ivf_prefix(c_ids_.as<idx_t>(), ni, np, d_list_len_.as<uint32_t>(), prefix_.as<uint32_t>(),
                          totals_.as<uint32_t>(), R.stream);
        std::vector<uint32_t> tot(ni);
        HIP_CHECK(hipMemcpyAsync(tot.data(), totals_.p, (size_t)ni * 4, hipMemcpyDeviceToHost, R.stream));
        R.sync();
        std::vector<int64_t> qoff(ni);
        int64_t nkeys = 0;
        for (int q = 0; q < ni; q++) {
            qoff[q] = nkeys;
            nkeys += tot[q];
        }
        q_off_.ensure((size_t)ni * 8);
        HIP_CHECK(hipMemcpyAsync(q_off_.p, qoff.data(), (size_t)ni * 8, hipMemcpyHostToDevice, R.stream));
        keys_.ensure(std::max<size_t>((size_t)nkeys * 8, 256));
        // ---- list scan: every candidate distance as a 64-bit key
        nprobe_eff_ = np;
        scan_(ni, q_pad_.as<float>(), (int)k, nullptr);
        // ---- k-selection + (probe, offset) -> user id
        SelectParams sp{};
        sp.metric = metric_type;
        sp.nq = ni;
        sp.k = (int)k;
        sp.keys = keys_.as<unsigned long long>();
        sp.q_off = q_off_.as<int64_t>();
        sp.q_stride = 0;
        sp.nseg = 1;
        sp.seg_stride = 0;
        sp.seg_cnt = totals_.as<uint32_t>();
        sp.mode = 1;
        sp.nprobe = np;
        sp.ivf_prefix = prefix_.as<uint32_t>();
        sp.coarse_ids = c_ids_.as<idx_t>();
        sp.list_start = d_list_start_.as<int64_t>();
        sp.arena_ids = arena_ids_.as<int64_t>();
        sp.out_dis = dD;
        sp.out_ids = dI;
        {
            SpanGuard sg(&R, "select_k_kernel");
            launch_select_k(sp, R.stream);
        }
        }; // query_major
        if (cur_lm_) {
            // ---- large batch: list-major (ivf_listmajor.hip, ivf_lm_filter.hip)
            std::vector<uint32_t> redo;
            search_listmajor_(ni, q_pad_.as<float>(), c_ids_.as<idx_t>(), c_dis_.as<float>(), np, (int)k, dD, dI, 0, &redo);
            if (!redo.empty()) {
                // filter path: queries whose candidate segment overflowed (a threshold that admits too much: fewer
                // granules than k, near-duplicate data) or that leave the fp16 range go through the query-major scan --
                // the same bits.  They are gathered to the front of the staging buffers (the list-major pass is done
                // with them) and scattered back into the tile's results.
                const int nr = (int)redo.size();
                DevBuf olist, gq, gids, gdis, gD, gI;
                olist.ensure((size_t)nr * 4);
                HIP_CHECK(hipMemcpyAsync(olist.p, redo.data(), (size_t)nr * 4, hipMemcpyHostToDevice, R.stream));
                gq.ensure((size_t)nr * dpad_ * 4);
                gids.ensure((size_t)nr * np * 8);
                gdis.ensure((size_t)nr * np * 4);
                gD.ensure((size_t)nr * k * 4);
                gI.ensure((size_t)nr * k * 8);
                launch_gather_rows(q_pad_.as<float>(), dpad_, dpad_, olist.as<uint32_t>(), nr, gq.as<float>(), R.stream);
                if (cur_coarse_bad_) {
                    // (some of them may be queries the coarse quantizer handed back: its general path serves the whole redo set)
                    quantizer->search_device(nr, gq.as<float>(), np, gdis.as<float>(), gids.as<idx_t>());
                } else {
                    launch_gather_rows((const float*)c_ids_.p, 2 * np, 2 * np, olist.as<uint32_t>(), nr, (float*)gids.p, R.stream);
                    launch_gather_rows(c_dis_.as<float>(), np, np, olist.as<uint32_t>(), nr, gdis.as<float>(), R.stream);
                }
                HIP_CHECK(hipMemcpyAsync(q_pad_.p, gq.p, (size_t)nr * dpad_ * 4, hipMemcpyDeviceToDevice, R.stream));
                HIP_CHECK(hipMemcpyAsync(c_ids_.p, gids.p, (size_t)nr * np * 8, hipMemcpyDeviceToDevice, R.stream));
                HIP_CHECK(hipMemcpyAsync(c_dis_.p, gdis.p, (size_t)nr * np * 4, hipMemcpyDeviceToDevice, R.stream));
                // the fused scan keeps no per-candidate scratch; the key-segment scan holds nprobe x longest list keys per
                // query: the redo set then goes through it in sub-tiles of the scratch budget (a forced scan_mode 2 on tie-heavy
                // data can redo every query of the tile)
                const int sub = fused ? nr : (int)std::max<size_t>(1, std::min<size_t>((size_t)nr, R.temp_budget_bytes / per_q));
                for (int s0 = 0; s0 < nr; s0 += sub) {
                    const int ns = std::min(sub, nr - s0);
                    if (s0 > 0) {
                        HIP_CHECK(hipMemcpyAsync(q_pad_.p, gq.as<float>() + (size_t)s0 * dpad_, (size_t)ns * dpad_ * 4,
                                                 hipMemcpyDeviceToDevice, R.stream));
                        HIP_CHECK(hipMemcpyAsync(c_ids_.p, gids.as<idx_t>() + (size_t)s0 * np, (size_t)ns * np * 8,
                                                 hipMemcpyDeviceToDevice, R.stream));
                        HIP_CHECK(hipMemcpyAsync(c_dis_.p, gdis.as<float>() + (size_t)s0 * np, (size_t)ns * np * 4,
                                                 hipMemcpyDeviceToDevice, R.stream));
                    }
                    query_major(ns, gD.as<float>() + (size_t)s0 * k, gI.as<idx_t>() + (size_t)s0 * k);
                }
                launch_scatter_results(gD.as<float>(), gI.as<idx_t>(), (int)k, olist.as<uint32_t>(), nr, dD, dI, R.stream);
                R.sync(); // `redo` and the gathered buffers die with this scope
            }
        } else {
            query_major(ni, dD, dI);
        }
        if (!out_dev_d) copy_out(R, distances + (size_t)i0 * k, dD, (size_t)ni * k * 4);
        if (!out_dev_i) copy_out(R, labels + (size_t)i0 * k, dI, (size_t)ni * k * 8);
        R.sync();
    }
}

// ---------------------------------------------------------------------- list-major search (ivf_listmajor.hip)
bool GpuIndexIVF::list_major_rule(idx_t n, int nprobe_now, idx_t k, bool has_selector) const {
    if (!(lm_capable_() || lmf_capable_()) || (has_selector && !lmf_capable_()) || k > kMaxSelectionK || nstored_ == 0) return false;
    const int64_t np = std::min<int64_t>(nprobe_now, nlist);
    const double avg_len = (double)nstored_ / (double)nlist;
    if (fused_kind_() != 0 && !lmf_capable_()) {
        // scalar quantizer / IVFPQ shapes the filter does not serve: round 3's f32 list-major scan and its rule (measured at
        // nlist 4096 / nprobe 32 only: IVFPQ break-even near (rows per list) x (queries per list) = 415 x 78): every list
        // has to meet >= 8 of the batch's queries on average, below that the 32-query MFMA blocks run mostly empty
        if ((int64_t)n * np < (int64_t)8 * nlist) return false;
        if (fused_kind_() == 1 &&
            !(lm_pq_lds_capable_() && (double)n * (double)np * (double)nstored_ >= 50000.0 * (double)nlist * (double)nlist))
            return false;
        return n >= 2048 && (double)nstored_ >= 64.0 * (double)nlist;
    }
    // ---- behind the f16 filter (round 4): both scans return the same bits, so the choice is a matter of time only, and it is
    // made by two small cost models fitted to side-by-side timings (profiles/r04_i_scan_rule_sweep.txt, r04_k_ivfpq_rule_
    // sweep.txt: nlist 1024 / 4096 / 16384 x nprobe 8 / 32 / 128 x 512 ... 10 000 queries at nb = 1M; profiles/r04_y_latency_
    // both_scans.txt: 16 ... 4096 queries at nb = 1M / 10M / 100M).  Against those 118 measurements the choice is off by more
    // than 10 % twice (11 % and 15 %); the thresholds it replaces (queries per list >= 8 and a fixed stream size) kept the
    // query-major scan for 128 ... 512 queries at nb >= 10M, where it is 1.4 ... 3.6 x slower.
    //   query-major: bound by the bytes it streams -- queries x probes x rows per list x bytes per row at 4.9 TB/s (IVFFlat) /
    //                4.3 TB/s (IVFPQ) on top of 0.12 ms;
    //   list-major:  0.3 ms of fixed launches + rerank / selection per query + two sweeps over the lists the batch touches
    //                (IVFFlat: the fp16 shadow at the fabric's rate; IVFPQ: per row and occupied 32-query block of an item).
    // the bound is the k-th best of the query's granule minima (16 rows each): a query that probes fewer than ~1.1 k granules
    // gets no bound and is redone query-major (bench sweep, nprobe 4 at nb = 1M: 61 granules for k = 100 -- 3.5 ms, all redone)
    if ((double)np * avg_len < 18.0 * (double)k) return false;
    const double pairs = (double)n * (double)np;
    const double touched = 1.0 - std::exp(-pairs / (double)nlist); // share of the lists the batch probes (uniform model)
    const double stream = pairs * avg_len * (double)ref_row_bytes_();
    const double kq = 0.5 + 0.5 * std::min<double>((double)k, 1000.0) / 100.0; // candidates per query grow with k
    double est_qm, est_lm; // ms
    if (fused_kind_() == 1 && lmf_sweep_kind_() == 2) {
        // IVFPQ through its decoded residuals (round 5; tools/pq_dim_sweep.py, profiles/r5_pq_dim_sweep.txt).  Query-major: the
        // fused scan is bound by its table lookups -- 2.6e12 per second with the 16-byte code chunks of M = 32 / 64, 0.77e12
        // otherwise (M = 48: 4.95 ms, M = 128: 12.8 ms for 160 k pairs x 488 rows).  List-major: the IVFFlat sweeps over 2 dh + 4
        // bytes per row + a rerank whose per-query table grows with M.
        const double lookups = pairs * avg_len * (double)fused_M_();
        est_qm = 0.13 + lookups / ((fused_M_() == 32 || fused_M_() == 64) ? 2.6e9 : 0.77e9);
        est_lm = 0.30 + 0.17e-3 * kq * (double)n + 2.0 * touched * (double)nstored_ * (2.0 * ivf_lmf_row_halfs(d) + 4.0) / 4.0e9;
    } else if (lmf_sweep_kind_() != 1) { // IVFFlat and the scalar quantizer: the same sweeps over an fp16 copy of the rows / codes
        est_qm = 0.12 + stream / 4.9e9;
        est_lm = 0.27 + 0.10e-3 * kq * (double)n + 2.0 * touched * (double)nstored_ * (2.0 * ivf_lmf_row_halfs(d) + 4.0) / 4.0e9;
    } else {
        if (avg_len < 128.0) return false; // (an item's set-up is never amortised: nlist 16384 at nb = 1M, 61 rows per list)
        est_qm = 0.13 + stream / 4.3e9;
        // ms per row and sweep pair at three occupied query blocks: 5.6e-8 with sweep 1 on every 2nd block (long lists),
        // 1.07e-7 otherwise, more on short lists (per-item latencies), scaled by the blocks an item really holds
        const double c = (avg_len >= 8192.0 ? 5.6e-8 : 1.07e-7) * (1.0 + 250.0 / avg_len) * ((double)d / 128.0);
        const double qpl = pairs / (touched * (double)nlist);
        const double nblk = std::min(3.0, std::max(1.0, std::ceil(qpl / 32.0)));
        est_lm = 0.30 + 0.055e-3 * kq * (double)n + touched * (double)nstored_ * c * nblk / 3.0;
    }
    return est_lm < est_qm;
}

// Queries [0, ni) with their coarse results on the device -> k best per query in dD / dI (device).  Splits the batch
// so that the key segments fit the scratch budget.
void GpuIndexIVF::search_listmajor_(int ni, const float* xq_pad, const idx_t* c_ids, const float* c_dis, int np, int k,
                                    float* dD, idx_t* dI, int level, std::vector<uint32_t>* redo) const {
    const GpuResources& R = *res_;
    uint32_t max_len = 1;
    for (auto l : list_len_) max_len = std::max(max_len, l);
    if (cur_lmf_ && level == 0) {
        // ---- behind the f16 filter (ivf_lm_filter.hip).  Rows of a list per work item: a quarter of an average list,
        // between 1024 and 8192, a multiple of the granule; candidate room per query: the k-th best granule estimate
        // admits ~ S ln(S / (S - k)) rows (S granule slots per query) + those inside the error band.
        FA_THROW_IF_NOT(redo != nullptr);
        const int64_t avg_len = std::max<int64_t>(1, nstored_ / std::max(nlist, 1));
        int RT = (int)std::min<int64_t>(8192, std::max<int64_t>(kLmRowsPerItem, (int64_t)round_up((size_t)(avg_len / 4), 256)));
        int64_t stride = std::max<int64_t>(1024, 6 * (int64_t)k);
        // granule: 32 G rows of a list hold two slots (16 G rows per lane half).  ~ 2048 slots per query at most: every
        // slot costs a scattered 4-byte store in sweep 1 and a read in the bound kernel, and S >> k slots already bound
        // the k-th best estimate tightly (S ln(S / (S - k)) candidates).  Measured (profiles/r04_c_filter_tuning_sweep.txt):
        // IVFFlat nb = 10M G = 1 / 2 / 4 / 8: 2.83 / 2.31 / 2.16 / 2.16 ms; nb = 1M G = 1 / 2 / 4: 1.00 / 1.04 / 1.19 ms
        if (lmf_rows_per_item > 0) RT = (int)round_up((size_t)lmf_rows_per_item, 256);
        // Sweep 1 on a SAMPLE of the rows (round 5): the first quarter of every work item's row chunk.  Any subset of the rows
        // bounds the k-th best estimate from above; the looser bound admits ~ 1 / share as many candidates into sweep 2's
        // segments, and the tightening launch behind it cuts them back to the rows inside the band of the k-th best
        // collected estimate, so the rerank sees no more than without sampling.  Lists of >= 1024 rows on average (shorter: an
        // item's set-up is not amortised over a quarter of it), and the sample must still hold >> k granules per query.
        // lmf_sample_shift: 0 = this rule, 1 .. 4 = prefix of RT >> shift rows, -1 = no sampling.
        int sample_rows = 0;
        {
            // (round 6, behind the cheaper candidate path of sweep 2: an EIGHTH of the chunk where the lists are long -- nb = 100M,
            // 24 000-row lists: sweep 1 0.82 -> 0.50 ms, sweep 2 3.22 -> 3.51, search 4.76 -> 4.64; a sixteenth loses again:
            // profiles/r6_ab_sweep1_sampling.txt)
            int shift = lmf_sample_shift > 0 ? lmf_sample_shift : (lmf_sample_shift == 0 ? (avg_len >= 16384 ? 3 : avg_len >= 1024 ? 2 : 0) : 0);
            while (shift > 0 && (double)np * (double)avg_len / (double)(1 << shift) < 64.0 * (double)k) --shift;
            if (shift > 0) sample_rows = (int)round_up((size_t)(RT >> shift), 256);
            if (sample_rows >= RT) sample_rows = 0;
        }
        const double share = sample_rows ? (double)sample_rows / (double)std::min<int64_t>(RT, std::max<int64_t>(avg_len, 1)) : 1.0;
        const double rows_per_query = (double)np * (double)avg_len * std::min(1.0, share);
        int G = 1;
        while (G < 8 && rows_per_query / (16.0 * G) > 2048.0) G *= 2;
        if (lmf_gran_blocks > 0) G = lmf_gran_blocks;
        if (G > 8 && G <= 32) RT = (int)round_up((size_t)RT, (size_t)32 * G); // (an item holds whole granules)
        if (sample_rows) sample_rows = (int)round_up((size_t)sample_rows, (size_t)32 * G);
        if (sample_rows >= RT) sample_rows = 0;
        // candidate room: ~ k / share rows at or below the sampled bound + those inside the band
        if (sample_rows) stride = std::max<int64_t>(stride, (int64_t)std::min(16384.0, 2048.0 + 3.0 * (double)k / std::min(1.0, share)));
        if (lmf_cand_cap > 0) stride = std::max<int64_t>(lmf_cand_cap, k);
        // (round 4's sampling of every 2nd 32-row block for long lists remains behind the tuning call: min_stride)
        const int min_stride = lmf_min_stride > 0 ? std::min(lmf_min_stride, 8) : 1;
        FA_THROW_IF_NOT_MSG(G >= 1 && G <= 32 && (G & (G - 1)) == 0 && RT <= 65280, "filter tuning: granule / rows per item");
        // granule slots a query can own: those of the np longest lists
        int64_t gstride = 0;
        {
            std::vector<uint32_t> len(list_len_);
            const size_t top = (size_t)std::min<int64_t>(np, nlist);
            std::nth_element(len.begin(), len.begin() + (top - 1), len.end(), std::greater<uint32_t>());
            for (size_t i = 0; i < top; i++)
                gstride += 2 * (int64_t)ivf_lmf_list_granules(len[i], (uint32_t)RT, (uint32_t)(32 * G), (uint32_t)sample_rows);
            // (a caller's own assignment may name a list more than once: the longest list np times)
            if (cur_preassigned_)
                gstride = 2 * (int64_t)np * (int64_t)ivf_lmf_list_granules(max_len, (uint32_t)RT, (uint32_t)(32 * G), (uint32_t)sample_rows);
            gstride = std::max<int64_t>(gstride, 2);
        }
        const size_t per_q = (size_t)stride * 10 + (size_t)gstride * 4 + (size_t)(np + 1) * 12 + 256 +
                             (fused_kind_() != 0 ? (size_t)np * ((size_t)ivf_lmf_row_halfs(d) * 2 + 4) : 0); // (IVFPQ / SQ: fp16 operands per probe)
        // (kind 1's sq-style prepare also writes lm_an_: sized in the chunk function)
        const int64_t fit = std::max<int64_t>(1, std::min<int64_t>((int64_t)(R.temp_budget_bytes / per_q), (1 << 20)));
        for (int c0 = 0; c0 < ni; c0 += (int)std::min<int64_t>(fit, ni)) {
            const int cn = (int)std::min<int64_t>(fit, ni - c0);
            search_listmajor_filter_chunk_(cn, c0, xq_pad + (size_t)c0 * dpad_, c_ids + (size_t)c0 * np, c_dis + (size_t)c0 * np, np,
                                           k, dD + (size_t)c0 * k, dI + (size_t)c0 * k, stride, RT | (G << 16), gstride,
                                           min_stride | (sample_rows << 4), *redo);
        }
        return;
    }
    // level 0: the regular search; level 1: queries whose candidate segment overflowed, with 16 x the room; level 2: every
    // probe and row in pass 1 (no bound, exact capacity) -- the last resort, its segments hold ALL probed rows
    const bool force_all = level >= 2;
    // Rows of a list per work item = the rows of a list pass 1 looks at (its first chunk): a quarter of an average
    // list, between 1024 and 8192.  The bound of pass 2 is the k-th best of a SAMPLE (the first chunk of the min_p1 nearest
    // lists); what it admits grows like k / (sampled fraction): with 1024-row chunks of 24 000-row lists (nb = 100M) 4 %
    // were sampled, ~6000 candidates per query overflowed every segment and the redo cost 7 x the search.
    const int64_t avg_len = std::max<int64_t>(1, nstored_ / std::max(nlist, 1));
    const int RT = (int)std::min<int64_t>(8192, std::max<int64_t>(kLmRowsPerItem, (int64_t)round_up((size_t)(avg_len / 4), 128)));
    // Probes of pass 1: enough to see k rows, and at least min_p1 -- chosen so that a list meets ~14 of them (half a
    // 32-query MFMA block: pass 1 then costs one sweep of the lists' first row chunks whatever min_p1 is), at most 8, at
    // most half of the probes, at most 4 when the lists are longer than a chunk (the bound kernel selects among min_p1
    // chunks of keys: 0.5 ms for 7 x 1024 at nb = 10M against 0.29 for 4 x 1024, more than the tighter bound saves).
    static const char* p1_env = experiment_env("FAISS_AMD_LM_P1"); // tuning experiments
    const int64_t chunk_rows = std::min<int64_t>(max_len, RT);
    int min_p1 = (int)std::min<int64_t>(8, std::max<int64_t>(1, (14 * (int64_t)nlist + ni / 2) / std::max(ni, 1)));
    if (avg_len >= kLmRowsPerItem) min_p1 = std::min(min_p1, 4);
    min_p1 = std::max(1, std::min(min_p1, np / 2));
    if (p1_env) min_p1 = std::max(1, std::min(np, atoi(p1_env)));
    // segment of a query: the rows of pass 1 (first row chunk of its lists: fewer than k + one chunk, or min_p1 chunks) +
    // room for the candidates of pass 2
    const int64_t cap2 = std::max<int64_t>(avg_len >= kLmRowsPerItem ? 4096 : 2048, 4 * (int64_t)k) * (level == 1 ? 16 : 1);
    const int64_t c1max = std::max<int64_t>((int64_t)k + chunk_rows, (int64_t)min_p1 * chunk_rows); // rows of pass 1
    const int64_t stride = force_all ? std::max<int64_t>((int64_t)np * max_len, k) : c1max + cap2;
    // (level 2: one query's segment holds every row it probes -- it has to fit the scratch budget by itself)
    FA_THROW_IF_NOT_MSG(!force_all || (size_t)stride * 8 <= std::max<size_t>(R.temp_budget_bytes, (size_t)1 << 30),
                        "list-major scan: the rows one query probes exceed the scratch budget (setTempMemory); use scan_mode 1");
    const int64_t fit = std::max<int64_t>(1, (int64_t)(R.temp_budget_bytes / ((size_t)stride * 8)));
    for (int c0 = 0; c0 < ni; c0 += (int)std::min<int64_t>(fit, ni)) {
        const int cn = (int)std::min<int64_t>(fit, ni - c0);
        search_listmajor_chunk_(cn, xq_pad + (size_t)c0 * dpad_, c_ids + (size_t)c0 * np, c_dis + (size_t)c0 * np, np, k,
                                dD + (size_t)c0 * k, dI + (size_t)c0 * k, level, stride, min_p1, RT, c1max);
    }
}

void GpuIndexIVF::search_listmajor_chunk_(int ni, const float* xq_pad, const idx_t* c_ids, const float* c_dis, int np, int k,
                                          float* dD, idx_t* dI, int level, int64_t stride, int min_p1, int RT, int64_t c1max) const {
    const GpuResources& R = *res_;
    const bool force_all = level >= 2;
    // upper bound of the work items: sum over (pass, list) of ceil(pairs / 128) * ceil(len / RT)
    int64_t sum_nrt = 0, nrt_max = 1;
    for (auto l : list_len_) {
        const int64_t nrt = (int64_t)div_up(l, (size_t)RT);
        sum_nrt += nrt;
        nrt_max = std::max(nrt_max, nrt);
    }
    const int64_t npairs = (int64_t)ni * np;
    const int qpi = ivf_lm_queries_per_item(fused_kind_());
    const int64_t max_items = nrt_max * (int64_t)div_up((size_t)npairs, (size_t)qpi) + 2 * sum_nrt + 16;
    FA_THROW_IF_NOT_MSG(max_items < ((int64_t)1 << 30), "list-major scan: too many work items");

    IvfLmParams P{};
    P.metric = metric_type;
    P.nq = ni;
    P.nprobe = np;
    P.d = d;
    P.dpad = dpad_;
    P.nlist = nlist;
    P.k = k;
    P.xq = xq_pad;
    P.ldq = dpad_;
    P.coarse_ids = c_ids;
    P.coarse_dis = c_dis;
    P.list_len = d_list_len_.as<uint32_t>();
    P.list_start = d_list_start_.as<int64_t>();
    fill_lm_(P);
    lm_prefix_.ensure((size_t)ni * (np + 1) * 4 * 2);
    lm_p0_.ensure((size_t)ni * 4);
    lm_cnt_.ensure((size_t)ni * 4);
    lm_bucket_.ensure((size_t)4 * nlist * 4);
    lm_bstart_.ensure((size_t)(2 * nlist + 1) * 4);
    lm_pairs_.ensure((size_t)npairs * 4);
    lm_items_.ensure((size_t)max_items * sizeof(IvfLmItem));
    lm_bounds_.ensure(kLmBoundsBytes);
    lm_thr_.ensure((size_t)ni * 4);
    lm_keys_.ensure((size_t)ni * stride * 8);
    lm_ovf_.ensure((size_t)(ni + 1) * 4);
    if (!h_lm_) HIP_CHECK(hipHostMalloc((void**)&h_lm_, 64, hipHostMallocDefault));
    P.prefix = lm_prefix_.as<uint32_t>();
    P.prefix1 = P.prefix + (size_t)ni * (np + 1);
    P.p0 = lm_p0_.as<uint32_t>();
    P.cnt = lm_cnt_.as<uint32_t>();
    P.bucket_cnt = lm_bucket_.as<uint32_t>();
    P.bucket_fill = P.bucket_cnt + 2 * nlist;
    P.bucket_start = lm_bstart_.as<uint32_t>();
    P.pairs = lm_pairs_.as<uint32_t>();
    P.items = lm_items_.as<IvfLmItem>();
    P.item_bounds = lm_bounds_.as<uint32_t>();
    P.max_items = (int)max_items;
    P.rows_per_item = RT;
    P.qpi = qpi;
    P.force_all = force_all ? 1 : 0;
    // timing experiments only (results are WRONG with it): honoured only under FAISS_AMD_EXPERIMENTS=1, read once
    static const char* dbg_env = experiment_env("FAISS_AMD_LM_DBG");
    P.min_p1 = min_p1;
    P.dbg = dbg_env ? atoi(dbg_env) : 0;
    P.keys = lm_keys_.as<unsigned long long>();
    P.stride = stride;
    P.thr = lm_thr_.as<uint32_t>();
    P.ovf = lm_ovf_.as<uint32_t>();
    if (P.kind == 0 && metric_type == METRIC_L2) {
        lm_qn_.ensure((size_t)ni * 4);
        launch_l2_norms(xq_pad, dpad_, ni, d, lm_qn_.as<float>(), R.stream);
        P.xqn = lm_qn_.as<float>();
    }
    {
        SpanGuard sg(&R, "ivf_lm_plan");
        launch_ivf_lm_plan(P, R.stream);
    }
    const int grid = ivf_lm_grid_blocks(P, R.num_cus); // one wave of persistent workgroups
    {
        SpanGuard sg(&R, "ivf_lm_scan_pass1");
        launch_ivf_lm_scan(P, 1, grid, R.stream);
    }
    SelectParams sp{};
    sp.metric = metric_type;
    sp.nq = ni;
    sp.k = k;
    sp.keys = P.keys;
    sp.q_stride = stride;
    sp.nseg = 1;
    sp.seg_stride = 0;
    sp.seg_cnt = P.cnt;
    if (!force_all) {
        {
            // bound of every query: the k-th smallest key among its pass-1 rows; the segment keeps only those k keys
            // (all of pass 1 that can still win): pass 2 appends behind them.  (A variant that holds the keys in LDS --
            // one read of the segment instead of one per radix pass -- measured slower: 64 KB of LDS per workgroup
            // leaves two of them per CU, profiles/r03_b_listmajor_experiments.txt.)
            SpanGuard sg(&R, "ivf_lm_threshold");
            sp.mode = 0;
            sp.kth_out = P.thr;
            sp.cnt_out = P.cnt;
            sp.max_cnt = c1max; // rows of pass 1: few enough for the wavefront-per-query kernel unless lists are long
            launch_select_k(sp, R.stream);
            sp.kth_out = nullptr;
            sp.cnt_out = nullptr;
        }
        HIP_CHECK(hipMemsetAsync(P.ovf, 0, 4, R.stream));
        {
            SpanGuard sg(&R, "ivf_lm_scan_pass2");
            launch_ivf_lm_scan(P, 2, grid, R.stream);
        }
        launch_ivf_lm_clamp(P, R.stream);
    }
    sp.mode = 1;
    sp.max_cnt = stride;
    sp.nprobe = np;
    sp.ivf_prefix = P.prefix;
    sp.coarse_ids = c_ids;
    sp.list_start = d_list_start_.as<int64_t>();
    sp.arena_ids = arena_ids_.as<int64_t>();
    sp.out_dis = dD;
    sp.out_ids = dI;
    {
        SpanGuard sg(&R, "select_k_kernel");
        launch_select_k(sp, R.stream);
    }
    // one read-back: queries whose segment overflowed in pass 2 (+ the item-table check)
    h_lm_[0] = 0;
    if (!force_all) HIP_CHECK(hipMemcpyAsync(&h_lm_[0], P.ovf, 4, hipMemcpyDeviceToHost, R.stream));
    HIP_CHECK(hipMemcpyAsync(&h_lm_[1], P.item_bounds + 3, 4, hipMemcpyDeviceToHost, R.stream));
    R.sync();
    FA_THROW_IF_NOT_MSG(h_lm_[1] == 0, "list-major scan: work-item table too small (internal error)");
    const int novf = (int)h_lm_[0];
    lm_overflows_ += novf;
    if (novf > 0) {
        // redo those queries with 16 x the room, and what still overflows with every probe in pass 1 (all their rows
        // written, exact capacity): same arithmetic, same answer as an unbounded segment would have given
        DevBuf olist, gq, gids, gdis, gD, gI;
        olist.ensure((size_t)novf * 4);
        HIP_CHECK(hipMemcpyAsync(olist.p, P.ovf + 1, (size_t)novf * 4, hipMemcpyDeviceToDevice, R.stream));
        gq.ensure((size_t)novf * dpad_ * 4);
        gids.ensure((size_t)novf * np * 8);
        gdis.ensure((size_t)novf * np * 4);
        gD.ensure((size_t)novf * k * 4);
        gI.ensure((size_t)novf * k * 8);
        launch_gather_rows(xq_pad, dpad_, dpad_, olist.as<uint32_t>(), novf, gq.as<float>(), R.stream);
        launch_gather_rows((const float*)c_ids, 2 * np, 2 * np, olist.as<uint32_t>(), novf, (float*)gids.p, R.stream);
        launch_gather_rows(c_dis, np, np, olist.as<uint32_t>(), novf, gdis.as<float>(), R.stream);
        search_listmajor_(novf, gq.as<float>(), gids.as<idx_t>(), gdis.as<float>(), np, k, gD.as<float>(), gI.as<idx_t>(), level + 1);
        launch_scatter_results(gD.as<float>(), gI.as<idx_t>(), k, olist.as<uint32_t>(), novf, dD, dI, R.stream);
        R.sync(); // the gathered buffers die with this scope
    }
}

// One chunk of queries through the filter path: plan -> sweep 1 (granule minima) -> bound -> sweep 2 (collect) -> exact
// rerank of the candidates -> k-selection.  rt_g = rows per item | granule blocks << 16.
void GpuIndexIVF::search_listmajor_filter_chunk_(int ni, int q0, const float* xq_pad, const idx_t* c_ids, const float* c_dis, int np,
                                                 int k, float* dD, idx_t* dI, int64_t stride, int rt_g, int64_t gstride,
                                                 int min_stride, std::vector<uint32_t>& redo) const {
    const GpuResources& R = *res_;
    const int RT = rt_g & 0xffff, G = rt_g >> 16;
    const int sample_rows = min_stride >> 4; // (packed by search_listmajor_)
    min_stride &= 15;
    int64_t sum_nrt = 0, nrt_max = 1;
    uint32_t max_len = 1;
    for (auto l : list_len_) {
        const int64_t nrt = (int64_t)div_up(l, (size_t)RT);
        sum_nrt += nrt;
        nrt_max = std::max(nrt_max, nrt);
        max_len = std::max(max_len, l);
    }
    const int64_t npairs = (int64_t)ni * np;
    const int qpi = ivf_lmf_queries_per_item(lmf_sweep_kind_(), d);
    const int64_t max_items = nrt_max * (int64_t)div_up((size_t)npairs, (size_t)qpi) + 2 * sum_nrt + 16;
    FA_THROW_IF_NOT_MSG(max_items < ((int64_t)1 << 30), "list-major scan: too many work items");

    IvfLmParams P{};
    P.metric = metric_type;
    P.nq = ni;
    P.nprobe = np;
    P.d = d;
    P.dpad = dpad_;
    P.nlist = nlist;
    P.k = k;
    P.xq = xq_pad;
    P.ldq = dpad_;
    P.coarse_ids = c_ids;
    P.coarse_dis = c_dis;
    P.list_len = d_list_len_.as<uint32_t>();
    P.list_start = d_list_start_.as<int64_t>();
    fill_lm_(P);
    FA_THROW_IF_NOT(lmf_prepare_(P)); // (built by search_core_; here it only fills the fields)
    lm_prefix_.ensure((size_t)ni * (np + 1) * 4 * 2);
    lm_prefixg_.ensure((size_t)ni * (np + 1) * 4);
    lm_p0_.ensure((size_t)ni * 4);
    lm_cnt_.ensure((size_t)ni * 4);
    lm_bucket_.ensure((size_t)4 * nlist * 4);
    lm_bstart_.ensure((size_t)(2 * nlist + 1) * 4);
    lm_pairs_.ensure((size_t)npairs * 4);
    lm_items_.ensure((size_t)max_items * sizeof(IvfLmItem));
    lm_bounds_.ensure(kLmBoundsBytes);
    lm_thrf_.ensure((size_t)ni * 4);
    lm_keys_.ensure((size_t)ni * stride * 8);
    lm_candpr_.ensure((size_t)ni * stride * 2);
    lm_rowbase_.ensure((size_t)ni * np * 8);
    lm_gmin_.ensure((size_t)ni * gstride * 4);
    lm_ovf_.ensure((size_t)(ni + 1) * 4);
    lm_qflags_.ensure((size_t)ni * 4);
    lm_xnb_.ensure((size_t)ni * 4);
    lm_qn_.ensure((size_t)ni * 4);
    lm_scalar_.ensure(64);
    if (!h_lm_) HIP_CHECK(hipHostMalloc((void**)&h_lm_, 64, hipHostMallocDefault));
    P.prefix = lm_prefix_.as<uint32_t>();
    P.prefix1 = P.prefix + (size_t)ni * (np + 1);
    P.prefixg = lm_prefixg_.as<uint32_t>();
    P.p0 = lm_p0_.as<uint32_t>();
    P.cnt = lm_cnt_.as<uint32_t>();
    P.bucket_cnt = lm_bucket_.as<uint32_t>();
    P.bucket_fill = P.bucket_cnt + 2 * nlist;
    P.bucket_start = lm_bstart_.as<uint32_t>();
    P.pairs = lm_pairs_.as<uint32_t>();
    P.items = lm_items_.as<IvfLmItem>();
    P.item_bounds = lm_bounds_.as<uint32_t>();
    P.max_items = (int)max_items;
    P.rows_per_item = RT;
    P.qpi = qpi;
    P.filter = 1;
    P.gran_blocks = G;
    P.min_stride = min_stride;
    P.sample_rows = sample_rows;
    lm_errf_.ensure((size_t)ni * 4);
    P.err_f = lm_errf_.as<float>();
    P.gmin = lm_gmin_.as<uint32_t>();
    P.gstride = gstride;
    P.thr_f = lm_thrf_.as<float>();
    P.keys = lm_keys_.as<unsigned long long>();
    P.cand_pr = lm_candpr_.as<uint16_t>();
    P.row_base = lm_rowbase_.as<int64_t>();
    P.stride = stride;
    P.ovf = lm_ovf_.as<uint32_t>();
    P.qflags = lm_qflags_.as<uint32_t>();
    P.sel_mask = cur_sel_mask_;
    P.lmf_pair = lmf_pair;
    P.coarse_bad = cur_coarse_bad_ ? cur_coarse_bad_ + q0 : nullptr;
    // ---- queries: fp16 copy + range flags + |q|^2 (the sequential chain of the flat index)
    {
        SpanGuard sg(&R, "ivf_lmf_prepare");
        const int dh = (P.kind != 1 || P.lmf_pairb) ? ivf_lmf_row_halfs(d) : (int)round_up(d, 16);
        lm_q16_.ensure((size_t)ni * dh * 2);
        // the search's scratch counters, zeroed by ONE launch (the query preparation itself where there is one) instead of a
        // fillBuffer packet each: the redo count, the plan's bucket counters, the norm bounds the prepare kernels raise atomically,
        // the scalar quantizer's flags
        const bool pair_ops = P.kind == 2 || P.lmf_pairb;
        if (pair_ops) lm_an_.ensure((size_t)ni * 4);
        ClearList cl{};
        cl.add(P.ovf, 1);
        cl.add(P.bucket_cnt, (size_t)4 * nlist);
        if (P.kind != 0) cl.add(lm_xnb_.p, (size_t)ni);
        if (pair_ops) cl.add(lm_an_.p, (size_t)ni);
        if (P.kind == 2) cl.add(lm_qflags_.p, (size_t)ni);
        P.pre_cleared = 1;
        // (the scalar quantizer's operands, flags and norms all come from launch_ivf_lmf_sq_prepare below)
        if (P.kind != 2)
            launch_prep_queries(xq_pad, dpad_, ni, d, dpad_, lm_q16_.p, dh, lm_qflags_.as<uint32_t>(), lm_qn_.as<float>(),
                                lm_scalar_.as<unsigned>(), R.stream, &cl);
        else
            launch_clear_words(cl, R.stream);
        P.xq16 = lm_q16_.p;
        P.ldq16 = dh;
        P.xqn = lm_qn_.as<float>();
        P.xn_full = lm_qn_.as<float>();
        if (P.kind == 1 && !P.lmf_pairb) {
            lm_pair16_.ensure((size_t)ni * (metric_type == METRIC_L2 ? np : 1) * d * 2);
            lm_pairxh_.ensure((size_t)ni * np * 4);
            P.pair16 = lm_pair16_.p;
            P.pair_xh = lm_pairxh_.as<float>();
            launch_ivf_lmf_pq_prepare(P, lm_xnb_.as<float>(), R.stream);
        } else if (P.kind == 2 || P.lmf_pairb) {
            // scalar quantizer: B operands (a o s as fp16) and query terms per (query, probe) pair, flags, norm bounds
            lm_pair16_.ensure((size_t)ni * np * dh * 2);
            lm_pairxh_.ensure((size_t)ni * np * 4);
            lm_an_.ensure((size_t)ni * 4);
            P.pair16 = lm_pair16_.p;
            P.pair_xh = lm_pairxh_.as<float>();
            P.ldq16 = dh;
            P.an_bound = lm_an_.as<float>();
            launch_ivf_lmf_sq_prepare(P, lm_xnb_.as<float>(), lm_an_.as<float>(), R.stream);
        }
        // granule slots nobody writes must never look like good estimates
        HIP_CHECK(hipMemsetAsync(P.gmin, 0xff, (size_t)ni * gstride * 4, R.stream));
    }
    {
        SpanGuard sg(&R, "ivf_lm_plan");
        launch_ivf_lm_plan(P, R.stream);
    }
    const int grid = ivf_lmf_grid_blocks(P, R.num_cus);
    {
        SpanGuard sg(&R, "ivf_lmf_sweep_min");
        launch_ivf_lmf_sweep(P, 1, grid, R.stream);
    }
    {
        SpanGuard sg(&R, "ivf_lmf_bound");
        launch_ivf_lmf_bound(P, P.kind != 0 ? lm_xnb_.as<float>() : lm_qn_.as<float>(), R.stream);
    }
    {
        SpanGuard sg(&R, "ivf_lmf_sweep_collect");
        launch_ivf_lmf_sweep(P, 2, grid, R.stream);
    }
    static const bool lmf_stats = experiment_env("FAISS_AMD_LMF_STATS") != nullptr; // diagnostics only: candidates sweep 2 collected
    if (lmf_stats) {
        std::vector<uint32_t> hc((size_t)ni);
        HIP_CHECK(hipMemcpyAsync(hc.data(), P.cnt, (size_t)ni * 4, hipMemcpyDeviceToHost, R.stream));
        R.sync();
        double sum = 0;
        uint32_t mx = 0;
        for (uint32_t c : hc) sum += c, mx = std::max(mx, c);
        fprintf(stderr, "ivf_lmf sweep 2: %.1f candidates per query (max %u, stride %lld)\n", sum / std::max<int64_t>(ni, 1), mx, (long long)stride);
    }
    // the workgroup that re-derives a query's candidates also selects its k best when both fit its LDS (no selection launch)
    // (IVFFlat: rerank 0.135 -> 0.19 ms for 0.105 ms of selection launch at nb = 1M; IVFPQ keeps the separate launch: its rerank
    // workgroups -- two per CU, the 64 KB table -- serialise the tail: 0.25 -> 0.52 ms; and a selection inside the wave-per-query
    // kernel -- ranks by v_readlane counting, k = 100 -- cost that kernel the 0.065 ms the selection launch takes: 0.092 + 0.063
    // -> 0.158).  The tightening launch leaves a query
    // with more than kLmfFusedSelectN candidates (that many rows inside the band of its k-th best) to the redo path.
    static const char* pqfs = experiment_env("FAISS_AMD_LMF_PQ_FUSED_SELECT"); // timing experiment
    const bool fused_select = (P.kind != 1 || (pqfs && atoi(pqfs) == 1)) && k <= kLmfFusedSelectK;
    // (IVFPQ through the decoded-residual copy prepared its operands like the scalar quantizer; everything from here on is IVFPQ's)
    {
        // clamp of overflowed segments + the smallest superset the band allows (launch_ivf_lmf_tighten)
        SpanGuard sg(&R, "ivf_lmf_tighten");
        static const char* notight = experiment_env("FAISS_AMD_LMF_NOTIGHTEN"); // diagnostics: keep every collected candidate
        if (notight) HIP_CHECK(hipMemsetAsync(P.err_f, 0x7f, (size_t)ni * 4, R.stream)); // (huge band: nothing is cut)
        launch_ivf_lmf_tighten(P, fused_select ? kLmfFusedSelectN : 0, R.stream);
    }
    if (fused_select) {
        P.fin_dis = dD;
        P.fin_ids = dI;
        P.arena_ids = arena_ids_.as<int64_t>();
    }
    {
        SpanGuard sg(&R, "ivf_lmf_rerank");
        static const char* rr_old = experiment_env("FAISS_AMD_LMF_RERANK_WG"); // A/B: 1 = the workgroup-per-query kernel
        if (P.kind == 1 && !(rr_old && atoi(rr_old) == 1)) P.rr_blocks = R.num_cus;
        if (P.kind == 2) launch_ivf_lmf_rerank_sq(P, R.stream);
        else launch_ivf_lmf_rerank(P, R.stream);
    }
    SelectParams sp{};
    sp.metric = metric_type;
    sp.nq = ni;
    sp.k = k;
    sp.keys = P.keys;
    sp.q_stride = stride;
    sp.nseg = 1;
    sp.seg_stride = 0;
    sp.seg_cnt = P.cnt;
    sp.mode = 1;
    sp.max_cnt = stride;
    sp.nprobe = np;
    sp.ivf_prefix = P.prefix;
    sp.coarse_ids = c_ids;
    sp.list_start = d_list_start_.as<int64_t>();
    sp.arena_ids = arena_ids_.as<int64_t>();
    sp.out_dis = dD;
    sp.out_ids = dI;
    if (!fused_select) {
        SpanGuard sg(&R, "select_k_kernel");
        launch_select_k(sp, R.stream);
    }
    // one read-back: queries to redo (+ the item-table check)
    HIP_CHECK(hipMemcpyAsync(&h_lm_[0], P.ovf, 4, hipMemcpyDeviceToHost, R.stream));
    HIP_CHECK(hipMemcpyAsync(&h_lm_[1], P.item_bounds + 3, 4, hipMemcpyDeviceToHost, R.stream));
    R.sync();
    FA_THROW_IF_NOT_MSG(h_lm_[1] == 0, "list-major scan: work-item table too small (internal error)");
    const int novf = (int)h_lm_[0];
    lm_overflows_ += novf;
    if (novf > 0) {
        std::vector<uint32_t> ol((size_t)novf);
        HIP_CHECK(hipMemcpy(ol.data(), P.ovf + 1, (size_t)novf * 4, hipMemcpyDeviceToHost));
        for (uint32_t q : ol) redo.push_back(q + (uint32_t)q0);
    }
}

void GpuIndexIVF::test_filter_dump(idx_t n, const float* x, int nprobe_now, idx_t k, int64_t stride, unsigned long long* keys_out,
                                   float* band_out) const {
    FA_THROW_IF_NOT_MSG(is_trained && nstored_ > 0 && n >= 1 && n <= 65536 && x && keys_out && band_out, "bad arguments");
    FA_THROW_IF_NOT_MSG(lmf_capable_(), "the f16 filter does not serve this index type / shape");
    std::lock_guard<std::mutex> g(mu_);
    res_->set_device();
    const GpuResources& R = *res_;
    const int ni = (int)n, np = std::min(nprobe_now, nlist);
    uint32_t max_len = 1;
    for (auto l : list_len_) max_len = std::max(max_len, l);
    FA_THROW_IF_NOT_MSG(stride >= (int64_t)np * max_len, "stride below nprobe x the longest list");
    q_pad_.ensure((size_t)ni * dpad_ * 4);
    stage_padded(R, x, ni, d, dpad_, q_raw_, q_pad_.as<float>());
    c_dis_.ensure((size_t)ni * np * 4);
    c_ids_.ensure((size_t)ni * np * 8);
    quantizer->search_device(ni, q_pad_.as<float>(), np, c_dis_.as<float>(), c_ids_.as<idx_t>());
    const int RT = kLmRowsPerItem, G = 1;
    int64_t sum_nrt = 0, nrt_max = 1;
    for (auto l : list_len_) {
        const int64_t nrt = (int64_t)div_up(l, (size_t)RT);
        sum_nrt += nrt;
        nrt_max = std::max(nrt_max, nrt);
    }
    const int64_t npairs = (int64_t)ni * np;
    const int qpi = ivf_lmf_queries_per_item(lmf_sweep_kind_(), d);
    const int64_t max_items = nrt_max * (int64_t)div_up((size_t)npairs, (size_t)qpi) + 2 * sum_nrt + 16;
    const int64_t gstride = 2 * (int64_t)np * (int64_t)div_up((size_t)max_len, (size_t)(32 * G));
    IvfLmParams P{};
    P.metric = metric_type;
    P.nq = ni;
    P.nprobe = np;
    P.d = d;
    P.dpad = dpad_;
    P.nlist = nlist;
    P.k = (int)k;
    P.xq = q_pad_.as<float>();
    P.ldq = dpad_;
    P.coarse_ids = c_ids_.as<idx_t>();
    P.coarse_dis = c_dis_.as<float>();
    P.list_len = d_list_len_.as<uint32_t>();
    P.list_start = d_list_start_.as<int64_t>();
    fill_lm_(P);
    FA_THROW_IF_NOT_MSG(lmf_prepare_(P), "stored values outside the fp16 range");
    DevBuf prefix, prefixg, p0, cnt, bucket, bstart, pairs, items, bounds, thrf, keys, gmin, ovf, qflags, xnb, qn, q16, scalar, grid,
            band;
    prefix.ensure((size_t)ni * (np + 1) * 8);
    prefixg.ensure((size_t)ni * (np + 1) * 4);
    p0.ensure((size_t)ni * 4);
    cnt.ensure((size_t)ni * 4);
    bucket.ensure((size_t)4 * nlist * 4);
    bstart.ensure((size_t)(2 * nlist + 1) * 4);
    pairs.ensure((size_t)npairs * 4);
    items.ensure((size_t)max_items * sizeof(IvfLmItem));
    bounds.ensure(kLmBoundsBytes);
    thrf.ensure((size_t)ni * 4);
    keys.ensure((size_t)ni * stride * 8);
    gmin.ensure((size_t)ni * gstride * 4);
    ovf.ensure((size_t)(ni + 1) * 4);
    qflags.ensure((size_t)ni * 4);
    xnb.ensure((size_t)ni * 4);
    qn.ensure((size_t)ni * 4);
    scalar.ensure(64);
    band.ensure((size_t)ni * 4);
    HIP_CHECK(hipMemcpyAsync(band.p, band_out, (size_t)ni * 4, hipMemcpyHostToDevice, R.stream));
    HIP_CHECK(hipMemsetAsync(keys.p, 0xff, (size_t)ni * stride * 8, R.stream));
    HIP_CHECK(hipMemsetAsync(gmin.p, 0xff, (size_t)ni * gstride * 4, R.stream));
    P.prefix = prefix.as<uint32_t>();
    P.prefix1 = P.prefix + (size_t)ni * (np + 1);
    P.prefixg = prefixg.as<uint32_t>();
    P.p0 = p0.as<uint32_t>();
    P.cnt = cnt.as<uint32_t>();
    P.bucket_cnt = bucket.as<uint32_t>();
    P.bucket_fill = P.bucket_cnt + 2 * nlist;
    P.bucket_start = bstart.as<uint32_t>();
    P.pairs = pairs.as<uint32_t>();
    P.items = items.as<IvfLmItem>();
    P.item_bounds = bounds.as<uint32_t>();
    P.max_items = (int)max_items;
    P.rows_per_item = RT;
    P.qpi = qpi;
    P.filter = 1;
    P.gran_blocks = G;
    P.min_stride = 1;
    P.gmin = gmin.as<uint32_t>();
    P.gstride = gstride;
    P.thr_f = thrf.as<float>();
    P.keys = keys.as<unsigned long long>();
    P.stride = stride;
    P.ovf = ovf.as<uint32_t>();
    P.qflags = qflags.as<uint32_t>();
    P.band_out = band.as<float>();
    const int dh = (P.kind != 1 || P.lmf_pairb) ? ivf_lmf_row_halfs(d) : (int)round_up(d, 16);
    q16.ensure((size_t)ni * dh * 2);
    launch_prep_queries(P.xq, dpad_, ni, d, dpad_, q16.p, dh, qflags.as<uint32_t>(), qn.as<float>(), scalar.as<unsigned>(), R.stream);
    P.xq16 = q16.p;
    P.ldq16 = dh;
    P.xqn = qn.as<float>();
    P.xn_full = qn.as<float>();
    DevBuf pair16, pairxh;
    if (P.kind == 1 && !P.lmf_pairb) {
        pair16.ensure((size_t)ni * (metric_type == METRIC_L2 ? np : 1) * d * 2);
        pairxh.ensure((size_t)ni * np * 4);
        P.pair16 = pair16.p;
        P.pair_xh = pairxh.as<float>();
        launch_ivf_lmf_pq_prepare(P, xnb.as<float>(), R.stream);
    }
    DevBuf anb;
    if (P.kind == 2 || P.lmf_pairb) {
        pair16.ensure((size_t)ni * np * dh * 2);
        pairxh.ensure((size_t)ni * np * 4);
        anb.ensure((size_t)ni * 4);
        P.pair16 = pair16.p;
        P.pair_xh = pairxh.as<float>();
        P.an_bound = anb.as<float>();
        launch_ivf_lmf_sq_prepare(P, xnb.as<float>(), anb.as<float>(), R.stream);
    }
    launch_ivf_lm_plan(P, R.stream);
    const int gb = ivf_lmf_grid_blocks(P, R.num_cus);
    launch_ivf_lmf_sweep(P, 1, gb, R.stream);
    launch_ivf_lmf_bound(P, P.kind != 0 ? xnb.as<float>() : qn.as<float>(), R.stream);
    launch_ivf_lmf_sweep(P, 3, gb, R.stream);
    HIP_CHECK(hipMemcpyAsync(keys_out, keys.p, (size_t)ni * stride * 8, hipMemcpyDeviceToHost, R.stream));
    HIP_CHECK(hipMemcpyAsync(band_out, band.p, (size_t)ni * 4, hipMemcpyDeviceToHost, R.stream));
    R.sync();
}

// ---------------------------------------------------------------------- IVF scalar quantizer
GpuIndexIVFScalarQuantizer::GpuIndexIVFScalarQuantizer(std::shared_ptr<GpuResources> res, int dims, int nlist, int qtype_,
                                                       int metric, bool encode_residual, GpuIndexFlat* coarse_quantizer,
                                                       bool coarse_f16, int indices_options_)
        : GpuIndexIVF(std::move(res), dims, metric, nlist, coarse_quantizer, coarse_f16, indices_options_), qtype(qtype_),
          by_residual(encode_residual) {
    // the types GpuIndexIVFScalarQuantizer accepts (faiss/gpu/impl/GpuScalarQuantizer.cuh:20-33 isSQSupported)
    dsq_ = (int)round_up(dims, 16);
    switch (qtype) {
        case QT_8bit:
        case QT_8bit_uniform:
            ct_ = SQ_U8, levels_ = 255.f, code_size = (size_t)dims;
            break;
        case QT_8bit_direct:
            ct_ = SQ_U8, levels_ = 0.f, code_size = (size_t)dims;
            break;
        case QT_4bit:
        case QT_4bit_uniform:
            ct_ = SQ_U4, levels_ = 15.f, code_size = ((size_t)dims + 1) / 2;
            break;
        case QT_6bit:
            ct_ = SQ_U6, levels_ = 63.f, code_size = ((size_t)dims * 6 + 7) / 8;
            break;
        case QT_fp16:
            ct_ = SQ_F16, levels_ = 0.f, code_size = (size_t)dims * 2;
            break;
        default:
            FA_THROW_MSG("unsupported scalar quantizer type (reference: GpuScalarQuantizer.cuh isSQSupported)");
    }
    FA_THROW_IF_NOT_MSG(dims <= 1024, "scalar-quantizer index: d <= 1024");
    code_bytes_ = (size_t)(dsq_ / 16) * sq_chunk_bytes(ct_); // arena row: whole 16-component chunks, zero padded
    granule_ = 64;                                           // 64-row chunk-major blocks (kernels.h sq_code_offset)
    // |s o code|^2 per stored row: the second L2 term of the list-major scan
    use_rn_ = metric == METRIC_L2 && (ivf_lm_supported(2, dpad_, ct_, dims) || ivf_lmf_supported(2, dims, dpad_, ct_));
    res_->set_device();
    sq_zero_.ensure((size_t)dsq_ * 4);
    HIP_CHECK(hipMemset(sq_zero_.p, 0, (size_t)dsq_ * 4));
    upload_tables_();
    adopt_quantizer_();
}

// decoder tables: x^_i = fmaf(code_i, s_i, b_i) with s = vdiff / levels, b = vmin + s / 2 -- the reconstruction
// vmin + (code + 0.5) / levels * vdiff of faiss/impl/scalar_quantizer/quantizers.h:92-150, codecs.h:36-58
void GpuIndexIVFScalarQuantizer::upload_tables_() {
    res_->set_device();
    std::vector<float> vmin(d, 0.f), vdiff(d, 0.f), s(dsq_, 0.f), b(dsq_, 0.f);
    if (needs_training_()) {
        if (trained.empty()) return;
        const bool uniform = qtype == QT_8bit_uniform || qtype == QT_4bit_uniform;
        FA_THROW_IF_NOT_MSG(trained.size() == (uniform ? 2u : 2u * (size_t)d), "scalar quantizer: wrong size of `trained`");
        for (int i = 0; i < d; i++) {
            vmin[i] = uniform ? trained[0] : trained[i];
            vdiff[i] = uniform ? trained[1] : trained[(size_t)d + i];
            s[i] = vdiff[i] / levels_;
            b[i] = vmin[i] + 0.5f * s[i];
        }
    } else if (qtype == QT_8bit_direct || qtype == QT_fp16) {
        // the code is the value (the query-major kernel never reads the tables of fp16 codes; the list-major one folds
        // s and b into its query operands whatever the type: a * 1 == a, fmaf(q, 0, acc) == acc)
        for (int i = 0; i < d; i++) s[i] = 1.f;
    }
    vmin_.ensure((size_t)d * 4);
    vdiff_.ensure((size_t)d * 4);
    sq_s_.ensure((size_t)dsq_ * 4);
    sq_b_.ensure((size_t)dsq_ * 4);
    HIP_CHECK(hipMemcpy(vmin_.p, vmin.data(), (size_t)d * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(vdiff_.p, vdiff.data(), (size_t)d * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(sq_s_.p, s.data(), (size_t)dsq_ * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(sq_b_.p, b.data(), (size_t)dsq_ * 4, hipMemcpyHostToDevice));
    // list-major scan: offsets of the CENTRED codes, b' = fmaf(mid, s, b) (kernels.h IvfLmParams)
    {
        const float mid = ct_ == SQ_U8 ? 127.5f : ct_ == SQ_U4 ? 7.5f : ct_ == SQ_U6 ? 31.5f : 0.f;
        std::vector<float> bm(dsq_, 0.f);
        for (int i = 0; i < d; i++) bm[i] = std::fmaf(mid, s[i], b[i]);
        sq_bm_.ensure((size_t)dsq_ * 4);
        HIP_CHECK(hipMemcpy(sq_bm_.p, bm.data(), (size_t)dsq_ * 4, hipMemcpyHostToDevice));
        double bn = 0.0;
        for (int i = 0; i < d; i++) bn += (double)bm[i] * (double)bm[i];
        sq_bn_ = (float)(bn * 1.0001);
    }
    // the scale changed under rows that are already stored (set_trained after copy_lists): their norms follow
    if (use_rn_ && nstored_ > 0 && arena_.p) row_norms_all_();
    shadow_dirty_ = true; // (the filter path's norm bounds are read from the norms when its copy of the lists is written)
}
void GpuIndexIVFScalarQuantizer::row_norms_all_() {
    if (!use_rn_ || arena_rows_ == 0) return;
    launch_ivfsq_row_norms(arena_.as<uint8_t>(), ct_, (int)code_bytes_, d, sq_s_.as<float>(), nullptr, 0, arena_rows_,
                           arena_rn_.as<float>(), res_->stream);
    res_->sync();
}
void GpuIndexIVFScalarQuantizer::lists_changed_() {
    // bulk load (copy_lists / compaction into a new arena): norms of every arena row (the slack holds whatever it holds)
    row_norms_all_();
}
bool GpuIndexIVFScalarQuantizer::lm_capable_() const {
    return ivf_lm_supported(2, dpad_, ct_, d);
}
void GpuIndexIVFScalarQuantizer::fill_lm_(IvfLmParams& p) const {
    p.kind = 2;
    p.arena_codes = arena_.as<uint8_t>();
    p.arena_rn = arena_rn_.as<float>();
    p.M = ct_; // (ivf_lm_supported's third argument for this kind)
    p.centroids = centroids_dev_();
    p.ldc = dpad_;
    p.sq_ct = ct_;
    p.sq_ld = (int)code_bytes_;
    p.sq_by_residual = by_residual ? 1 : 0;
    p.sq_s = sq_s_.as<float>();
    p.sq_b = sq_bm_.as<float>();
    p.sq_zero = sq_zero_.as<float>();
}
bool GpuIndexIVFScalarQuantizer::lmf_capable_() const {
    return ivf_lmf_supported(2, d, dpad_, ct_);
}
void GpuIndexIVFScalarQuantizer::lmf_shadow_room_() const {
    const size_t need = ((size_t)arena_cap_rows_ / 32 + 10) * (size_t)(ivf_lmf_row_halfs(d) / 16) * 1024;
    if (need > arena_h_.cap) arena_h_.ensure(need, shadow_dirty_ ? 0 : arena_h_.cap, res_->stream);
}
// the copy of the lists the sweeps read (whole, or the blocks add() touched) + the norm bounds of the error band
void GpuIndexIVFScalarQuantizer::lmf_write_copy_(const uint32_t* d_first_row, bool merge) const {
    const GpuResources& R = *res_;
    lmf_shadow_room_();
    lm_scalar_.ensure(64);
    HIP_CHECK(hipMemsetAsync(lm_scalar_.p, 0, 8, R.stream));
    launch_ivf_lmf_sq_shadow(arena_.as<uint8_t>(), ct_, (int)code_bytes_, use_rn_ ? arena_rn_.as<float>() : nullptr, d, nlist,
                             d_list_len_.as<uint32_t>(), d_list_start_.as<int64_t>(), arena_h_.p, ivf_lmf_row_halfs(d),
                             lm_scalar_.as<unsigned>(), d_first_row, R.stream);
    unsigned bits[2] = {0, 0};
    HIP_CHECK(hipMemcpyAsync(bits, lm_scalar_.p, 8, hipMemcpyDeviceToHost, R.stream));
    R.sync();
    float rn, cn;
    memcpy(&rn, &bits[0], 4);
    memcpy(&cn, &bits[1], 4);
    const bool ok = bits[0] != 0x7f800000u && cn < 3.0e38f;
    if (!merge) shadow_in_range_ = ok, sq_rn_max_ = 0.f, sq_cn_max_ = 0.f;
    else if (!ok) shadow_in_range_ = false;
    if (ok) sq_rn_max_ = std::max(sq_rn_max_, rn), sq_cn_max_ = std::max(sq_cn_max_, cn);
}
void GpuIndexIVFScalarQuantizer::lmf_patch_(const uint32_t* d_first_row) {
    lmf_write_copy_(d_first_row, true);
}
bool GpuIndexIVFScalarQuantizer::lmf_prepare_(IvfLmParams& p) const {
    lmf_shadow_room_();
    if (shadow_dirty_) {
        lmf_write_copy_(nullptr, false);
        shadow_dirty_ = false;
    }
    if (!shadow_in_range_) return false;
    const float mid = ct_ == SQ_U8 ? 127.5f : ct_ == SQ_U4 ? 7.5f : ct_ == SQ_U6 ? 31.5f : 0.f;
    p.filter = 1;
    p.arena_h = arena_h_.p;
    p.ldh = ivf_lmf_row_halfs(d);
    p.cmid2 = (float)d * mid * mid;
    p.yn_max = ct_ == SQ_F16 ? sq_cn_max_ : p.cmid2; // |code'|^2 <= d mid^2 for the integer types
    p.rn_max = sq_rn_max_;
    p.bn = sq_bn_;
    p.sq_b_plain = sq_b_.as<float>();
    return true;
}
void GpuIndexIVFScalarQuantizer::set_trained(const float* t, size_t n) {
    FA_THROW_IF_NOT_MSG(needs_training_(), "this scalar quantizer type has no trained range");
    FA_THROW_IF_NOT_MSG(t && n > 0, "null `trained`");
    const bool uniform = qtype == QT_8bit_uniform || qtype == QT_4bit_uniform;
    FA_THROW_IF_NOT_MSG(n == (uniform ? 2u : 2u * (size_t)d), "scalar quantizer: wrong size of `trained`");
    std::lock_guard<std::mutex> g(mu_);
    trained.assign(t, t + n);
    upload_tables_();
    update_is_trained_();
}
// ---- ScalarQuantizer range statistics other than RS_minmax (faiss/impl/scalar_quantizer/training.cpp:209-332 train_Uniform,
// :334-385 train_NonUniform).  The reference's GPU class trains its ScalarQuantizer on the HOST (gpu/GpuIndexIVFScalarQuantizer.cu:
// 96-160: sq.train_residual), and these statistics are sequential float recurrences over <= 100 000 training rows (RS_optim: up to
// 2000 passes of a least-squares fit, every pass a running float sum) -- so this is a host restatement too, column-parallel, of the
// very operation order of the reference as GCC compiles it (-O3 -mfma contracts a * b + c into one fused multiply-add; this file
// is built with -ffp-contract=off, so every fmaf below is one the reference build has): `trained` comes out byte-identical
// (tests/test_oracle_cpu.py::test_sq_rangestat_training_is_byte_identical_to_the_reference, no GPU needed).
// x: n values (stride 1).  k = 2^bits levels.  Returns {vmin, vdiff}.
static void sq_train_uniform_host(int rs, float rs_arg, int64_t n, int k, const float* x, float* out) {
    float vmin, vmax;
    if (rs == 1) { // RS_meanstd
        double sum = 0, sum2 = 0;
        for (int64_t i = 0; i < n; i++) {
            sum += x[i];
            sum2 += x[i] * x[i];
        }
        const float mean = (float)(sum / n);
        const float var = (float)(sum2 / n - (double)(mean * mean));
        const float sd = var <= 0 ? 1.0f : std::sqrt(var);
        const float w = sd * rs_arg; // (used twice: the reference build keeps the product, no fused multiply-add here)
        vmin = mean - w;
        vmax = mean + w;
    } else if (rs == 2) { // RS_quantiles
        std::vector<float> c(x, x + n);
        int64_t o = (int64_t)(rs_arg * n);
        if (o < 0) o = 0;
        if (o > n - o) o = n / 2;
        std::nth_element(c.begin(), c.begin() + o, c.end());
        vmin = c[o];
        std::nth_element(c.begin(), c.begin() + (n - 1 - o), c.end());
        vmax = c[n - 1 - o];
    } else { // RS_optim: alternating least squares of x ~ a * round((x - b) / a) + b
        float a, b, sx = 0;
        vmin = HUGE_VALF, vmax = -HUGE_VALF;
        for (int64_t i = 0; i < n; i++) {
            if (x[i] < vmin) vmin = x[i];
            if (x[i] > vmax) vmax = x[i];
            sx += x[i];
        }
        b = vmin;
        a = (vmax - vmin) / (float)(k - 1);
        float last_err = -1;
        int iter_last_err = 0;
        const float nf = (float)n, kf = (float)k;
        for (int it = 0; it < 2000; it++) {
            float sn = 0, sn2 = 0, sxn = 0, err1 = 0;
            for (int64_t i = 0; i < n; i++) {
                const float xi = x[i];
                float ni = (float)std::floor((double)((xi - b) / a) + 0.5);
                if (ni < 0) ni = 0;
                if (ni >= kf) ni = (float)(k - 1);
                const float u = xi - fmaf(ni, a, b);
                err1 = fmaf(u, u, err1);
                sn += ni;
                sn2 = fmaf(ni, ni, sn2);
                sxn = fmaf(ni, xi, sxn);
            }
            if (err1 == last_err) {
                if (++iter_last_err == 16) break;
            } else {
                last_err = err1;
                iter_last_err = 0;
            }
            const float det = fmaf(sn, sn, -(sn2 * nf));
            const float nb = fmaf(sn, sxn, -(sn2 * sx)) / det;
            const float na = fmaf(sn, sx, -(nf * sxn)) / det;
            b = nb;
            a = na;
        }
        vmin = b;
        vmax = fmaf(a, (float)(k - 1), b);
    }
    out[0] = vmin;
    out[1] = vmax - vmin;
}
// rows: [n][d] dense.  uniform: one range over all n * d values; else one per dimension -- over the column as the reference
// builds it, i.e. with ROW 0 LEFT AT ZERO (training.cpp:366 transposes from i = 1; kept, because `trained` has to match).
void sq_train_rangestat_host(int rangestat, float rangestat_arg, int64_t n, int d, int k, bool uniform, const float* rows,
                             std::vector<float>& trained) {
    FA_THROW_IF_NOT_MSG(rangestat >= 1 && rangestat <= 3, "invalid range statistic");
    FA_THROW_IF_NOT(n > 0 && rows);
    if (uniform) {
        trained.assign(2, 0.f);
        sq_train_uniform_host(rangestat, rangestat_arg, n * d, k, rows, trained.data());
        return;
    }
    trained.assign((size_t)2 * d, 0.f);
    const int nthr = (int)std::max(1u, std::min<unsigned>(std::thread::hardware_concurrency(), (unsigned)d));
    std::vector<std::thread> pool;
    std::atomic<int> next{0};
    for (int t = 0; t < nthr; t++)
        pool.emplace_back([&]() {
            std::vector<float> col((size_t)n);
            float o[2];
            for (int j = next.fetch_add(1); j < d; j = next.fetch_add(1)) {
                col[0] = 0.f;
                for (int64_t i = 1; i < n; i++) col[i] = rows[(size_t)i * d + j];
                sq_train_uniform_host(rangestat, rangestat_arg, n, k, col.data(), o);
                trained[j] = o[0];
                trained[(size_t)d + j] = o[1];
            }
        });
    for (auto& th : pool) th.join();
}

void GpuIndexIVFScalarQuantizer::train_residual_(idx_t n, const float* x_dev_pad) {
    // IndexIVF::train_encoder on at most 100000 vectors (IndexScalarQuantizer.cpp:152-161), residuals when
    // by_residual (IndexIVF.cpp train_encoder path), then ScalarQuantizer::train = per-dimension (or global) range
    // (impl/scalar_quantizer/training.cpp:209-232, 333-365)
    if (!needs_training_()) return;
    FA_THROW_IF_NOT_MSG(rangestat >= 0 && rangestat <= 3, "invalid ScalarQuantizer::RangeStat");
    const GpuResources& R = *res_;
    idx_t nt = std::min<idx_t>(n, 100000);
    DevBuf dsel, dsub_rows, dlab, ddis, dres, dmm;
    const float* xs = x_dev_pad;
    if (nt < n) {
        std::mt19937_64 rng((uint64_t)cp_seed + 0x51ed270b7c3f9a1dull);
        std::vector<uint32_t> perm((size_t)n);
        std::iota(perm.begin(), perm.end(), 0u);
        for (idx_t i = 0; i < nt; i++) {
            const idx_t j = i + (idx_t)(rng() % (uint64_t)(n - i));
            std::swap(perm[i], perm[j]);
        }
        dsel.ensure((size_t)nt * 4);
        HIP_CHECK(hipMemcpyAsync(dsel.p, perm.data(), (size_t)nt * 4, hipMemcpyHostToDevice, R.stream));
        dsub_rows.ensure((size_t)nt * dpad_ * 4);
        launch_gather_rows(x_dev_pad, dpad_, dpad_, dsel.as<uint32_t>(), (int)nt, dsub_rows.as<float>(), R.stream);
        R.sync();
        xs = dsub_rows.as<float>();
    }
    int64_t ldr = dpad_;
    if (by_residual) {
        dlab.ensure((size_t)nt * 8);
        ddis.ensure((size_t)nt * 4);
        dres.ensure((size_t)nt * d * 4);
        quantizer->search_device((int)nt, xs, 1, ddis.as<float>(), dlab.as<idx_t>());
        launch_residual(xs, dpad_, nt, d, dlab.as<idx_t>(), centroids_dev_(), dpad_, dres.as<float>(), d, R.stream);
        xs = dres.as<float>();
        ldr = d;
    }
    const bool uniform = qtype == QT_8bit_uniform || qtype == QT_4bit_uniform;
    if (rangestat != 0) {
        // RS_meanstd / RS_quantiles / RS_optim: the (residual) training rows come back to the host (sq_train_rangestat_host above)
        std::vector<float> rows((size_t)nt * d);
        HIP_CHECK(hipMemcpy2DAsync(rows.data(), (size_t)d * 4, xs, (size_t)ldr * 4, (size_t)d * 4, (size_t)nt, hipMemcpyDeviceToHost,
                                   R.stream));
        R.sync();
        const int bits = (qtype == QT_4bit || qtype == QT_4bit_uniform) ? 4 : qtype == QT_6bit ? 6 : 8;
        sq_train_rangestat_host(rangestat, rangestat_arg, nt, d, 1 << bits, uniform, rows.data(), trained);
        upload_tables_();
        return;
    }
    const int nb = ivfsq_minmax_blocks(nt);
    dmm.ensure((size_t)nb * 2 * d * 4);
    launch_ivfsq_minmax(xs, ldr, nt, d, dmm.as<float>(), R.stream);
    std::vector<float> mm((size_t)nb * 2 * d);
    HIP_CHECK(hipMemcpyAsync(mm.data(), dmm.p, mm.size() * 4, hipMemcpyDeviceToHost, R.stream));
    R.sync();
    std::vector<float> lo(d, INFINITY), hi(d, -INFINITY);
    for (int b = 0; b < nb; b++)
        for (int j = 0; j < d; j++) {
            lo[j] = std::min(lo[j], mm[((size_t)b * 2 + 0) * d + j]);
            hi[j] = std::max(hi[j], mm[((size_t)b * 2 + 1) * d + j]);
        }
    if (uniform) {
        float vmin = INFINITY, vmax = -INFINITY;
        for (int j = 0; j < d; j++) {
            vmin = std::min(vmin, lo[j]);
            vmax = std::max(vmax, hi[j]);
        }
        const float vexp = (vmax - vmin) * rangestat_arg;
        vmin -= vexp;
        vmax += vexp;
        trained = {vmin, vmax - vmin};
    } else {
        trained.assign((size_t)2 * d, 0.f);
        for (int j = 0; j < d; j++) {
            float vmin = lo[j], vmax = hi[j];
            const float vexp = (vmax - vmin) * rangestat_arg;
            vmin -= vexp;
            vmax += vexp;
            trained[j] = vmin;
            trained[(size_t)d + j] = vmax - vmin;
        }
    }
    upload_tables_();
}
void GpuIndexIVFScalarQuantizer::append_(int n, const float* x_pad, const int64_t* d_labels, const int64_t* d_dest) {
    launch_ivfsq_encode_append(qtype, x_pad, dpad_, n, d, d_labels, d_dest, centroids_dev_(), dpad_, by_residual,
                               vmin_.as<float>(), vdiff_.as<float>(), arena_.as<uint8_t>(), (int)code_bytes_, res_->stream);
    if (use_rn_)
        launch_ivfsq_row_norms(arena_.as<uint8_t>(), ct_, (int)code_bytes_, d, sq_s_.as<float>(), d_dest, 0, n,
                               arena_rn_.as<float>(), res_->stream);
}
void GpuIndexIVFScalarQuantizer::fill_fused_(IvfFusedParams& p) const {
    p.arena_codes = arena_.as<uint8_t>();
    p.sq_ct = ct_;
    p.sq_dsq = dsq_;
    p.sq_ld = (int)code_bytes_;
    p.sq_by_residual = by_residual ? 1 : 0;
    p.sq_s = sq_s_.as<float>();
    p.sq_b = sq_b_.as<float>();
    p.centroids = centroids_dev_();
    p.ldc = dpad_;
}
int GpuIndexIVFScalarQuantizer::sq_chunk_bytes_() const {
    return sq_chunk_bytes(ct_);
}
void GpuIndexIVFScalarQuantizer::scan_(int, const float*, int, const int64_t*) const {
    FA_THROW_MSG("the scalar-quantizer scan exists only as the fused kernel");
}

// ---------------------------------------------------------------------- IVFFlat
GpuIndexIVFFlat::GpuIndexIVFFlat(std::shared_ptr<GpuResources> res, int dims, int nlist, int metric, GpuIndexFlat* coarse_quantizer,
                                 bool coarse_f16, int indices_options_)
        : GpuIndexIVF(std::move(res), dims, metric, nlist, coarse_quantizer, coarse_f16, indices_options_) {
    code_bytes_ = (size_t)dpad_ * 4;
    // lists start on multiples of 32 rows: the fp16 shadow of the filter sweeps is kept in 32-row operand-major blocks
    granule_ = 32;
    // |y|^2 per stored row: second term of the list-major scans' L2 distances / estimates; the filter's error band needs
    // max |y|^2 for the inner product as well
    use_rn_ = true;
    adopt_quantizer_();
}
bool GpuIndexIVFFlat::lmf_capable_() const {
    return ivf_lmf_supported(0, d, dpad_, 0);
}
// fp16 shadow of the arena rows for the f16 sweeps of ivf_lm_filter.hip.  Rebuilt as a whole (one pass over the lists,
// 6 bytes per coordinate) at the first list-major search after a list changed (add / copy_lists / compaction): a
// database that is built once and searched many times pays it once, interleaved add / search workloads pay one arena
// pass per add call.
void GpuIndexIVFFlat::lmf_shadow_room_() const {
    // whole 32-row blocks + padding (the sweeps prefetch the next block they look at: up to 8 blocks behind the last list)
    const size_t need = ((size_t)arena_cap_rows_ / 32 + 10) * (size_t)(ivf_lmf_row_halfs(d) / 16) * 1024;
    if (need > arena_h_.cap) arena_h_.ensure(need, shadow_dirty_ ? 0 : arena_h_.cap, res_->stream);
}
// add(): the blocks that hold new rows, and the relocated lists (GpuIndexIVF::add_core_).  max |y|^2 and the fp16-range flag
// of the rows written merge into those of the copy.
void GpuIndexIVFFlat::lmf_patch_(const uint32_t* d_first_row) {
    const GpuResources& R = *res_;
    const int dh = ivf_lmf_row_halfs(d);
    lmf_shadow_room_();
    lm_scalar_.ensure(64);
    HIP_CHECK(hipMemsetAsync(lm_scalar_.p, 0, 4, R.stream));
    launch_ivf_lmf_shadow(arena_.as<float>(), dpad_, arena_rn_.as<float>(), d, nlist, d_list_len_.as<uint32_t>(),
                          d_list_start_.as<int64_t>(), arena_h_.p, dh, lm_scalar_.as<unsigned>(), d_first_row, R.stream);
    unsigned bits = 0;
    HIP_CHECK(hipMemcpyAsync(&bits, lm_scalar_.p, 4, hipMemcpyDeviceToHost, R.stream));
    R.sync();
    if (bits == 0x7f800000u) shadow_in_range_ = false;
    float mx;
    memcpy(&mx, &bits, 4);
    if (shadow_in_range_) shadow_yn_max_ = std::max(shadow_yn_max_, mx);
}
bool GpuIndexIVFFlat::lmf_prepare_(IvfLmParams& p) const {
    const int dh = ivf_lmf_row_halfs(d);
    lmf_shadow_room_();
    if (shadow_dirty_) {
        const GpuResources& R = *res_;
        lm_scalar_.ensure(64);
        HIP_CHECK(hipMemsetAsync(lm_scalar_.p, 0, 4, R.stream));
        launch_ivf_lmf_shadow(arena_.as<float>(), dpad_, arena_rn_.as<float>(), d, nlist, d_list_len_.as<uint32_t>(),
                              d_list_start_.as<int64_t>(), arena_h_.p, dh, lm_scalar_.as<unsigned>(), nullptr, R.stream);
        unsigned bits = 0;
        HIP_CHECK(hipMemcpyAsync(&bits, lm_scalar_.p, 4, hipMemcpyDeviceToHost, R.stream));
        R.sync();
        shadow_in_range_ = bits != 0x7f800000u;
        memcpy(&shadow_yn_max_, &bits, 4);
        shadow_dirty_ = false;
    }
    if (!shadow_in_range_) return false;
    p.filter = 1;
    p.arena_h = arena_h_.p;
    p.ldh = dh;
    p.yn_max = shadow_yn_max_;
    return true;
}
void GpuIndexIVFFlat::append_(int n, const float* x_pad, const int64_t*, const int64_t* d_dest) {
    launch_ivfflat_append(x_pad, dpad_, n, d, d_dest, arena_.as<float>(), dpad_, dpad_, res_->stream);
    // |y|^2 of the new rows (the sequential chain of the flat index's norms): the list-major scan's second term
    if (use_rn_) launch_l2_norms_scatter(x_pad, dpad_, n, d, d_dest, arena_rn_.as<float>(), res_->stream);
}
void GpuIndexIVFFlat::lists_changed_() {
    // bulk load: norms of every arena row (rows of the slack hold whatever they hold; no scan reads them)
    if (!use_rn_ || arena_rows_ == 0) return;
    launch_l2_norms(arena_.as<float>(), dpad_, arena_rows_, d, arena_rn_.as<float>(), res_->stream);
    res_->sync();
}
bool GpuIndexIVFFlat::lm_capable_() const {
    return ivf_lm_supported(0, dpad_, 0, d);
}
void GpuIndexIVFFlat::fill_lm_(IvfLmParams& p) const {
    p.kind = 0;
    p.arena_vecs = arena_.as<float>();
    p.ldv = dpad_;
    p.arena_rn = arena_rn_.as<float>();
}
void GpuIndexIVFFlat::fill_fused_(IvfFusedParams& p) const {
    p.arena_vecs = arena_.as<float>();
    p.ldv = dpad_;
}
void GpuIndexIVFFlat::reconstruct_n(idx_t i0, idx_t ni, float* recons) const {
    FA_THROW_IF_NOT_MSG(i0 >= 0 && ni >= 0, "negative range");
    if (ni == 0) return;
    FA_THROW_IF_NOT_MSG(recons, "null output");
    // range check of the reference (faiss/gpu/GpuIndexIVFFlat.cu:378-387)
    FA_THROW_IF_NOT_MSG(i0 < ntotal && i0 + ni - 1 < ntotal, "reconstruct_n: id range out of bounds (ntotal)");
    std::lock_guard<std::mutex> g(mu_);
    res_->set_device();
    const GpuResources& R = *res_;
    // rows of the ids in [i0, i0 + ni) wherever their lists hold them; an id of the range that no list holds (a NaN
    // vector that add() counted but skipped, a user id never added) comes back as a NaN row -- like the -1 keys of
    // GpuIndexFlat::reconstruct_batch -- so that a caller can tell it from a stored zero vector
    DevBuf out;
    out.ensure((size_t)ni * d * 4);
    HIP_CHECK(hipMemsetAsync(out.p, 0xff, (size_t)ni * d * 4, R.stream)); // 0xffffffff = a quiet NaN
    launch_ivfflat_rows_by_id(arena_.as<float>(), dpad_, arena_ids_.as<int64_t>(), d_list_start_.as<int64_t>(),
                              d_list_len_.as<uint32_t>(), nlist, d, i0, ni, out.as<float>(), R.stream);
    copy_out(R, recons, out.p, (size_t)ni * d * 4);
    R.sync();
}
void GpuIndexIVFFlat::scan_(int nq, const float* xq_pad, int, const int64_t*) const {
    IvfScanParams p{};
    p.metric = metric_type;
    p.nq = nq;
    p.nprobe = nprobe_eff_;
    p.d = d;
    p.dpad = dpad_;
    p.xq = xq_pad;
    p.ldq = dpad_;
    p.coarse_ids = c_ids_.as<idx_t>();
    p.coarse_dis = c_dis_.as<float>();
    p.list_len = d_list_len_.as<uint32_t>();
    p.list_start = d_list_start_.as<int64_t>();
    p.prefix = prefix_.as<uint32_t>();
    p.q_off = q_off_.as<int64_t>();
    p.keys = keys_.as<unsigned long long>();
    p.arena_vecs = arena_.as<float>();
    p.ldv = dpad_;
    p.sel_mask = cur_sel_mask_;
    SpanGuard sg(res_.get(), "ivfflat_scan_kernel");
    launch_ivfflat_scan(p, res_->stream);
}

// ---------------------------------------------------------------------- IVFPQ
GpuIndexIVFPQ::GpuIndexIVFPQ(std::shared_ptr<GpuResources> res, int dims, int nlist, int M_, int nbits_,
                             int metric, GpuIndexFlat* coarse_quantizer, bool coarse_f16, int indices_options_)
        : GpuIndexIVF(std::move(res), dims, metric, nlist, coarse_quantizer, coarse_f16, indices_options_), M(M_), nbits(nbits_) {
    // same restrictions as the reference GPU index (faiss/gpu/GpuIndexIVFPQ.cu:574-617), minus its
    // shared-memory limit on M: 160 KB of LDS holds an fp32 table up to M = 128.
    FA_THROW_IF_NOT_MSG(nbits == 8, "only 8 bits per code are supported");
    FA_THROW_IF_NOT_MSG(M > 0 && dims % M == 0, "d must be a multiple of M");
    dsub = dims / M;
    FA_THROW_IF_NOT_MSG(ivfpq_scan_lds_bytes(M, dpad_) <= 160 * 1024, "M too large for the LDS lookup table");
    code_bytes_ = (size_t)M;
    granule_ = kPqBlockRows;
    use_t2_ = metric == METRIC_L2;
    use_rn_ = metric == METRIC_L2;
    FA_THROW_IF_NOT_MSG(M % 4 == 0, "M must be a multiple of 4");
    res_->set_device();
    zero_row_.ensure((size_t)dpad_ * 4);
    HIP_CHECK(hipMemset(zero_row_.p, 0, (size_t)dpad_ * 4));
    adopt_quantizer_();
}
void GpuIndexIVFPQ::set_pq_centroids(const float* pq) {
    FA_THROW_IF_NOT_MSG(pq, "null codebook");
    std::lock_guard<std::mutex> g(mu_);
    res_->set_device();
    const size_t bytes = (size_t)M * 256 * dsub * 4;
    DevBuf npq;
    npq.ensure(bytes);
    HIP_CHECK(hipMemcpy(npq.p, pq, bytes, hipMemcpyDefault));
    pq_t_.ensure(bytes);
    launch_pq_transpose(npq.as<float>(), M, dsub, pq_t_.as<float>(), res_->stream);
    res_->sync();
    std::swap(pq_.p, npq.p);
    std::swap(pq_.cap, npq.cap);
    lmf_quant_dirty_ = true;
    // the decoded-residual copy of the lists (lmf_decoded_()) holds fp16 values DECODED WITH pq_: a new codebook makes it stale
    // (the codebook mode's copy holds raw code bytes and stays valid)
    if (lmf_decoded_()) shadow_dirty_ = true;
    update_is_trained_();
    if (nstored_ > 0 && is_trained) lists_changed_();
}
std::vector<float> GpuIndexIVFPQ::get_pq_centroids() const {
    res_->set_device();
    std::vector<float> out((size_t)M * 256 * dsub);
    FA_THROW_IF_NOT_MSG(pq_.p, "PQ not trained");
    HIP_CHECK(hipMemcpy(out.data(), pq_.p, out.size() * 4, hipMemcpyDeviceToHost));
    return out;
}
void GpuIndexIVFPQ::train_residual_(idx_t n, const float* x_dev_pad) {
    // reference: GpuIndexIVFPQ::trainResidualQuantizer_ (faiss/gpu/GpuIndexIVFPQ.cu:287-340):
    // residuals of the training set w.r.t. their nearest centroid, then one k-means (256
    // centroids, dsub dims) per sub-quantizer with a GPU flat index as assignment engine
    // (faiss/impl/ProductQuantizer.cpp:130-207).
    const GpuResources& R = *res_;
    idx_t nt = std::min<idx_t>(n, (idx_t)256 * 256); // max_points_per_centroid * ksub
    DevBuf dlab, ddis, dres, dsel, dsub_rows;
    const float* xs = x_dev_pad;
    if (nt < n) {
        // a random subset of the training set (seeded), as ProductQuantizer::train / Clustering subsample
        // (faiss/Clustering.cpp subsample_training_set): the first rows of an ordered training set would bias the codebook
        std::mt19937_64 rng((uint64_t)cp_seed + 0x9e3779b97f4a7c15ull);
        std::vector<uint32_t> perm((size_t)n);
        std::iota(perm.begin(), perm.end(), 0u);
        for (idx_t i = 0; i < nt; i++) {
            const idx_t j = i + (idx_t)(rng() % (uint64_t)(n - i));
            std::swap(perm[i], perm[j]);
        }
        dsel.ensure((size_t)nt * 4);
        HIP_CHECK(hipMemcpyAsync(dsel.p, perm.data(), (size_t)nt * 4, hipMemcpyHostToDevice, R.stream));
        dsub_rows.ensure((size_t)nt * dpad_ * 4);
        launch_gather_rows(x_dev_pad, dpad_, dpad_, dsel.as<uint32_t>(), (int)nt, dsub_rows.as<float>(), R.stream);
        R.sync();
        xs = dsub_rows.as<float>();
    }
    dlab.ensure((size_t)nt * 8);
    ddis.ensure((size_t)nt * 4);
    dres.ensure((size_t)nt * d * 4);
    quantizer->search_device((int)nt, xs, 1, ddis.as<float>(), dlab.as<idx_t>());
    launch_residual(xs, dpad_, nt, d, dlab.as<idx_t>(), centroids_dev_(), dpad_,
                    dres.as<float>(), d, R.stream);
    R.sync();
    std::vector<float> pq((size_t)M * 256 * dsub);
    const char* loop_env = experiment_env("FAISS_AMD_PQ_TRAIN_LOOP");
    if (pq_train_batched && !(loop_env && atoi(loop_env) == 1) && pq_train_batched_supported(dsub) && nt >= 256) {
        train_pq_batched_(nt, dres.as<float>(), pq);
    } else {
        // one k-means per sub-quantizer on its dsub columns of the device-resident residuals (row stride d)
        GpuIndexFlat assign_index(res_, dsub, METRIC_L2);
        for (int m = 0; m < M; m++) {
            Clustering clus(dsub, 256);
            clus.niter = pq_niter;
            clus.seed = cp_seed + m;
            clus.train(nt, dres.as<float>() + (size_t)m * dsub, assign_index, d);
            memcpy(&pq[(size_t)m * 256 * dsub], clus.centroids.data(), sizeof(float) * 256 * dsub);
        }
    }
    pq_.ensure(pq.size() * 4);
    HIP_CHECK(hipMemcpy(pq_.p, pq.data(), pq.size() * 4, hipMemcpyHostToDevice));
    pq_t_.ensure(pq.size() * 4);
    launch_pq_transpose(pq_.as<float>(), M, dsub, pq_t_.as<float>(), R.stream);
    R.sync();
}
// All M sub-quantizers as ONE k-means over M * nt points and M * 256 clusters (kernels: ivf_kernels.hip pq_train_*): what
// Clustering::train_device_ does per sub-space -- same initial draw per sub-space (seed cp_seed + m), same assignment and
// update arithmetic, same refill of empty clusters with the sub-space's own generator -- with one set of launches and one
// host synchronisation per iteration instead of M.  res: [nt][d] residuals on the device.
void GpuIndexIVFPQ::train_pq_batched_(idx_t nt, const float* res, std::vector<float>& pq) {
    const GpuResources& R = *res_;
    const int K = 256, KT = M * K;
    const int64_t N = (int64_t)M * nt;
    FA_THROW_IF_NOT_MSG(N < ((int64_t)1 << 32), "product quantizer training: too many points");
    // ---- init: k distinct random points per sub-space (Clustering::train_device_: the first k of a partial shuffle)
    std::vector<std::mt19937_64> rng;
    std::vector<uint32_t> hsel((size_t)KT);
    {
        std::vector<uint32_t> perm((size_t)nt);
        for (int m = 0; m < M; m++) {
            rng.emplace_back((uint64_t)(cp_seed + m));
            std::iota(perm.begin(), perm.end(), 0u);
            for (int i = 0; i < K; i++) {
                const idx_t j = i + (idx_t)(rng[m]() % (uint64_t)(nt - i));
                std::swap(perm[i], perm[j]);
            }
            memcpy(&hsel[(size_t)m * K], perm.data(), sizeof(uint32_t) * K);
        }
    }
    DevBuf sel, cen, lab, dest, order, hist, cnt, zero, start;
    sel.ensure((size_t)KT * 4);
    cen.ensure((size_t)KT * dsub * 4);
    HIP_CHECK(hipMemcpyAsync(sel.p, hsel.data(), (size_t)KT * 4, hipMemcpyHostToDevice, R.stream));
    launch_pq_train_init(res, d, M, dsub, sel.as<uint32_t>(), cen.as<float>(), R.stream);
    int chunk = 256;
    while (div_up((size_t)N, (size_t)chunk) > 512) chunk *= 2;
    const int nchunks = (int)div_up((size_t)N, (size_t)chunk);
    lab.ensure((size_t)N * 8);
    dest.ensure((size_t)N * 8);
    order.ensure((size_t)N * 4);
    hist.ensure((size_t)nchunks * KT * 4);
    cnt.ensure((size_t)KT * 4);
    zero.ensure((size_t)KT * 4);
    start.ensure((size_t)(KT + 1) * 8);
    HIP_CHECK(hipMemsetAsync(zero.p, 0, (size_t)KT * 4, R.stream));
    std::vector<uint32_t> hcnt((size_t)KT);
    std::vector<idx_t> hassign(K);
    Clustering split(dsub, K);
    for (int it = 0; it < pq_niter; it++) {
        check_interrupt();
        launch_pq_train_assign(res, d, nt, M, dsub, cen.as<float>(), lab.as<int64_t>(), R.stream);
        HIP_CHECK(hipMemsetAsync(hist.p, 0, (size_t)nchunks * KT * 4, R.stream));
        launch_ivf_histogram(lab.as<int64_t>(), N, KT, chunk, hist.as<uint32_t>(), R.stream);
        launch_ivf_chunk_scan(hist.as<uint32_t>(), nchunks, KT, zero.as<uint32_t>(), cnt.as<uint32_t>(), R.stream);
        HIP_CHECK(hipMemcpyAsync(hcnt.data(), cnt.p, (size_t)KT * 4, hipMemcpyDeviceToHost, R.stream));
        launch_exclusive_scan(cnt.as<uint32_t>(), KT, start.as<int64_t>(), R.stream);
        launch_ivf_rank(lab.as<int64_t>(), N, KT, chunk, hist.as<uint32_t>(), start.as<int64_t>(), dest.as<int64_t>(), R.stream);
        launch_invert_dest(dest.as<int64_t>(), N, order.as<uint32_t>(), R.stream);
        launch_pq_train_update(res, d, nt, M, dsub, order.as<uint32_t>(), start.as<int64_t>(), cnt.as<uint32_t>(), cen.as<float>(),
                               R.stream);
        R.sync();
        // empty clusters (rare): the sub-space's centroids come to the host, are refilled as Clustering does, go back
        for (int m = 0; m < M; m++) {
            bool any_empty = false;
            for (int c = 0; c < K; c++) {
                hassign[c] = hcnt[(size_t)m * K + c];
                any_empty |= hassign[c] == 0;
            }
            if (!any_empty) continue;
            split.centroids.resize((size_t)K * dsub);
            float* dm = cen.as<float>() + (size_t)m * K * dsub;
            HIP_CHECK(hipMemcpy(split.centroids.data(), dm, (size_t)K * dsub * 4, hipMemcpyDeviceToHost));
            split.split_empty_clusters(rng[m], nt, hassign);
            HIP_CHECK(hipMemcpy(dm, split.centroids.data(), (size_t)K * dsub * 4, hipMemcpyHostToDevice));
        }
    }
    HIP_CHECK(hipMemcpy(pq.data(), cen.p, (size_t)KT * dsub * 4, hipMemcpyDeviceToHost));
}
void GpuIndexIVFPQ::append_(int n, const float* x_pad, const int64_t* d_labels, const int64_t* d_dest) {
    launch_ivfpq_encode_append(x_pad, dpad_, n, d, d_labels, d_dest, centroids_dev_(), dpad_, M,
                               dsub, pq_.as<float>(), arena_.as<uint8_t>(), res_->stream);
    if (use_t2_)
        launch_ivfpq_t2_rows(arena_.as<uint8_t>(), d_labels, d_dest, n, centroids_dev_(), dpad_, M, dsub,
                             pq_.as<float>(), arena_t2_.as<float>(), res_->stream);
    // |r^|^2 of the new rows = the same chain against a zero centroid (fmaf(2, 0, r) == r): the list-major scan's term
    if (use_rn_)
        launch_ivfpq_t2_rows(arena_.as<uint8_t>(), d_labels, d_dest, n, zero_row_.as<float>(), 0, M, dsub, pq_.as<float>(),
                             arena_rn_.as<float>(), res_->stream);
}
void GpuIndexIVFPQ::lists_changed_() {
    // recompute the per-vector L2 terms for every stored row (bulk load, or the quantizers were replaced)
    if (nstored_ == 0) return;
    if (use_t2_)
        launch_ivfpq_t2_lists(arena_.as<uint8_t>(), d_list_start_.as<int64_t>(), d_list_len_.as<uint32_t>(), nlist,
                              centroids_dev_(), dpad_, M, dsub, pq_.as<float>(), arena_t2_.as<float>(), res_->stream);
    if (use_rn_)
        launch_ivfpq_t2_lists(arena_.as<uint8_t>(), d_list_start_.as<int64_t>(), d_list_len_.as<uint32_t>(), nlist,
                              zero_row_.as<float>(), 0, M, dsub, pq_.as<float>(), arena_rn_.as<float>(), res_->stream);
    res_->sync();
}
bool GpuIndexIVFPQ::ivf_lm_pq_lds_supported_() const {
    return ivf_lm_pq_lds_supported(d, dpad_, M);
}
bool GpuIndexIVFPQ::lm_capable_() const {
    return ivf_lm_supported(1, dpad_, M, d);
}
// the LDS-codebook sweeps (d <= 128, d a multiple of 16, dsub 1 / 2 / 4 / 8 k) ...
bool GpuIndexIVFPQ::lmf_codebook_capable_() const {
    return ivf_lmf_supported(1, d, dpad_, M) && (size_t)d * 512 + 8 * 256 * 12 <= 160 * 1024;
}
// ... and every other shape up to d = 512 through an fp16 copy of the DECODED residuals (round 5: 2 d bytes per row instead of
// M; the IVFFlat sweeps with per-pair operands, ivf_lm_filter.hip lmf_pq_decode_kernel)
bool GpuIndexIVFPQ::lmf_decoded_() const {
    return !lmf_codebook_capable_() && ivf_lmf_pq_decoded_supported(d, dpad_, M);
}
bool GpuIndexIVFPQ::lmf_capable_() const {
    return lmf_codebook_capable_() || lmf_decoded_();
}
// fp16 codebook + the norm bounds of the filter's error band: upper bound of |r^|^2 (sum over the sub-quantizers of their
// largest squared entry norm), max |centroid|^2.  Rebuilt when a quantizer changed.
bool GpuIndexIVFPQ::lmf_two_copies_() const {
    return lmf_two_copies && ivf_lmf_choice_shape(d, M);
}
void GpuIndexIVFPQ::lmf_shadow_room_() const {
    size_t blk, pad = 4;
    if (lmf_decoded_()) {
        blk = (size_t)(ivf_lmf_row_halfs(d) / 16) * 1024; // (the IVFFlat shadow's blocks; its sweeps prefetch up to 8 blocks ahead)
        pad = 10;
    } else {
        int bpl = 0, piece = 0;
        ivf_lmf_code_shadow_shape(d, M, &bpl, &piece);
        blk = lmf_two_copies_() ? (size_t)64 * 40 : (size_t)64 * ((bpl + piece - 1) / piece) * piece;
    }
    const size_t need = ((size_t)arena_cap_rows_ / 32 + pad) * blk;
    if (need > arena_cs_.cap) arena_cs_.ensure(need, shadow_dirty_ ? 0 : arena_cs_.cap, res_->stream);
}
void GpuIndexIVFPQ::lmf_write_copy_(const uint32_t* d_first_row) const {
    if (lmf_decoded_())
        launch_ivf_lmf_pq_decode(arena_.as<uint8_t>(), pq_.as<float>(), d, M, nlist, d_list_len_.as<uint32_t>(),
                                 d_list_start_.as<int64_t>(), arena_cs_.p, ivf_lmf_row_halfs(d), d_first_row, res_->stream);
    else if (lmf_two_copies_())
        launch_ivf_lmf_code_choice(arena_.as<uint8_t>(), nlist, d_list_len_.as<uint32_t>(), d_list_start_.as<int64_t>(),
                                   arena_cs_.as<uint8_t>(), d_first_row, res_->stream);
    else
        launch_ivf_lmf_code_shadow(arena_.as<uint8_t>(), d, M, nlist, d_list_len_.as<uint32_t>(), d_list_start_.as<int64_t>(),
                                   arena_cs_.as<uint8_t>(), d_first_row, res_->stream);
}
void GpuIndexIVFPQ::lmf_patch_(const uint32_t* d_first_row) {
    lmf_shadow_room_();
    lmf_write_copy_(d_first_row);
}
bool GpuIndexIVFPQ::lmf_prepare_(IvfLmParams& p) const {
    if (lmf_quant_dirty_) {
        FA_THROW_IF_NOT_MSG(pq_.p && quantizer->ntotal == nlist, "index not trained");
        std::vector<float> cb((size_t)M * 256 * dsub);
        HIP_CHECK(hipMemcpy(cb.data(), pq_.p, cb.size() * 4, hipMemcpyDeviceToHost));
        std::vector<_Float16> h(cb.size());
        bool ok = true;
        double yn = 0.0;
        for (int m = 0; m < M; m++) {
            double mx = 0.0;
            for (int c = 0; c < 256; c++) {
                double nn = 0.0;
                for (int jd = 0; jd < dsub; jd++) {
                    const float v = cb[((size_t)m * 256 + c) * dsub + jd];
                    if (!(std::fabs(v) <= 65000.f)) ok = false;
                    h[((size_t)m * 256 + c) * dsub + jd] = (_Float16)v;
                    nn += (double)v * (double)v;
                }
                mx = std::max(mx, nn);
            }
            yn += mx;
        }
        std::vector<float> cen((size_t)nlist * dpad_);
        HIP_CHECK(hipMemcpy(cen.data(), centroids_dev_(), cen.size() * 4, hipMemcpyDeviceToHost));
        double cn = 0.0;
        for (int l = 0; l < nlist; l++) {
            double nn = 0.0;
            for (int jd = 0; jd < d; jd++) {
                const float v = cen[(size_t)l * dpad_ + jd];
                if (!(std::fabs(v) <= 30000.f)) ok = false;
                nn += (double)v * (double)v;
            }
            cn = std::max(cn, nn);
        }
        pq16_.ensure(h.size() * 2);
        HIP_CHECK(hipMemcpy(pq16_.p, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        pq16_in_range_ = ok && yn < 1e30 && cn < 1e30;
        pq_yn_max_ = (float)(yn * 1.0001);
        cn_max_ = (float)(cn * 1.0001);
        lmf_quant_dirty_ = false;
    }
    if (!pq16_in_range_) return false;
    int bpl = 0, piece = 0;
    if (!lmf_decoded_()) ivf_lmf_code_shadow_shape(d, M, &bpl, &piece);
    lmf_shadow_room_();
    if (shadow_dirty_) {
        // operand-major copy of the codes (kernels.h IvfLmParams::arena_cs): built as a whole at the first list-major search
        // that finds none (add() keeps a live one up to date, lmf_patch_)
        lmf_write_copy_(nullptr);
        res_->sync();
        shadow_dirty_ = false;
    }
    if (lmf_decoded_()) {
        // the sweeps of the scalar quantizer (pair operands) over the decoded residuals: scale 1, offset 0
        if (!sq_one_.p) {
            const int dsq = (int)round_up(d, 16);
            std::vector<float> one((size_t)dsq, 0.f);
            for (int i = 0; i < d; i++) one[i] = 1.f;
            sq_one_.ensure((size_t)dsq * 4);
            sq_nil_.ensure((size_t)dsq * 4);
            HIP_CHECK(hipMemcpy(sq_one_.p, one.data(), (size_t)dsq * 4, hipMemcpyHostToDevice));
            HIP_CHECK(hipMemset(sq_nil_.p, 0, (size_t)dsq * 4));
        }
        p.lmf_pairb = 1;
        p.arena_h = arena_cs_.p;
        p.ldh = ivf_lmf_row_halfs(d);
        p.sq_s = sq_one_.as<float>();
        p.sq_b = sq_nil_.as<float>();
        p.sq_by_residual = 1; // (L2: a = q - centroid; inner product: the coarse term joins the query term)
    } else {
        p.arena_cs = arena_cs_.as<uint8_t>();
        p.cs_bpl = bpl;
        p.cs_piece = piece;
        p.cs_choice = lmf_two_copies_() ? 1 : 0;
        p.lmf_fast_gather = lmf_fast_gather ? 1 : 0;
    }
    p.filter = 1;
    p.pq_t = pq_t_.as<float>();
    p.pq16 = pq16_.p;
    p.yn_max = pq_yn_max_;
    p.cn_max = cn_max_;
    p.arena_t2 = arena_t2_.as<float>();
    return true;
}
void GpuIndexIVFPQ::fill_lm_(IvfLmParams& p) const {
    p.kind = 1;
    p.arena_codes = arena_.as<uint8_t>();
    p.arena_rn = arena_rn_.as<float>();
    p.M = M;
    p.dsub = dsub;
    p.pq_centroids = pq_.as<float>();
    p.centroids = centroids_dev_();
    p.ldc = dpad_;
}
void GpuIndexIVFPQ::fill_fused_(IvfFusedParams& p) const {
    p.arena_t2 = arena_t2_.as<float>();
    p.centroids = centroids_dev_();
    p.ldc = dpad_;
    p.M = M;
    p.dsub = dsub;
    p.pq_centroids = pq_.as<float>();
    p.pq_t = pq_t_.as<float>();
    p.arena_codes = arena_.as<uint8_t>();
}
void GpuIndexIVFPQ::scan_(int nq, const float* xq_pad, int, const int64_t*) const {
    IvfScanParams p{};
    p.metric = metric_type;
    p.nq = nq;
    p.nprobe = nprobe_eff_;
    p.d = d;
    p.dpad = dpad_;
    p.xq = xq_pad;
    p.ldq = dpad_;
    p.coarse_ids = c_ids_.as<idx_t>();
    p.coarse_dis = c_dis_.as<float>();
    p.list_len = d_list_len_.as<uint32_t>();
    p.list_start = d_list_start_.as<int64_t>();
    p.prefix = prefix_.as<uint32_t>();
    p.q_off = q_off_.as<int64_t>();
    p.keys = keys_.as<unsigned long long>();
    p.centroids = centroids_dev_();
    p.ldc = dpad_;
    p.M = M;
    p.dsub = dsub;
    p.pq_centroids = pq_.as<float>();
    p.pq_t = pq_t_.as<float>();
    p.arena_codes = arena_.as<uint8_t>();
    p.arena_t2 = arena_t2_.as<float>();
    p.sel_mask = cur_sel_mask_;
    SpanGuard sg(res_.get(), "ivfpq_scan_kernel");
    launch_ivfpq_scan(p, res_->stream);
}

// ====================================================================== shards
void merge_knn_results(int metric, idx_t nq, idx_t k, int nshard, const float* all_d, const idx_t* all_i,
                       const idx_t* base, float* D, idx_t* I) {
    // k-way merge of sorted lists; ties on distance go to the smaller label, so the merged
    // result equals an unsharded search (reference merge: faiss/utils/Heap.cpp:166-240)
    const bool l2 = metric == METRIC_L2;
    std::vector<idx_t> ptr(nshard);
    for (idx_t q = 0; q < nq; q++) {
        std::fill(ptr.begin(), ptr.end(), 0);
        for (idx_t j = 0; j < k; j++) {
            int best = -1;
            float bd = 0;
            idx_t bi = 0;
            for (int s = 0; s < nshard; s++) {
                if (ptr[s] >= k) continue;
                size_t off = ((size_t)s * nq + q) * k + ptr[s];
                idx_t id = all_i[off];
                if (id < 0) continue; // exhausted shard (padding)
                float dv = all_d[off];
                idx_t gid = id + (base ? base[s] : 0);
                bool better = best < 0 || (l2 ? dv < bd : dv > bd) || (dv == bd && gid < bi);
                if (better) {
                    best = s;
                    bd = dv;
                    bi = gid;
                }
            }
            if (best < 0) {
                D[q * k + j] = neutral_distance(metric);
                I[q * k + j] = -1;
            } else {
                D[q * k + j] = bd;
                I[q * k + j] = bi;
                ptr[best]++;
            }
        }
    }
}

void merge_knn_results_device(GpuResources& R, int metric, int nq, int k, int nshard, const float* all_d,
                              const idx_t* all_i, const idx_t* base_host, float* D, idx_t* I) {
    FA_THROW_IF_NOT_MSG(k >= 1 && k <= kMaxSelectionK && nshard >= 1, "bad arguments");
    FA_THROW_IF_NOT_MSG((int64_t)nshard * k < ((int64_t)1 << 31), "too many candidates");
    if (nq == 0) return;
    R.set_device();
    static thread_local DevBuf keys, cnt, dbase; // per host thread == per device in our usage
    keys.ensure((size_t)nq * nshard * k * 8);
    cnt.ensure((size_t)nq * 4);
    const idx_t* dbase_p = nullptr;
    if (base_host) {
        dbase.ensure((size_t)nshard * 8);
        HIP_CHECK(hipMemcpyAsync(dbase.p, base_host, (size_t)nshard * 8, hipMemcpyHostToDevice, R.stream));
        dbase_p = dbase.as<idx_t>();
    }
    {
        SpanGuard sg(&R, "pack_merge_keys_kernel");
        launch_pack_merge_keys(metric, all_d, all_i, nshard, nq, k, keys.as<unsigned long long>(),
                               cnt.as<uint32_t>(), R.stream);
    }
    SelectParams sp{};
    sp.metric = metric;
    sp.nq = nq;
    sp.k = k;
    sp.keys = keys.as<unsigned long long>();
    sp.q_stride = (int64_t)nshard * k;
    sp.nseg = 1;
    sp.seg_cnt = cnt.as<uint32_t>();
    sp.mode = 2;
    sp.merge_ids = all_i;
    sp.merge_base = dbase_p;
    sp.out_dis = D;
    sp.out_ids = I;
    {
        SpanGuard sg(&R, "select_k_kernel");
        launch_select_k(sp, R.stream);
    }
    R.sync();
}

IndexShards::IndexShards(int d_, bool threaded_, bool successive_ids_)
        : Index(d_, METRIC_L2), threaded(threaded_), successive_ids(successive_ids_) {}
IndexShards::~IndexShards() {
    if (own_indices)
        for (auto* s : shards_) delete s;
}
void IndexShards::sync_() {
    // reference: IndexShardsTemplate::syncWithSubIndexes (faiss/IndexShards.cpp:87-110)
    if (shards_.empty()) {
        ntotal = 0;
        is_trained = false;
        return;
    }
    metric_type = shards_[0]->metric_type;
    is_trained = shards_[0]->is_trained;
    ntotal = 0;
    for (auto* s : shards_) {
        FA_THROW_IF_NOT_MSG(s->d == d, "shard dimension mismatch");
        FA_THROW_IF_NOT_MSG(s->metric_type == metric_type, "shard metric mismatch");
        FA_THROW_IF_NOT_MSG(s->is_trained == is_trained, "shard training state mismatch");
        ntotal += s->ntotal;
    }
}
void IndexShards::add_shard(Index* idx) {
    shards_.push_back(idx);
    sync_();
}
template <typename F>
static void run_on_shards(const std::vector<Index*>& shards, bool threaded, F f) {
    // one host thread per shard, exceptions gathered (reference: ThreadedIndex-inl.h:120-193)
    std::vector<std::string> errors(shards.size());
    auto body = [&](int s) {
        try {
            f(s, shards[s]);
        } catch (std::exception& e) {
            errors[s] = e.what();
            if (errors[s].empty()) errors[s] = "unknown error";
        }
    };
    if (threaded && shards.size() > 1) {
        std::vector<std::thread> th;
        for (int s = 0; s < (int)shards.size(); s++) th.emplace_back(body, s);
        for (auto& t : th) t.join();
    } else {
        for (int s = 0; s < (int)shards.size(); s++) body(s);
    }
    std::string all;
    for (size_t s = 0; s < errors.size(); s++)
        if (!errors[s].empty()) all += "shard " + std::to_string(s) + ": " + errors[s] + "; ";
    if (!all.empty()) FA_THROW_MSG(all);
}
void IndexShards::train(idx_t n, const float* x) {
    run_on_shards(shards_, threaded, [&](int, Index* s) { s->train(n, x); });
    sync_();
}
void IndexShards::add(idx_t n, const float* x) {
    add_with_ids(n, x, nullptr);
}
void IndexShards::add_with_ids(idx_t n, const float* x, const idx_t* xids) {
    // reference: faiss/IndexShards.cpp:135-194
    FA_THROW_IF_NOT_MSG(!(successive_ids && xids), "It makes no sense to pass in ids and request them to be shifted");
    FA_THROW_IF_NOT_MSG(!shards_.empty(), "no shards");
    if (successive_ids && xids == nullptr) {
        // ok
    }
    const idx_t nshard = (idx_t)shards_.size();
    std::vector<idx_t> aids;
    const idx_t* ids = xids;
    if (!ids && !successive_ids) {
        aids.resize(n);
        for (idx_t i = 0; i < n; i++) aids[i] = ntotal + i;
        ids = aids.data();
    }
    run_on_shards(shards_, threaded, [&](int no, Index* s) {
        idx_t i0 = (idx_t)no * n / nshard;
        idx_t i1 = ((idx_t)no + 1) * n / nshard;
        const float* x0 = x + (size_t)i0 * d;
        if (i1 == i0) return;
        if (ids) s->add_with_ids(i1 - i0, x0, ids + i0);
        else s->add(i1 - i0, x0);
    });
    sync_();
}
void IndexShards::reset() {
    run_on_shards(shards_, threaded, [&](int, Index* s) { s->reset(); });
    sync_();
}
void IndexShards::search(idx_t n, const float* x, idx_t k, float* distances, idx_t* labels,
                         const SearchParameters* params) const {
    // reference: faiss/IndexShards.cpp:196-265
    FA_THROW_IF_NOT_MSG(k > 0, "k must be positive");
    const int nshard = (int)shards_.size();
    FA_THROW_IF_NOT_MSG(nshard > 0, "no shards");
    std::vector<float> all_d((size_t)nshard * n * k);
    std::vector<idx_t> all_i((size_t)nshard * n * k);
    std::vector<idx_t> base(nshard, 0);
    if (successive_ids) {
        idx_t t = 0;
        for (int s = 0; s < nshard; s++) {
            base[s] = t;
            t += shards_[s]->ntotal;
        }
    }
    run_on_shards(shards_, threaded, [&](int no, Index* s) {
        s->search(n, x, k, all_d.data() + (size_t)no * n * k, all_i.data() + (size_t)no * n * k, params);
    });
    merge_knn_results(order_metric(metric_type), n, k, nshard, all_d.data(), all_i.data(),
                      successive_ids ? base.data() : nullptr, distances, labels);
}

// ====================================================================== IndexReplicas
IndexReplicas::IndexReplicas(int d_, bool threaded_) : Index(d_, METRIC_L2), threaded(threaded_) {}
IndexReplicas::~IndexReplicas() {
    if (own_indices)
        for (auto* r : replicas_) delete r;
}
void IndexReplicas::sync_() {
    // reference: IndexReplicasTemplate::syncWithSubIndexes (faiss/IndexReplicas.cpp:177-198)
    if (replicas_.empty()) {
        ntotal = 0;
        is_trained = false;
        return;
    }
    metric_type = replicas_[0]->metric_type;
    is_trained = replicas_[0]->is_trained;
    ntotal = replicas_[0]->ntotal;
    for (auto* r : replicas_) {
        FA_THROW_IF_NOT_MSG(r->d == d, "replica dimension mismatch");
        FA_THROW_IF_NOT_MSG(r->metric_type == metric_type, "replica metric mismatch");
        FA_THROW_IF_NOT_MSG(r->is_trained == is_trained, "replica training state mismatch");
        FA_THROW_IF_NOT_MSG(r->ntotal == ntotal, "replicas hold different numbers of vectors");
    }
}
void IndexReplicas::add_replica(Index* idx) {
    replicas_.push_back(idx);
    sync_();
}
void IndexReplicas::train(idx_t n, const float* x) {
    run_on_shards(replicas_, threaded, [&](int, Index* r) { r->train(n, x); });
    sync_();
}
void IndexReplicas::add(idx_t n, const float* x) {
    run_on_shards(replicas_, threaded, [&](int, Index* r) { r->add(n, x); });
    sync_();
}
void IndexReplicas::add_with_ids(idx_t n, const float* x, const idx_t* xids) {
    run_on_shards(replicas_, threaded, [&](int, Index* r) { r->add_with_ids(n, x, xids); });
    sync_();
}
void IndexReplicas::reset() {
    run_on_shards(replicas_, threaded, [&](int, Index* r) { r->reset(); });
    sync_();
}
void IndexReplicas::reconstruct(idx_t key, float* recons) const {
    FA_THROW_IF_NOT_MSG(!replicas_.empty(), "no replicas in index");
    replicas_[0]->reconstruct(key, recons); // faiss/IndexReplicas.cpp:83-89
}
void IndexReplicas::search(idx_t n, const float* x, idx_t k, float* distances, idx_t* labels,
                           const SearchParameters* params) const {
    // reference: faiss/IndexReplicas.cpp:123-175 (queries dealt out in ceil(n / count) blocks)
    FA_THROW_IF_NOT_MSG(!params, "search params not supported for this index");
    FA_THROW_IF_NOT_MSG(k > 0, "k must be positive");
    const idx_t cnt = (idx_t)replicas_.size();
    FA_THROW_IF_NOT_MSG(cnt > 0, "no replicas in index");
    if (n == 0) return;
    const idx_t per = (n + cnt - 1) / cnt;
    run_on_shards(replicas_, threaded, [&](int i, Index* r) {
        const idx_t base = (idx_t)i * per;
        if (base < n) {
            const idx_t ni = std::min(per, n - base);
            r->search(ni, x + (size_t)base * d, k, distances + (size_t)base * k, labels + (size_t)base * k);
        }
    });
}

} // namespace faiss_amd
