// faiss_amd/csrc/kernels.h -- host-callable launchers of the hand-written gfx950 kernels.
// All pointers are DEVICE pointers; all launches are asynchronous on `stream`.
#pragma once
#include <cmath>
#include "common.h"

namespace faiss_amd {

// ------------------------------------------------------------------ utility kernels
// out[i] = sum_k x[i][k]^2 as a sequential fmaf chain k = 0..d-1 (one rounding per term).
// Replaces faiss/gpu/impl/L2Norm.cu:33-144 (l2NormRowMajor).
void launch_l2_norms(const float* x, int64_t ld, int64_t n, int d, float* out, hipStream_t stream);

// dst[i][0..dpad) = src[i][0..d) followed by zeros.  (Padded private copies let every
// kernel use 16-byte loads and make fmaf(0,0,acc) the exact no-op tail of the dot chain.)
void launch_pad_rows(const float* src, int64_t lds_src, int64_t n, int d, float* dst, int64_t ld_dst,
                     int dpad, hipStream_t stream);

// out[i] = x[i] - rows[keys[i]] (x given) or rows[keys[i]] (x null); key outside [0, nrows) -> NaN row.
// Replaces faiss/gpu/impl/VectorResidual.cu:26-97 (runCalcResidual) and the reconstruct-by-ids gather.
// rows16 (fp16 storage) is used when rows is null.
void launch_rows_by_key(const float* x, int64_t ld_x, const int64_t* keys, int64_t n, int d, const float* rows,
                        int64_t ld_rows, int64_t nrows, float* out, int64_t ld_out, hipStream_t stream,
                        const _Float16* rows16 = nullptr, int64_t ld_rows16 = 0);
// dst [n][d] fp32 row-major <- src [n][d] (row_major) or [d][n] (column major) of type 1 = f32, 2 = f16, 3 = bf16
// (faiss::gpu::DistanceDataType, faiss/gpu/GpuDistance.h:19-23); widening is exact
void launch_convert_matrix(const void* src, int type, bool row_major, int64_t n, int d, float* dst, hipStream_t stream);
void launch_i64_to_i32(const int64_t* src, int64_t n, int32_t* dst, hipStream_t stream);
// x[i] = float(half(x[i])): the values an fp16-storage index holds (queries are rounded the same way, as
// FlatIndex::query converts them, faiss/gpu/impl/FlatIndex.cu:112-135)
void launch_round_f16_inplace(float* x, int64_t n, hipStream_t stream);
// dst[i][0..dpad) = float(src[i][0..dpad)) for fp16 rows of stride ld_src (dpad <= ld_src)
void launch_f16_rows_to_f32(const _Float16* src, int64_t ld_src, int64_t n, int dpad, float* dst, hipStream_t stream);

// ------------------------------------------------------------------ Flat: fused distance + k-selection
constexpr int kFlatQueriesPerBlock = 256; // 8 waves x 32 queries
constexpr int kFlatTileRows = 64;

struct FlatScanParams {
    int metric;         // MetricType
    const float* xq;    // [nq][ldq] padded queries
    const float* xqn;   // [nq] query squared norms (L2 only; may be null for IP)
    const float* xb;    // [nb][ldb] padded database
    const float* xbn;   // [nb] database squared norms (L2 only)
    // IP, optional: additive start value per row, 0 for rows that take part and -inf for rows an IDSelector excludes
    // (the L2 kernels get the same effect from +inf entries in xbn); null = every row takes part
    const float* ip_bias;
    int64_t ldq, ldb;
    int nq;
    int nb;             // < 2^31 per device index
    int dpad;           // multiple of 8
    int nsplit;         // database splits (each scanned by its own workgroups)
    int rows_per_split; // multiple of kFlatTileRows
    int ngroups;        // ceil(nq / 256)
    int k;
    int cap;            // reservoir capacity per (query, split); >= k + 32
    unsigned long long* res_keys; // [nq][nsplit][cap]
    uint32_t* res_cnt;            // [nq][nsplit]
    float* dump;                  // optional [nq][nb] full distance matrix (tests), else null
};
// Fused MFMA (v_mfma_f32_32x32x2_f32) distance + threshold filter + reservoir append.
// Replaces the GEMM -> l2SelectMinK -> sumAlongRows chain of
// faiss/gpu/impl/Distance.cu:120-406 (runDistance) without materialising the distance tile.
void launch_flat_scan(const FlatScanParams& p, hipStream_t stream);
size_t flat_scan_lds_bytes();

// k = 1 over a database small enough for LDS (dpad <= 32, nb * (dpad + 1) * 4 <= 48 KB): distance + argmin in one launch,
// same arithmetic and tie rule as the scan + select pair (k-means assignment of PQ sub-spaces)
bool flat_assign_small_supported(int nb, int dpad);
void launch_flat_assign_small(int metric, const float* xq, int64_t ldq, int nq, const float* xb, const float* xbn, int64_t ldb,
                              int nb, int dpad, float* out_dis, int64_t* out_ids, hipStream_t stream);

// Debug / cross-check path: scalar VALU distances with the identical fmaf chain, every
// distance written as a 64-bit key.  keys: [nq][nb].
void launch_flat_simple(int metric, const float* xq, const float* xqn, int64_t ldq, int nq,
                        const float* xb, const float* xbn, int64_t ldb, int nb, int dpad,
                        unsigned long long* keys, hipStream_t stream);

// The "extra" metrics of the flat index (L1, Linf, Lp, Canberra, BrayCurtis, JensenShannon, Jaccard): one sequential
// fp32 pass over the d dimensions per (query, row), every distance written as a 64-bit key (ordered by order_metric).
// keys: [nq][nb].  Replaces faiss/gpu/impl/GeneralDistance.cuh:runGeneralDistance for these functors; restated by
// oracle/faiss_oracle.c orc_flat_search_general.
void launch_flat_general(int metric, float metric_arg, const float* xq, int64_t ldq, int nq, const float* xb, int64_t ldb,
                         int nb, int d, unsigned long long* keys, hipStream_t stream);

// ------------------------------------------------------------------ Flat: fp16 MFMA filter + exact fp32 re-rank
// (flat_filter.hip; see the header there for the superset argument)
// geometry of the filter kernel: 0 = 4 waves x 64 queries (any dh, two workgroups per CU),
// 2 = 8 waves x 128 queries (dh == 128 only, one workgroup per CU)
int flat_filter_queries_per_block(int geom); // 256 / 1024
int flat_filter_chunks_per_split(int geom);  // chunk maxima per (query, split): 16 / 8
constexpr int kFilterTileRows = 64;
constexpr int kFilterSlab = 128;            // fp16 rows are padded to a multiple of this many halfs

struct FlatFilterParams {
    int metric;
    const _Float16* xqh; // [nq][ldqh] fp16 queries
    const float* xqn;    // [nq] exact fp32 squared norms of the queries (both metrics: error bound)
    const _Float16* xbh; // [nb][ldbh] fp16 database
    const float* xbhn;   // [nb + 64] start value of the approximate score: -|y|^2 / 2 (L2) or 0 (IP), then 64 x -inf
    int64_t ldqh, ldbh;
    int nq, nb, d, dh;   // dh = padded fp16 row length, multiple of kFilterSlab
    int geom;            // kernel geometry, see flat_filter_queries_per_block
    int cps;             // = flat_filter_chunks_per_split(geom)
    int nsplit, ngroups; // split s owns tiles s, s + nsplit, ...; cps * nsplit chunk maxima per query
    int tstride;         // maxima pass: every tstride-th tile of a split
    int k, cap;          // collect pass: segment capacity per (query, split)
    float yn_max;        // max squared norm over the database
    float* maxes;        // [nq][nsplit * cps] chunk maxima (maxima pass out, tighten in)
    float* thr;          // [nq] collect thresholds (tighten out; +inf = query not served by the filter)
    unsigned long long* res_keys; // [nq][nsplit][cap]  (score key << 32 | row)
    uint32_t* res_cnt;            // [nq][nsplit]
    uint32_t* flags;              // [nq] in: fp16 overflow of the query; out: |= segment overflow
    float* dump;                  // optional [nq][nb] approximate scores (tests)
    int exact_inputs;             // fp16 storage: q and y are fp16 values, the error band shrinks to the accumulation terms
    int dbg;                      // timing experiments only (env FAISS_AMD_FILTER_DBG): 1 no parking, 2 no sift, 4 no hits
};
// Bound on |t~ - s| for one query against any database row (see flat_filter.hip header).  fp16
// round-to-nearest: |dx| <= 2^-11 |x| in the normal range and <= 2^-25 below it, so
//   |<q~,y~> - <q,y>| <= 2^-10 (1+2^-11) |q||y| + 2^-25 (1+2^-11) sqrt(d) (|q| + |y|) + d 2^-50;
// MFMA fp32 accumulation and the exact path's own fmaf chain each contribute at most
// d 2^-24 |q||y| (first-order, doubled below); the L2 epilogue rounds fl(|q|^2+|y|^2) and the
// fmaf result once each: 2^-23 (|q|^2 + |y|^2) in distance units = 2^-24 (...) in score units.
// The L2 accumulators start from -|y|^2/2, so each of the (at most d) fp32 additions inside the MFMA
// chain also rounds that magnitude: + d 2^-24 |y|^2/2, doubled.
// A 1.25x safety factor covers the second-order terms and sqrtf.
// exact_inputs: queries and database ARE fp16 values (GpuIndexFlatConfig::useFloat16 storage): no conversion error,
// only the accumulation terms remain.
__host__ __device__ static inline float flat_filter_err_bound(int metric, int d, float xn, float yn_max,
                                                               bool exact_inputs = false) {
    const float nq = sqrtf(xn), ny = sqrtf(yn_max);
    float e = ((exact_inputs ? 0.f : 9.775e-4f /*2^-10 * 1.001*/) + 2.4e-7f * (float)d /*4 d 2^-24*/) * nq * ny +
              (exact_inputs ? 0.f : 3.0e-8f /*2^-25 * 1.001*/ * sqrtf((float)d) * (nq + ny)) + 1e-30f;
    if (metric == METRIC_L2) e += 6.0e-8f /*2^-24*/ * (xn + yn_max) + 6.0e-8f * (float)d * yn_max;
    return 1.25f * e;
}
// mode: 0 = maxima pass, 1 = collect pass, 2 = dump every score (tests)
// fp16 copy + range flags + |q|^2 (sequential fmaf chain) of n padded queries and *counter = 0, one launch
// ClearList: up to six word ranges zeroed by the same launch (the scratch counters of the IVF filter path: four fillBuffer packets
// of ~ 5 us each per search before round 6)
struct ClearList {
    uint32_t* p[6];
    uint32_t n[6]; // words
    int cnt;
    void add(void* ptr, size_t words) {
        if (ptr && words) p[cnt] = (uint32_t*)ptr, n[cnt] = (uint32_t)words, ++cnt;
    }
};
void launch_prep_queries(const float* xq_pad, int64_t ld, int64_t n, int d, int dpad, void* qh, int dh, uint32_t* flags,
                         float* qnorm, unsigned* counter, hipStream_t stream, const ClearList* clear = nullptr);
void launch_clear_words(const ClearList& c, hipStream_t stream);
void launch_flat_filter(const FlatFilterParams& p, int mode, hipStream_t stream);
void launch_flat_tighten(const FlatFilterParams& p, hipStream_t stream);
size_t flat_filter_lds_bytes();

// one-launch search of a small database (flat_filter.hip flat_small_fused_kernel, round 6): the IVF coarse quantizer
struct FlatSmallParams {
    int metric, nq, nb, d, dpad, k;
    const _Float16* xqh; // [nq][ldqh] fp16 queries (prep_queries)
    int64_t ldqh;
    const float* xq; // [nq][ldq] fp32 padded queries
    int64_t ldq;
    const float* xqn;      // [nq] |q|^2 (prep_queries)
    const uint32_t* flags; // [nq] fp16 range / NaN flags of the queries (prep_queries)
    const _Float16* xbh;   // [nb + tile][ldbh] fp16 rows
    int64_t ldbh;
    uint32_t* bad_out; // nullable [nq]: DEFERRED overflow -- a query the kernel cannot serve (fp16 range, > FS_CAP candidates) gets 1 here
                       // and labels of -1 instead of an entry in ovf_list; the caller redoes it later (GpuIndexIVF: with its own redo set)
    const void* xbo; // the same rows operand-major (launch_flat_operand_major): [ceil(nb / 32)][8 k-steps][64 lanes][16 bytes]
    const float* xbhn; // [nb + 64] -|y|^2 / 2 (L2) / 0 (IP), then -inf
    const float* xb;   // [nb][ldb] fp32 rows
    int64_t ldb;
    const float* xbn; // [nb]
    float yn_max;
    float* out_dis;
    int64_t* out_ids;
    uint32_t* ovf_list; // queries for the exact scan
    unsigned* ovf_cnt;
};
bool flat_small_fused_supported(int metric, int nb, int d, int dh, int k);
void launch_flat_operand_major(const void* xbh, int64_t ldbh, int nb, void* out, hipStream_t stream);
void launch_flat_small_fused(const FlatSmallParams& p, hipStream_t stream);

struct FlatRerankParams {
    int metric;
    int nq, k, kp, d, dpad, nsplit, cap;
    int gcap; // candidates per query the kernel can gather into LDS (<= 4096); more -> exact fallback for that query
    const unsigned long long* res_keys;
    const uint32_t* res_cnt;
    const uint32_t* flags;
    const float* xq;  // [nq][ldq] fp32 padded queries
    const float* xqn; // [nq]
    const float* xb;  // [nb][ldb] fp32 padded database (null with fp16 storage)
    const _Float16* xb16; // [nb][ldb16] fp16 rows, zero padded to a multiple of 8 (fp16 storage: the exact chain runs on them)
    const float* xbn; // [nb]
    int64_t ldq, ldb, ldb16;
    int exact_inputs; // see FlatFilterParams
    float yn_max;
    int64_t id_base;
    float* out_dis;     // [nq][k]
    int64_t* out_ids;   // [nq][k]
    uint32_t* ovf_list; // [nq] queries that must be re-run through the exact fp32 scan
    uint32_t* ovf_cnt;  // [1] (zeroed by the caller)
};
void launch_flat_rerank(const FlatRerankParams& p, hipStream_t stream);

// dst[i][0..dh) = fp16(src[i][0..d)), zero padded; *absmax_bits = max |x| (float bits, 0x7f800000 when
// a value is NaN/inf/outside the fp16 range); flags[i] = that condition per row (nullable)
void launch_convert_f16(const float* src, int64_t ld_src, int64_t n, int d, void* dst, int dh,
                        unsigned* absmax_bits, uint32_t* flags, hipStream_t stream);
void launch_max_f32(const float* x, int64_t n, unsigned* out_bits, hipStream_t stream);
void launch_half_norms(const float* xn, int64_t n, int npad, int metric, float* out, hipStream_t stream);
void launch_gather_rows(const float* src, int64_t ld, int width, const uint32_t* list, int n, float* dst,
                        hipStream_t stream);
void launch_scatter_results(const float* sd, const int64_t* si, int k, const uint32_t* list, int n, float* dd,
                            int64_t* di, hipStream_t stream);

// ------------------------------------------------------------------ IDSelector on the device
// faiss::IDSelector (faiss/impl/IDSelector.h:21-215) compiled to a postfix program that a kernel evaluates per id:
// leaves push one truth value (ALL, RANGE imin <= id < imax, SET = id in a sorted array of distinct ids -- what
// IDSelectorArray / IDSelectorBatch mean --, BITMAP = bit id of an n-byte bitmap, ids >= 8 n excluded), NOT / AND / OR /
// XOR combine the top of the stack.  All pointers are device pointers owned by the host-side selector object.
enum SelOp : int { SEL_ALL = 0, SEL_RANGE = 1, SEL_SET = 2, SEL_BITMAP = 3, SEL_NOT = 4, SEL_AND = 5, SEL_OR = 6, SEL_XOR = 7 };
constexpr int kSelMaxInstr = 24; // instructions per program (a tree of up to 12 leaves)
struct SelInstr {
    int op;
    int64_t a, b;    // RANGE: imin, imax; SET: a = number of ids; BITMAP: a = bytes
    const void* ptr; // SET: sorted int64 ids; BITMAP: the bytes
};
struct SelProgram {
    int n;
    SelInstr ins[kSelMaxInstr];
};
// mask bit i (bit i & 63 of word i >> 6, equally bit i & 31 of 32-bit word i >> 5) = selector(ids ? ids[i] : id_base + i)
// for i < n; bits n .. of the last word are 0.  *count (nullable, zeroed by the caller) += number of set bits.
void launch_selector_mask(const int64_t* ids, int64_t n, int64_t id_base, const SelProgram& prog, uint64_t* mask,
                          unsigned long long* count, hipStream_t stream);
// dst[i] = bit i of mask ? src[i] : excluded for i < n, dst[n .. n + npad) = pad: the start values / norms of a flat
// search restricted by a selector (FlatFilterParams::xbhn, FlatScanParams::xbn / ip_bias)
void launch_mask_bias(const float* src, const uint32_t* mask, int64_t n, int npad, float excluded, float pad, float* dst,
                      hipStream_t stream);

// ------------------------------------------------------------------ k-selection
struct SelectParams {
    int metric;
    int nq;
    int k;
    // candidates of query q: nseg segments; segment s holds seg_cnt[q*nseg+s] keys starting
    // at keys + (q_off ? q_off[q] : q*q_stride) + s*seg_stride
    const unsigned long long* keys;
    const int64_t* q_off; // nullable
    int64_t q_stride;
    int nseg;
    int64_t seg_stride;
    const uint32_t* seg_cnt; // [nq][nseg]
    // payload -> label translation
    int mode; // 0: label = payload (+ id_base); 1: IVF position -> ids[list_start + off];
              // 2: shard merge, payload = shard*k + rank -> merge_ids[shard][q][rank] + merge_base[shard]
    int64_t id_base;
    // mode 1:
    int nprobe;
    const uint32_t* ivf_prefix;  // [nq][nprobe+1] exclusive prefix of probed list lengths
    const int64_t* coarse_ids;   // [nq][nprobe]
    const int64_t* list_start;   // [nlist] first entry of each list in the arena
    const int64_t* arena_ids;    // [ntotal] user ids in arena order
    // mode 2:
    const int64_t* merge_ids;    // [nshard][nq][k]
    const int64_t* merge_base;   // [nshard] or null
    // outputs, best first; padded with (-1, neutral distance)
    float* out_dis;   // [nq][k]
    int64_t* out_ids; // [nq][k]
    // when set: only kth_out[q] = the ordkey (upper 32 bits) of the k-th smallest key is written (0xffffffff when the
    // query has no more than k keys... then every key qualifies); nothing else is produced
    uint32_t* kth_out; // [nq] or null
    // with kth_out (nseg == 1): the segment is also cut back to its keys <= the k-th (they move to its front) and
    // cnt_out[q] = their number
    uint32_t* cnt_out; // [nq] or null
    // upper bound of seg_cnt known to the caller (0 = unknown).  nseg == 1, k <= 256 and max_cnt <= 4096 (bound) /
    // 8192 (selection) take the wavefront-per-query kernel (select_kernels.hip: keys in registers, bisection)
    int64_t max_cnt;
    // set by launch_select_k: queries with <= 256 keys were served by small_select_kernel, the general wavefront kernel skips them
    int small_done;
};
// Exact k-selection: MSB radix select on 64-bit keys + bitonic sort of the k winners by
// (distance, label).  Replaces faiss/gpu/utils/BlockSelectKernel.cuh:15-132 and the
// pass1/pass2 kernels of faiss/gpu/impl/IVFUtilsSelect{1,2}.cu.
void launch_select_k(const SelectParams& p, hipStream_t stream);

// test hook: the k best entries of every row of vals [rows][cols] through one of the selection primitives
// (which = 0: select_k_kernel, 1: the LDS reservoir of wg_select.h streamed like the fused IVF scans, 2: the wavefront
// select of wave_select.h -- winners unordered); keys [rows * cols] and cnt [rows] are scratch
void launch_select_test(int which, int metric, const float* vals, int rows, int cols, int k, float* out_dis,
                        int64_t* out_ids, unsigned long long* keys, uint32_t* cnt, hipStream_t stream);

// Shard merge on the device: per-shard sorted results all_d/all_i [nshard][nq][k] are packed
// into keys [nq][nshard*k] (payload = shard*k + rank, so equal distances resolve to the lower
// shard / earlier rank = the smaller global id when shards hold successive id ranges); missing
// entries (label < 0) become invalid keys.  cnt[q] = nshard*k.
void launch_pack_merge_keys(int metric, const float* all_d, const int64_t* all_i, int nshard, int nq, int k,
                            unsigned long long* keys, uint32_t* cnt, hipStream_t stream);

// ------------------------------------------------------------------ IVF
// caller-supplied probe lists (search_preassigned): ids outside [0, nlist) become -1 = "no list"
void launch_ivf_sanitize_assign(int64_t* ids, int64_t n, int nlist, hipStream_t stream);
// prefix[q][0..nprobe] = exclusive prefix sum of list_len[coarse_ids[q][p]] (0 for id<0);
// total[q] = prefix[q][nprobe].  (faiss/gpu/impl/IVFUtils.cu:131-186 runCalcListOffsets)
void launch_ivf_prefix(const int64_t* coarse_ids, int nq, int nprobe, const uint32_t* list_len,
                       uint32_t* prefix, uint32_t* total, hipStream_t stream);

// probe_len[i] / probe_start[i] = list_len / list_start of coarse_ids[i] (0 for "no list"), i < n = nq * nprobe
void launch_ivf_probe_info(const int64_t* coarse_ids, int64_t n, const uint32_t* list_len, const int64_t* list_start,
                           uint32_t* probe_len, int64_t* probe_start, hipStream_t stream);

struct IvfScanParams {
    int metric;
    int nq, nprobe, d, dpad;
    const float* xq; // [nq][ldq]
    int64_t ldq;
    const int64_t* coarse_ids;  // [nq][nprobe]
    const float* coarse_dis;    // [nq][nprobe]
    const uint32_t* list_len;   // [nlist]
    const int64_t* list_start;  // [nlist]
    const uint32_t* prefix;     // [nq][nprobe+1]
    const int64_t* q_off;       // [nq] key offset of each query's candidate array
    unsigned long long* keys;
    // IVFFlat
    const float* arena_vecs; // [ntotal][ldv]
    int64_t ldv;
    // IVFPQ
    const float* centroids; // [nlist][ldc] coarse centroids (padded rows)
    int64_t ldc;
    int M, dsub;
    const float* pq_centroids; // [M][256][dsub]
    const float* pq_t;         // [256][M][dsub] transposed codebook
    const uint8_t* arena_codes; // rotated 64-row block layout, see pq_code_offset
    const float* arena_t2;      // [arena rows] L2: |r^|^2 + 2 <centroid, r^> of every stored vector
    const uint32_t* sel_mask;   // optional: one bit per arena row (launch_selector_mask), rows with a 0 bit are skipped
};
// One workgroup per (query, probe): direct sum((q-y)^2) over the list (L2) or dot (IP).
// Replaces faiss/gpu/impl/IVFFlatScan.cu:135-183 / IVFInterleaved.cuh:33-224.
void launch_ivfflat_scan(const IvfScanParams& p, hipStream_t stream);
// One workgroup per (query, probe): residual -> LDS lookup table [M][256] -> code scan.
// Replaces PQCodeDistances-inl.cuh:29-285 + PQScanMultiPassNoPrecomputed-inl.cuh:173-270;
// the table never leaves LDS.
void launch_ivfpq_scan(const IvfScanParams& p, hipStream_t stream);
size_t ivfpq_scan_lds_bytes(int M, int dpad);

// ------------------------------------------------------------------ fused IVF search (ivf_fused.hip)
struct IvfFusedParams {
    int metric;
    int kind; // 0 = IVFFlat, 1 = IVFPQ, 2 = IVF scalar quantizer
    int nq, nprobe, d, dpad;
    const float* xq; // [nq][ldq]
    int64_t ldq;
    const int64_t* coarse_ids;  // [nq][nprobe]
    const float* coarse_dis;    // [nq][nprobe]
    const uint32_t* list_len;   // [nlist]
    const int64_t* list_start;  // [nlist]
    const int64_t* arena_ids;   // [ntotal]
    // optional (both or neither): length and first arena row of every probed list, [nq][nprobe], gathered by
    // launch_ivf_probe_info -- one load round trip per workgroup instead of two dependent ones
    const uint32_t* probe_len;
    const int64_t* probe_start;
    // diagnostics (env FAISS_AMD_IVF_PHASES=1): shader clock ticks of workgroup phases summed over the launch,
    // [0] probe tables + query, [1] table build, [2] scan, [3] final selection + write-out, [4] workgroups; else null
    unsigned long long* phase_ticks;
    int k, kp, cap;             // kp = pow2 >= k; cap = LDS reservoir capacity (>= k + 512)
    int G, npc;                 // workgroups per query, probes per workgroup (G * npc >= nprobe)
    int defer_finish;           // G == 1: leave the final k-selection / id translation / ordering to launch_select_k
                                // (mode 1): the workgroup writes its reservoir (<= cap keys) to part_keys[q][cap]
    int nlut;                   // lookup tables in LDS: 2 = build of probe p+1 overlaps the scan of probe p
    float* out_dis;             // [nq][k]   (G == 1)
    int64_t* out_ids;           // [nq][k]   (G == 1)
    unsigned long long* part_keys; // [nq][G][k] partial winners (G > 1), merged by launch_select_k mode 1
    uint32_t* part_cnt;            // [nq][G]
    uint32_t* prefix_out;          // [nq][nprobe + 1] (G > 1)
    // IVFFlat
    const float* arena_vecs; // [ntotal][ldv]
    int64_t ldv;
    // IVFPQ
    const float* centroids; // [nlist][ldc]
    int64_t ldc;
    int M, dsub;
    const float* pq_centroids;  // [M][256][dsub]
    const float* pq_t;          // [256][M][dsub] transposed codebook
    const uint8_t* arena_codes; // rotated 64-row block layout, see pq_code_offset
    const float* arena_t2;      // [arena rows] L2: |r^|^2 + 2 <centroid, r^> of every stored vector
    // IVF scalar quantizer (kind 2): rows of sq_ld bytes in arena_codes; component i of a row decodes to
    // fmaf(code_i, sq_s[i], sq_b[i]) (fp16 codes: the half itself); centroids / ldc as for IVFPQ
    int sq_ct;             // SqCodeType
    int sq_dsq;            // d rounded up to 16 (components per row as stored)
    int sq_ld;             // bytes per arena row
    int sq_by_residual;
    const float* sq_s;     // [sq_dsq] scale per dimension (0 beyond d)
    const float* sq_b;     // [sq_dsq] offset per dimension
    // IDSelector (faiss/impl/IDSelector.h): one bit per arena row, built by launch_selector_mask from the stored ids for
    // the search in flight; a row whose bit is 0 never becomes a candidate.  null = no selector (the kernels are
    // instantiated separately for the two cases, the unfiltered scan carries no test)
    const uint32_t* sel_mask;
};
// code types of the scalar quantizer as the scan kernel sees them (16 components per lane chunk)
enum SqCodeType { SQ_U8 = 0, SQ_U4 = 1, SQ_U6 = 2, SQ_F16 = 3 };
__host__ __device__ inline int sq_chunk_bytes(int ct) { return ct == SQ_U8 ? 16 : ct == SQ_U4 ? 8 : ct == SQ_U6 ? 12 : 32; }
// Scalar-quantizer code layout: the arena is a sequence of 64-row blocks (lists start on block boundaries); a block
// holds its rows chunk-major, [chunk][64 rows][chunk bytes] with a chunk = 16 components, so that a wavefront reads
// one chunk of all 64 rows with one coalesced load and lane l owns row l.  ld = bytes per row = chunks * chunk bytes.
// Byte offset of byte `byte` (0 .. ld) of arena row `row`:
__host__ __device__ inline int64_t sq_code_offset(int64_t row, int byte, int ld, int chb) {
    const int c = byte / chb;
    return (row >> 6) * 64 * (int64_t)ld + (int64_t)c * 64 * chb + (row & 63) * chb + (byte - c * chb);
}
// table rows the fused scalar-quantizer scan keeps in LDS: the scale row, then one row per probe of a workgroup (L2
// with residual encoding: the query residual changes with the list) or a single one
inline int sq_table_rows(int metric, bool by_residual, int npc) { return 1 + ((metric == METRIC_L2 && by_residual) ? npc : 1); }
// One workgroup per (query, probe group): table build + code scan + running top-k all in LDS.
// Replaces PQCodeDistances + PQScanMultiPassNoPrecomputed + IVFUtilsSelect{1,2} (IVFPQ) and
// IVFInterleaved scan + scan2 (IVFFlat) of the reference in a single launch.
void launch_ivf_fused(const IvfFusedParams& p, hipStream_t stream);
// second launch of a search with IvfFusedParams::defer_finish: k-selection, id translation and ordering of every
// query's reservoir (needs probe_len / probe_start, part_keys, part_cnt, prefix_out of the scan launch)
void launch_ivf_finish(const IvfFusedParams& p, hipStream_t stream);
// does the problem fit the fused kernel (LDS budget, reservoir size)?  Returns cap / kp to use.
bool ivf_fused_supported(int kind, int M, int dpad, int k, int nprobe, int* cap_out, int* kp_out, int* nlut_out);
size_t ivf_fused_lds_bytes(int kind, int M, int dpad, int kp, int cap, int nprobe, int nlut);

// ------------------------------------------------------------------ list-major IVF search (ivf_listmajor.hip, round 3)
// Large batches: every inverted list is visited once per GROUP of the queries that probe it instead of once per
// query.  A workgroup takes (list, up to 128 of its queries, up to rows_per_item of its rows), keeps the queries as
// MFMA B operands in registers, streams the list's rows through LDS in 64-row tiles (IVFPQ: decoded there from the
// codes, once per workgroup) and computes the 64 x 128 distance block on v_mfma_f32_32x32x2_f32 -- bit for bit the
// k-ordered fmaf chain of the exact flat scan.  Two passes share the kernel:
//   pass 1  the first row chunk (rows_per_item rows) of the lists of every query's leading probes -- min_p1 of them, more if
//           it takes more to see k rows: EVERY distance is written to the query's key segment (dense slots, no atomics);
//           the k-th smallest key of the segment is an upper bound of the query's k-th best distance over all its
//           probes; the segment is cut back to those k keys (select_k_kernel with kth_out / cnt_out);
//   pass 2  everything else (the lists of the remaining probes, the further row chunks of the pass-1 lists): rows at or
//           below the bound are appended to the segment (one atomic per lane and 32-row block that holds any).
// The k best keys of the segment are the answer (select_k_kernel mode 1).  A segment that overflows is redone with
// all probes and rows in pass 1 (exact capacity).  SURVEY 7 H4; the reference scans query-major (IVFInterleaved.cuh:
// 33-224, PQScanMultiPassNoPrecomputed-inl.cuh:173-270).
// Arithmetic (restated by oracle/faiss_oracle.c orc_ivf_search, arith = 1):
//   IVFFlat  L2: max(0, fmaf(-2, <q, y>, |q|^2 + |y|^2)),  IP: <q, y>             (= the flat index's distances)
//   IVFPQ    L2: max(0, fmaf(-2, <q - c, r^>, |q - c|^2 + |r^|^2)),  IP: <q, c> + <q, r^>   (r^ = decoded residual)
//   <.,.> = the MFMA chain of flat_scan_kernel (orc_ip_chain); |y|^2, |r^|^2 sequential chains kept per stored row
//   (arena_rn); |q - c|^2 = the sum of the two interleaved half chains a lane pair holds (orc_lm_residual_norm).
struct IvfLmItem {
    int bucket, qt, rt; // bucket = 2 * list + (0: pairs of pass 1, 1: the other pairs), group of qpi pairs, row chunk
    int both;           // 1: the item's pairs are those of bucket and bucket + 1 (all pairs of the list)
};
constexpr int kLmRowsPerItem = 1024; // rows of a list per work item (16 tiles); pass 1 sees the first chunk of a list
constexpr int kLmQueriesPerItem = 64; // two 32-query blocks (x the two 32-row blocks of a tile = 4 waves)
struct IvfLmParams {
    int metric;
    int kind; // 0 = IVFFlat, 1 = IVFPQ, 2 = IVF scalar quantizer (round 3, second session)
    int nq, nprobe, d, dpad, nlist, k;
    const float* xq; // [nq][ldq] padded queries
    int64_t ldq;
    const float* xqn;           // [nq] |q|^2, sequential chain (IVFFlat L2)
    const int64_t* coarse_ids;  // [nq][nprobe]
    const float* coarse_dis;    // [nq][nprobe] (IVFPQ IP: first term)
    int pre_cleared;            // the caller zeroed ovf[0], bucket_cnt, the norm bounds and (kind 2) qflags itself, in one launch
                                // (launch_clear_words) or inside launch_prep_queries: the launchers below skip their memsets
    const uint32_t* coarse_bad; // nullable [nq]: queries the one-launch coarse quantizer handed back (FlatSmallParams::bad_out): redo set
    const uint32_t* list_len;   // [nlist]
    const int64_t* list_start;  // [nlist]
    // ---- plan, built on the device by launch_ivf_lm_plan
    uint32_t* prefix;        // [nq][nprobe + 1] exclusive prefix of the probed lists' lengths = scan positions
    uint32_t* prefix1;       // [nq][nprobe + 1] the same over min(length, rows_per_item): segment slots of pass 1
    uint32_t* p0;            // [nq] probes of pass 1
    uint32_t* cnt;           // [nq] keys in the segment (after the plan: the pass-1 rows)
    uint32_t* bucket_cnt;    // [2 nlist] pairs per (list, pass); zeroed by the plan launcher
    uint32_t* bucket_fill;   // [2 nlist] cursors; zeroed by the plan launcher
    uint32_t* bucket_start;  // [2 nlist + 1]
    uint32_t* pairs;         // [nq * nprobe] pair = q * nprobe + p, grouped by bucket
    IvfLmItem* items;        // [max_items], pass 1 first
    uint32_t* item_bounds;   // [8] = 0, items of pass 1, all items, 1 if max_items was too small (a bug: the host checks), [4] work counter of pass 2
                             // filter sweeps: + per-XCD work counters at [kLmXcdCtr + (sweep * 8 + xcd) * 32] (kLmBoundsBytes in all)
    int max_items;
    int qpi;                 // queries per work item: 32 (register-fed IVFFlat kernel: one wavefront per item) or 64
    int rows_per_item;       // multiple of 64
    int force_all;           // every probe in pass 1
    int min_p1;              // at least this many probes of every query in pass 1
    int dbg;                 // timing experiments (env FAISS_AMD_LM_DBG): 1 no key writes, 2 no MFMAs, 4 no tile loads (LDS-tile
                             // kernel), 8 rows of four lists only (L2-resident), 64 static deal of the items instead of the
                             // work counter, 128 workgroup-level draws of four consecutive items (register-fed kernels)
    // ---- candidates
    unsigned long long* keys; // [nq][stride]
    int64_t stride;
    uint32_t* thr;            // [nq] pass 2 admits ordkey(distance) <= thr
    uint32_t* ovf;            // [1 + nq]: number of queries whose segment overflowed, then their numbers
    // ---- storage
    const float* arena_vecs; // IVFFlat rows [arena rows][ldv]
    int64_t ldv;
    const float* arena_rn;   // [arena rows] L2: squared norm of the stored (decoded) row, sequential chain
    const uint8_t* arena_codes; // IVFPQ: rotated 64-row blocks (pq_code_offset)
    int M, dsub;
    const float* pq_centroids;  // [M][256][dsub]
    const float* centroids;     // [nlist][ldc] coarse centroids
    int64_t ldc;
    // IVF scalar quantizer (kind 2): arena_codes in the chunk-major 64-row blocks of sq_code_offset; component j of a row
    // decodes to b_j + s_j * code_j (fp16 codes: the half itself).  The scan never materialises it.  With the codes
    // centred on the middle of their range, code' = code - mid (mid = 127.5 / 31.5 / 7.5 for 8- / 6- / 4-bit codes, 0 for fp16),
    // b' = fmaf(mid, s, b) and a = (q [- centroid]) - b', the B operand is a o s and the A operand code' as a float:
    //   L2  max(0, fmaf(-2, <a o s, code'>, |a|^2 + |s o code'|^2)),   IP  (<q, b'> + coarse) + <q o s, code'>
    // (|s o code'|^2 per stored row in arena_rn, launch_ivfsq_row_norms; |a|^2 and <q, b'> as the two interleaved half
    // chains of a lane pair, like IVFPQ's |q - c|^2)
    int sq_ct;             // SqCodeType
    int sq_ld;             // bytes per arena row
    int sq_by_residual;
    const float* sq_s;     // [>= dpad] scale per dimension (0 beyond d)
    const float* sq_b;     // [>= dpad] b': offset per dimension moved to the middle of the code range (fp16: s = 1, b' = 0)
    const float* sq_zero;  // [>= dpad] zeros: the "centroid" of a search without residual encoding
    const float* sq_b_plain; // [>= dpad] b (not centred): the table of the query-major scan, for the filter path's exact rerank
    // ---- f16 filter scan (round 4, ivf_lm_filter.hip; kinds 0 and 1).  Every (query, probe) pair is a "pass 2" pair
    // (filter = 1: the plan puts p0 = 0); two sweeps over the same work items -- granule minima of the ESTIMATED
    // distances, then the collection of every row whose estimate is at or below the query's threshold -- and the
    // exact arithmetic of the query-major scan on the few survivors (launch_ivf_lmf_rerank).
    int filter;
    // kind 1 through the DECODED-residual copy (ivf_lm_filter.hip lmf_pq_decode_kernel): the sweeps are the pair-operand IVFFlat
    // kernel over arena_h = fp16 r^, prepared like the scalar quantizer with sq_s = 1, sq_b = 0; bound / tighten / rerank are IVFPQ's
    int lmf_pairb;
    int gran_blocks;            // G: 32-row blocks per granule; a granule of a list = 32 G rows = two slots (lane halves)
    int min_stride;             // sweep 1 looks at every min_stride-th 32-row block of a row chunk (1 = all rows)
    // sweep 1 looks only at the first sample_rows rows of every work item's row chunk (0 = the whole chunk; a multiple of
    // 32 gran_blocks below rows_per_item): ANY subset of the rows bounds the k-th best estimate from above, and a
    // contiguous prefix streams at the full rate (every 2nd 32-row block of the codes moved 2 KB of every 4: round 4's sampling).
    // The looser bound admits ~ 1 / (sampled share) as many candidates; launch_ivf_lmf_tighten cuts them back to the rows
    // inside the band of the k-th best COLLECTED estimate before anything exact is computed.
    int sample_rows;
    uint32_t* prefixg;          // [nq][nprobe + 1] exclusive prefix of 2 * ceil(len / (32 G)): granule slots of the probes
    uint32_t* gmin;             // [nq][gstride] ordkey of the best estimate in every granule slot
    int64_t gstride;
    float* thr_f;               // [nq] collect threshold on the estimate (bound + 2 x error band; +/-inf = everything / nothing)
    const void* xq16;           // [nq][ldq16] fp16 queries, zero padded (kind 0)
    int64_t ldq16;
    // fp16 shadow of the IVFFlat rows in OPERAND-MAJOR 32-row blocks (lists start on multiples of 32 rows): block b = arena
    // rows 32 b .. 32 b + 31 holds ldh / 16 k-steps of 1 KB each, k-step s = the 16-byte pieces [lane h * 32 + j] = coordinates
    // 16 s + 8 h .. + 7 of row j -- exactly what lane (h, j) feeds the MFMA, so every load instruction of a wavefront reads
    // one contiguous KB (the row-major shadow of the first version moved 32-byte segments: 3.2 TB/s at best)
    const void* arena_h;
    int64_t ldh;                // fp16 coordinates per row, zero padded to a multiple of 16
    // IVFPQ: the code bytes again, operand-major (kind 1): per 32-row block and lane (h, j) the cs_bpl bytes the lane's
    // operands need -- byte b belongs to k-step b / ncode (ncode = max(1, 8 / dsub) codes per 8-coordinate operand), code
    // (16 s + 8 h) / dsub + b % ncode of row j -- stored [piece][lane][cs_piece bytes] (cs_piece = 16 when 16 | cs_bpl, else 4):
    // the sweeps load their codes straight into registers, no LDS staging, no un-rotation
    const uint8_t* arena_cs;
    int cs_bpl, cs_piece;
    // PQ64 over d = 128 only: arena_cs holds 9-bit fields copy << 8 | slot for a codebook kept TWICE in LDS with different
    // code -> bank maps, the copy of every (row, sub-quantizer) chosen when the copy of the codes was written so that the
    // gathers of a 32-lane group spread over the banks (ivf_lm_filter.hip, lmf_code_choice_kernel): 40 bytes per lane and block
    int cs_choice;
    int lmf_pair;        // IVFFlat / scalar-quantizer sweeps at <= 128 coordinates: two-wave workgroups in lock-step over sibling items (round 6)
    int lmf_fast_gather; // IVFPQ sweeps at PQ64 over d = 128, one copy: one-instruction codebook gathers (ivf_lm_filter.hip FG, round 6)
    const void* pq16;           // [M][256][dsub] fp16 codebook (kind 1)
    const float* pq_t;          // [256][M][dsub] fp32 codebook, transposed: the order the exact path builds its table in
    const uint32_t* qflags;     // [nq] nonzero: the query leaves the fp16 range / holds NaN -> not filtered (fallback)
    // kind 1: the sweeps' B operands, prepared once per search (launch_ivf_lmf_pq_prepare) instead of at every work item:
    // L2 [nq * nprobe][d] fp16 of (q - centroid of the probe), pair_xh [nq * nprobe] = -|q - c|^2 / 2; inner product
    // [nq][d] fp16 of q (the coarse term of a pair comes from coarse_dis)
    void* pair16;
    float* pair_xh;
    uint16_t* cand_pr;          // [nq][stride] probe number of every collected candidate (beside keys)
    int64_t* row_base;          // [nq][nprobe] list_start[list of the probe] - prefix[q][probe]: arena row = row_base + scan position
                                // (written by the plan when not null; the rerank kernels read it instead of three dependent loads)
    const float* arena_t2;      // kind 1, L2: the per-row term of the query-major scan (rerank)
    const float* xn_full;       // [nq] |q|^2 (kind 0: == xqn)
    float yn_max;               // max |y|^2 over the stored rows (kind 0) / upper bound of |r^|^2 from the codebook (kind 1)
    float cn_max;               // kind 1: max |centroid|^2
    // kind 2 (scalar quantizer behind the filter): [nq] max over the probes of |a|^2 (L2) / |q|^2 (IP), |b'|^2 of the centred
    // offsets, max |s o code'|^2 over the stored rows (L2), d mid^2 (the centring constant's share of the plain codes' norm)
    const float* an_bound;
    float bn, rn_max, cmid2;
    float* band_out;            // optional [nq]: the error band E_q the bound kernel used (tests)
    float* err_f;               // [nq] the error band E_q (written by the bound kernel, read by launch_ivf_lmf_tighten)
    // IDSelector of the search in flight (filter path only): one bit per arena row (launch_selector_mask), null = none.
    // Excluded rows take no part in the bound nor in the collection: the result is that of the selected subset.
    const uint32_t* sel_mask;
    // rerank kernels: when fin_dis is set (k <= kLmfFusedSelectK, stride <= kLmfFusedSelectN) the workgroup that re-derived a
    // query's candidates also selects its k best -- (distance, scan position) picks them, (distance, label) orders them,
    // the rule of select_k_kernel / wave_select_kernel -- and writes the result rows; no selection launch follows
    float* fin_dis;
    int64_t* fin_ids;
    // IVFPQ rerank, PQ64 over d = 128 (lmf_rerank_pq64_kernel, a wavefront per query): workgroups to launch (one per CU; 0: the
    // workgroup-per-query kernel serves the shape too)
    int rr_blocks;
    const int64_t* arena_ids;
};
constexpr int kLmfFusedSelectK = 256, kLmfFusedSelectN = 1024;
// |estimate - exact| <= this for every stored row, whatever the data: `estimate` = what the f16 MFMA sweeps of
// ivf_lm_filter.hip compute (L2: fmaf(-2, <f16 q', f16 y'>, |q'|^2 + |y'|^2) with the two norms as fp32 chains; IP:
// <f16 q', f16 y'> [+ coarse term]), `exact` = the distance the query-major scan returns for the same row (and the rerank
// kernel recomputes).  q' / y' = query / stored row (kind 0) or residual query / decoded residual (kind 1); xn >= |q'|^2,
// yn_max >= |y'|^2.  Terms: both operands rounded to fp16 (2^-11 relative each, Cauchy-Schwarz), products exact in fp32,
// d fp32 accumulations (4 d 2^-24: margin for the pipe's internal order), fp16 denormals; the fp32 chains of the norms,
// the final fmaf and the exact path's own rounding ((d + 8) 2^-23 of the magnitudes involved).  extra: kind 1's table
// grid and per-row terms, see lmf_bound_kernel.  Verified on hardware by test_list_filter_error_bound_holds.
// the matrix pipe's share alone: |<f16 B, f16 A> (fp32 accumulation) - <B, A>| for operands of squared norms xn, yn
__host__ __device__ static inline float ivf_filter_err_mfma(int d, float xn, float yn) {
    const float nq = sqrtf(xn), ny = sqrtf(yn);
    return (9.775e-4f /*2^-10 * 1.001*/ + 2.4e-7f * (float)d) * nq * ny + 3.0e-8f * sqrtf((float)d) * (nq + ny);
}
__host__ __device__ static inline float ivf_filter_err_bound(int metric, int d, float xn, float yn_max, float extra = 0.f) {
    const float nq = sqrtf(xn), ny = sqrtf(yn_max);
    float e = (9.775e-4f /*2^-10 * 1.001*/ + 2.4e-7f * (float)d) * nq * ny + 3.0e-8f * sqrtf((float)d) * (nq + ny);
    if (metric == METRIC_L2) e = 2.f * e + 1.2e-7f * (float)(d + 8) * (xn + yn_max);
    else e += 1.2e-7f * (float)(d + 8) * nq * ny;
    return 1.25f * (e + extra) + 1e-30f;
}
constexpr int kLmXcdCtr = 64;                                  // first per-XCD counter (u32 index into item_bounds), 128 bytes apart
constexpr size_t kLmBoundsBytes = (kLmXcdCtr + 2 * 8 * 32) * 4; // bytes behind IvfLmParams::item_bounds
constexpr int kLmfQueryBlocks = 3;  // 32-query MFMA blocks per work item of the filter sweeps (B operands: 32 VGPRs each)
bool ivf_lmf_supported(int kind, int d, int dpad, int M);
int ivf_lmf_queries_per_item(int kind, int d);
int ivf_lmf_row_halfs(int d); // halfs per row of the IVFFlat fp16 shadow and of the fp16 queries
int ivf_lmf_grid_blocks(const IvfLmParams& p, int num_cus);
// mode 1: granule minima -> gmin; mode 2: collect (keys / cand_pr / cnt); mode 3 (tests): every estimate as a key at its
// scan position (keys [nq][stride], stride >= rows probed)
void launch_ivf_lmf_sweep(const IvfLmParams& p, int mode, int grid_blocks, hipStream_t stream);
// thr_f[q] from gmin (k-th best granule estimate + 2 x ivf_filter_err_bound); queries with qflags set get "nothing" and are
// appended to ovf (zeroed here).  xn_bound: [nq] upper bound of |q'|^2 over the query's probes (kind 0: |q|^2).
void launch_ivf_lmf_bound(const IvfLmParams& p, const float* xn_bound, hipStream_t stream);
// Behind sweep 2, before the rerank: cnt[q] > stride -> listed in ovf (what launch_ivf_lm_clamp does); otherwise T2 = the k-th
// best estimate among the query's collected candidates -- the k-th best estimate of ALL its rows, since everything at or
// below thr_f >= T2 was collected -- and only candidates with estimate <= T2 + 2 E_q stay (compacted in place, cnt updated):
// the smallest superset the error band allows, whatever sample the first sweep's bound came from.  fin_cap > 0: the rerank
// workgroup will select in LDS; a query left with more than fin_cap candidates is listed in ovf instead.
void launch_ivf_lmf_tighten(const IvfLmParams& p, int fin_cap, hipStream_t stream);
// granule slots (per lane half) sweep 1 fills for a list of `len` rows: all its granules, or those of the sampled prefixes
__host__ __device__ static inline uint32_t ivf_lmf_list_granules(uint32_t len, uint32_t rows_per_item, uint32_t grows,
                                                                uint32_t sample_rows) {
    if (sample_rows == 0 || sample_rows >= rows_per_item) return (len + grows - 1) / grows;
    const uint32_t nfull = len / rows_per_item, rem = len - nfull * rows_per_item, gs = sample_rows / grows;
    const uint32_t gr = (rem + grows - 1) / grows;
    return nfull * gs + (gr < gs ? gr : gs);
}
// kind 1: xn_bound[q] = max over the probes of |q - c|^2; pair16 / pair_xh = the sweeps' B operands and query terms
void launch_ivf_lmf_pq_prepare(const IvfLmParams& p, float* xn_bound, hipStream_t stream);
// keys[q][0 .. cnt[q]) <- the exact distance of the query-major scan (ivf_fused.hip) for the same row, bit for bit
void launch_ivf_lmf_rerank(const IvfLmParams& p, hipStream_t stream);
// fp16 shadow of the rows of every list (rows [0, len) of each; dh halfs per row, zero padded) + max |y|^2 over them
// (float bits by atomicMax; 0x7f800000 when a stored value is NaN / inf / beyond the fp16 range)
// first_row (device, [nlist], may be null = every list from row 0): only the 32-row blocks from that row of the list on are
// (re)written, 0xffffffff = the list is skipped -- the incremental maintenance of add() (GpuIndexIVF::lmf_patch_)
void launch_ivf_lmf_shadow(const float* arena, int64_t ldv, const float* arena_rn, int d, int nlist, const uint32_t* list_len,
                           const int64_t* list_start, void* arena_h, int dh, unsigned* yn_max_bits, const uint32_t* first_row,
                           hipStream_t stream);
// scalar quantizer: the operand-major fp16 blocks of launch_ivf_lmf_shadow filled with the CENTRED codes (exact in fp16);
// stat_bits[0] = max |s o code'|^2 of the rows written (from arena_rn; 0x7f800000: a stored fp16 code is not finite),
// stat_bits[1] = an upper bound of max |code'|^2 (fp16 codes; the integer types use d mid^2)
void launch_ivf_lmf_sq_shadow(const uint8_t* arena, int ct, int ld, const float* arena_rn, int d, int nlist, const uint32_t* list_len,
                              const int64_t* list_start, void* arena_h, int dh, unsigned* stat_bits, const uint32_t* first_row,
                              hipStream_t stream);
// kind 2: pair16 / pair_xh / qflags, xn_bound[q] = max |B|^2, an_bound[q] = max |a|^2 (see lmf_sq_prepare_kernel)
void launch_ivf_lmf_sq_prepare(const IvfLmParams& p, float* xn_bound, float* an_bound, hipStream_t stream);
// kind 2: keys[q][0 .. cnt[q]) <- the exact distance of ivfsq_fused_kernel for the same row (ivf_fused.hip: same code)
void launch_ivf_lmf_rerank_sq(const IvfLmParams& p, hipStream_t stream);
// IVFPQ: operand-major copy of the codes of every list (see IvfLmParams::arena_cs); bytes per lane and block / piece size
void ivf_lmf_code_shadow_shape(int d, int M, int* bpl, int* piece);
bool ivf_lmf_choice_shape(int d, int M); // the shape the two-copy codebook serves
bool ivf_lmf_pq_decoded_supported(int d, int dpad, int M);
// fp16 copy of the decoded residuals of every list in the operand-major blocks of launch_ivf_lmf_shadow (pq: [M][256][dsub] fp32)
void launch_ivf_lmf_pq_decode(const uint8_t* arena_codes, const float* pq, int d, int M, int nlist, const uint32_t* list_len,
                              const int64_t* list_start, void* arena_h, int dh, const uint32_t* first_row, hipStream_t stream);
// the copy of the codes in the two-copy format (2560 bytes per 32-row block); first_row as for launch_ivf_lmf_code_shadow
void launch_ivf_lmf_code_choice(const uint8_t* arena_codes, int nlist, const uint32_t* list_len, const int64_t* list_start,
                                uint8_t* arena_cs, const uint32_t* first_row, hipStream_t stream);
void launch_ivf_lmf_code_shadow(const uint8_t* arena_codes, int d, int M, int nlist, const uint32_t* list_len,
                                const int64_t* list_start, uint8_t* arena_cs, const uint32_t* first_row, hipStream_t stream);
// kind 0: M unused; kind 1: M = sub-quantizers; kind 2: M = SqCodeType
bool ivf_lm_supported(int kind, int dpad, int M, int d);
// prefix / p0 / cnt, the pairs grouped by (pass, list), the work items.  (4 launches + 1 memset)
void launch_ivf_lm_plan(const IvfLmParams& p, hipStream_t stream);
// pass = 1 or 2; grid_blocks workgroups walk the pass's items (a multiple of 8: consecutive items on one XCD)
void launch_ivf_lm_scan(const IvfLmParams& p, int pass, int grid_blocks, hipStream_t stream);
// workgroups of the scan kernel that are resident per CU (IVFFlat: two, with two tiles in LDS each; IVFPQ: three)
int ivf_lm_blocks_per_cu(int kind);
int ivf_lm_queries_per_item(int kind); // IvfLmParams::qpi the scan kernel of this index type works with
// persistent workgroups to launch for this problem (IVFPQ with the codebook in LDS: one 8-wave workgroup per CU)
int ivf_lm_grid_blocks(const IvfLmParams& p, int num_cus);
bool ivf_lm_pq_lds_supported(int d, int dpad, int M);
// cnt[q] > stride -> cnt[q] = stride, the query is listed in ovf
void launch_ivf_lm_clamp(const IvfLmParams& p, hipStream_t stream);
// IVF scalar quantizer: out[row] = sum_j (s_j * code'_j)^2 (centred codes; fp16: the half itself), sequential fmaf chain over
// j = 0 .. d - 1, for the n arena rows dest[i] (dest != null; negative entries skipped) or row0 .. row0 + n - 1
void launch_ivfsq_row_norms(const uint8_t* arena, int ct, int ld, int d, const float* sq_s, const int64_t* dest, int64_t row0,
                            int64_t n, float* out, hipStream_t stream);
// out[dest[i]] = |x_i|^2 as the sequential fmaf chain of l2_norms_kernel, for dest[i] >= 0
void launch_l2_norms_scatter(const float* x, int64_t ld, int64_t n, int d, const int64_t* dest, float* out, hipStream_t stream);

// ------------------------------------------------------------------ IVF storage (round 2)
// Every inverted list owns a row range [list_start, list_start + list_cap) of one arena, list_cap a multiple of
// the granule (64 rows for IVFPQ, 8 for IVFFlat); lists grow in place inside their slack and move to the end of
// the arena (geometric capacity) when it is used up -- the reference's one-growable-DeviceVector-per-list
// (faiss/gpu/impl/IVFBase.cuh:220-299, DeviceVector.cuh) folded into one allocation.
//
// IVFPQ code layout: the arena is a sequence of 64-row BLOCKS (lists start on block boundaries).  A block holds
// its rows' codes as [M / CH chunks][64 rows][CH bytes] (CH = 16 when M % 16 == 0, else 4, else 1: one coalesced
// 1 KB load per chunk and wavefront), and row r stores its code ROTATED by r mod 64: stored byte j = code byte
// (j + r) mod M.  Lane l of a wavefront scanning a block therefore looks up sub-quantizer (j + l) mod M at step j:
// with the lookup table laid out [256][M] the 32 lanes of an LDS access group hit 32 different banks -- the scan's
// 64 gathers per code run conflict-free instead of 3.5-way conflicted (profiles/r02_a_*: 60 % of the LDS cycles of
// the round-1 kernel were bank conflicts).  (The reference has its own interleaved layouts for the same reason of
// access shape, faiss/gpu/impl/InterleavedCodes.cpp; translation happens in copy_lists / get_list_codes.)
constexpr int kPqBlockRows = 64;
__host__ __device__ static inline int pq_chunk_bytes(int M) {
    return (M & 15) == 0 ? 16 : (M & 3) == 0 ? 4 : 1;
}
// byte offset inside the code arena of code byte m of arena row `row`
__host__ __device__ static inline size_t pq_code_offset(int M, int64_t row, int m) {
    const int ch = pq_chunk_bytes(M);
    const int l = (int)(row & 63);
    int j = (m - l) % M;
    if (j < 0) j += M;
    return (size_t)(row >> 6) * 64 * M + (size_t)(j / ch) * 64 * ch + (size_t)l * ch + (size_t)(j % ch);
}

// ---- IVFPQ lookup tables on a power-of-two grid (round 2).
// A query's table lut[m][c] = <q_m, pq[m][c]> (fmaf chain over the dsub coordinates) is rounded to multiples of
// delta = 2^ed, ed chosen per query such that 2^24 * delta > 1.0001 * B with B = sum_m max_c |lut[m][c]| (sequential
// fp32 sum over m).  Every partial sum of table entries is then a multiple of delta below 2^24 * delta, i.e. EXACT in
// fp32: the ADC sum S = sum_m lut[m][code_m] no longer depends on the summation order, which is what lets each lane
// walk the sub-quantizers in its own rotated order (pq_code_offset) and lets the result be independent of where a
// vector is stored.  Rounding error per entry <= delta / 2 <= 2^-24 * 1.0001 * B: the same order as the rounding
// of a plain fp32 sum of M terms of that magnitude (the reference GPU index offers an fp16 table,
// GpuIndexIVFPQConfig::useFloat16LookupTables -- 2^-11 per entry).  oracle/faiss_oracle.c restates this bit for bit.
// Returns false (no rounding: NaN / inf / all-zero tables) or true with delta and 1 / delta.
__host__ __device__ static inline bool pq_lut_grid(float B, float* delta, float* inv) {
    if (!(B > 0.f)) return false;
    const float Bs = B * 1.0001f;
    if (!(Bs <= FLT_MAX)) return false;
    union {
        float f;
        uint32_t u;
    } v;
    v.f = Bs;
    int ed = (int)((v.u >> 23) & 255u) - 126 - 24; // Bs < 2^(ed + 24)
    if (ed < -126) ed = -126;
    v.u = (uint32_t)(ed + 127) << 23;
    *delta = v.f;
    v.u = (uint32_t)(127 - ed) << 23;
    *inv = v.f;
    return true;
}

// ---- add path (faiss/gpu/impl/IVFBase.cu:595-905 addVectors / IVFAppend.cu), all on the device:
// histogram of the coarse labels per chunk of `chunk` consecutive vectors: hist[c][l] (zeroed by the caller)
void launch_ivf_histogram(const int64_t* labels, int64_t n, int nlist, int chunk, uint32_t* hist, hipStream_t stream);
// per list: hist[c][l] <- list_len[l] + sum_{c' < c} hist[c'][l] (offset of the chunk's first entry inside the
// list); new_len[l] = list_len[l] + total
void launch_ivf_chunk_scan(uint32_t* hist, int nchunks, int nlist, const uint32_t* list_len, uint32_t* new_len,
                           hipStream_t stream);
// dest[i] = list_start[label] + (offset of vector i inside its list), insertion order kept (a stable counting
// sort: one wavefront per chunk walks its vectors in order); label < 0 (NaN vectors) -> -1
void launch_ivf_rank(const int64_t* labels, int64_t n, int nlist, int chunk, uint32_t* hist,
                     const int64_t* list_start, int64_t* dest, hipStream_t stream);
// k-means centroid update (faiss/Clustering.cpp:307-324): start[0..n] = exclusive prefix of cnt; order[dest[i]] = i;
// centroids[c][j] = float(sum over the members of c, in index order, of double(x[i][j]) / count) for count > 0
void launch_exclusive_scan(const uint32_t* cnt, int n, int64_t* start, hipStream_t stream);
void launch_invert_dest(const int64_t* dest, int64_t n, uint32_t* order, hipStream_t stream);
// product-quantizer training, all sub-spaces in one k-means (ivf_kernels.hip): labels m * 256 + c over M * nt points
bool pq_train_batched_supported(int dsub);
void launch_pq_train_init(const float* res, int64_t ld, int M, int dsub, const uint32_t* sel, float* cen, hipStream_t stream);
void launch_pq_train_assign(const float* res, int64_t ld, int64_t nt, int M, int dsub, const float* cen, int64_t* labels,
                            hipStream_t stream);
void launch_pq_train_update(const float* res, int64_t ld, int64_t nt, int M, int dsub, const uint32_t* order,
                            const int64_t* start, const uint32_t* cnt, float* cen, hipStream_t stream);
void launch_kmeans_update(const float* x, int64_t ldx, int d, const uint32_t* order, const int64_t* start,
                          const uint32_t* cnt, int k, float* centroids, hipStream_t stream);
// list relocation: job j moves rows[j] rows of bytes_per_row bytes from row src[j] to row dst[j] (regions never
// overlap: destinations are fresh space at the end of the arena)
struct IvfMoveJob {
    int64_t src, dst, rows;
};
void launch_ivf_move(const uint8_t* arena_src, uint8_t* arena_dst, const IvfMoveJob* jobs, int njobs,
                     int bytes_per_row, hipStream_t stream);
void launch_iota_i64(int64_t* out, int64_t n, int64_t base, hipStream_t stream);
void launch_fill_knn(float* D, int64_t* I, int64_t n, int metric, hipStream_t stream);
void launch_ivfflat_append(const float* x, int64_t ldx, int n, int d, const int64_t* dest,
                           float* arena_vecs, int64_t ldv, int dpad, hipStream_t stream);
// out[id - i0][0..d) = the stored row whose user id is `id`, for every stored id in [i0, i0 + ni)
// (GpuIndexIVFFlat::reconstruct_n, faiss/gpu/impl/IVFFlat.cu:289-335, as one pass over the id arena)
void launch_ivfflat_rows_by_id(const float* arena_vecs, int64_t ldv, const int64_t* arena_ids, const int64_t* list_start,
                               const uint32_t* list_len, int nlist, int d, int64_t i0, int64_t ni, float* out,
                               hipStream_t stream);
// ---- IVF scalar quantizer (faiss/impl/ScalarQuantizer.h, faiss/gpu/impl/GpuScalarQuantizer.cuh)
// qtype values of faiss::ScalarQuantizer::QuantizerType that this backend stores (ScalarQuantizer.h:27-34)
enum SqQuantizerType { QT_8bit = 0, QT_4bit = 1, QT_8bit_uniform = 2, QT_4bit_uniform = 3, QT_fp16 = 4, QT_8bit_direct = 5, QT_6bit = 6 };
// encode n staged vectors (x: [n][ldx] fp32, residual w.r.t. centroids[labels[i]] when by_residual) exactly as
// ScalarQuantizer::compute_codes does (impl/scalar_quantizer/quantizers.h:60-150, codecs.h:24-110) and write them to
// arena rows dest[i] (row stride ld bytes, zero padded).  vmin / vdiff: [d] (uniform types: replicated).
void launch_ivfsq_encode_append(int qtype, const float* x, int64_t ldx, int n, int d, const int64_t* labels,
                                const int64_t* dest, const float* centroids, int64_t ldc, bool by_residual,
                                const float* vmin, const float* vdiff, uint8_t* arena, int ld, hipStream_t stream);
// plain rows [n][ld] (the reference's entries, zero padded to whole chunks) <-> the block layout: list l's rows
// src_start[l] .. of `rows` go to arena rows dst_start[l] .. (pack); rows [row0, row0 + n) of the arena come back as
// plain rows (unpack)
void launch_ivfsq_pack_lists(const uint8_t* rows, const int64_t* src_start, const int64_t* dst_start, const uint32_t* len,
                             int nlist, int ld, int chb, uint8_t* arena, hipStream_t stream);
void launch_ivfsq_unpack_rows(const uint8_t* arena, int64_t row0, int64_t n, int ld, int chb, uint8_t* rows,
                              hipStream_t stream);
// per-dimension minimum and maximum of x [n][ldx] over column blocks: out [nblocks][2][d] (min row, max row)
int ivfsq_minmax_blocks(int64_t n);
void launch_ivfsq_minmax(const float* x, int64_t ldx, int64_t n, int d, float* out, hipStream_t stream);
// PQ-encode the residuals and write the code bytes into the rotated block layout
void launch_ivfpq_encode_append(const float* x, int64_t ldx, int n, int d, const int64_t* labels,
                                const int64_t* dest, const float* centroids, int64_t ldc, int M,
                                int dsub, const float* pq_centroids, uint8_t* arena_codes,
                                hipStream_t stream);
// t2[row] = chain_k fmaf(r^_k, fmaf(2, c_k, r^_k), acc) over k = 0..d-1: r^ = decoded PQ residual of the row,
// c = centroid of its list.  The list-dependent term of the IVFPQ L2 distance
// (faiss/impl/pq_code_distance/IVFPQ_QueryTables.cpp:126-192, term 2), kept per vector.
// rows variant: the n freshly appended rows dest[i] (list labels[i]); lists variant: every row of every list.
void launch_ivfpq_t2_rows(const uint8_t* arena_codes, const int64_t* labels, const int64_t* dest, int n,
                          const float* centroids, int64_t ldc, int M, int dsub, const float* pq_centroids, float* t2,
                          hipStream_t stream);
void launch_ivfpq_t2_lists(const uint8_t* arena_codes, const int64_t* list_start, const uint32_t* list_len, int nlist,
                           const float* centroids, int64_t ldc, int M, int dsub, const float* pq_centroids, float* t2,
                           hipStream_t stream);
// layout translation (copy_lists / get_list_codes): plain [row][M] codes <-> rotated block layout.
// pack: list l's len[l] codes start at plain row src_start[l]; unpack: one list
void launch_ivfpq_pack_lists(const uint8_t* plain, const int64_t* src_start, const int64_t* list_start,
                             const uint32_t* list_len, int nlist, int M, uint8_t* arena_codes, hipStream_t stream);
void launch_ivfpq_unpack_list(const uint8_t* arena_codes, int64_t first_row, uint32_t len, int M, uint8_t* plain,
                              hipStream_t stream);
// pq [M][256][dsub] -> pq_t [256][M][dsub] (the order the scan kernels build their [256][M] lookup table in)
void launch_pq_transpose(const float* pq, int M, int dsub, float* pq_t, hipStream_t stream);
// dst[dest[i]] = src[i]  (dest < 0 skipped)
void launch_scatter_i64(const int64_t* src, const int64_t* dest, int64_t n, int64_t* dst,
                        hipStream_t stream);
// dst[dest[i]] = labels[i] << 32 | (dest[i] - list_start[labels[i]])  (INDICES_IVF labels; dest / labels < 0 skipped)
void launch_ivf_pair_ids(const int64_t* labels, const int64_t* dest, int64_t n, const int64_t* list_start, int64_t* dst,
                         hipStream_t stream);
// out[i] = x[i] - centroids[labels[i]]  (faiss/gpu/impl/VectorResidual.cu:26)
void launch_residual(const float* x, int64_t ldx, int64_t n, int d, const int64_t* labels,
                     const float* centroids, int64_t ldc, float* out, int64_t ldo, hipStream_t stream);

} // namespace faiss_amd
