// Internal declarations shared by the translation units of libcellmap_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "cellmap.h"

// Diagnostics / test knobs (CELLMAP_* environment variables): read ONCE per cm_create (and by
// cm_reload_knobs) into this process-wide record -- nothing on a call path reads the environment.
struct cm_knobs {
    float seed_widen = 2.0f;     // CELLMAP_SEED_WIDEN: sample-scan seeds of a more precise level are widened by this many plain hi*hi error bounds
    int prune_max = 70;          // CELLMAP_PRUNE_MAX: largest share (%) of the (block, tile) pairs a single-pass pruned stage may visit (its gate: 10 below)
    int prune_max_sub = 60;      // CELLMAP_PRUNE_MAX_SUB: ... and pass A + B of a seeded subset stage
    int prune_min_nq = 16384;    // CELLMAP_PRUNE_MIN_NQ: fewest queries of a search whose main stage may be pruned / bucket-centred
    int ivf_submin = 32768;      // CELLMAP_IVF_SUBMIN: smallest set of flagged queries that is regrouped by bucket and pruned (42 k queries at sigma 0.1: 35 -> 20 ms pruned)
    int fb_max = 64;             // CELLMAP_FB_MAX: at most this many uncertified queries (all with a finite bound) go straight to the exact scan
    int dbg = 0;                 // CELLMAP_DBG: in-kernel timing modes / counters of the filter kernels
    int filter_products = 0;     // CELLMAP_FILTER_PRODUCTS: pin the filter level (no escalation)
    int r16 = 0;                 // CELLMAP_R16: 1 = the 128-queries-per-wave stage at K = 64 runs on the 16x16x32 re-tile (knn_cand_f16r) instead of the 32x32x16 kernel
                                 // (knn_cand_f16q); measured level to 1 % slower in production form (profiles/r04_mfma_shape_probe.md), so off by default
    int q4 = 1;                  // CELLMAP_Q4: 0 never / 1 by size / 2 wherever instantiated: the four-accumulator filter
    int slack = -1;              // CELLMAP_SLACK: candidate slack override
    int phase1 = 0;              // CELLMAP_PHASE1: 1 f32, 2 f16, 3 direct
    int spgemm_ct = 0;           // CELLMAP_SPGEMM_CT: columns per accumulator pass of the sparse product (tests)
    int spgemm_dbg = 0;          // CELLMAP_SPGEMM_DBG: timing experiments
    int a3_gather = 0;           // CELLMAP_A3_GATHER: 1 = entry-per-lane gather of the wave-sliced sparse product (A/B against the 16-byte one)
    long long spmm_budget_mb = -1, spgemm_budget_mb = -1;   // CELLMAP_SPMM_BUDGET_MB / CELLMAP_SPGEMM_BUDGET_MB: device budget of the streaming host tiers
    long long pool_mb = -1;      // CELLMAP_POOL_MB: cap of the per-handle device allocation pool
    bool dbg_ws = false, weights_lds = false, lowd_brute = false, no_sort = false, no_chain = false, no_centre = false, no_ivf = false,
         no_hint = false, no_perm = false, no_seed = false, no_subseed = false, no_skip3 = false,
         multi_staged = false,       // CELLMAP_MULTI_STAGED: multi-device graphs exchange their lists through host memory (the route of platforms without peer access)
         multi_ref_shard = false,    // CELLMAP_MULTI_REF_SHARD: a multi-device graph shards the REFERENCE rows (exchange + merge) even when the reference fits one device
         multi_check_fail = false,   // CELLMAP_MULTI_CHECK_FAIL: the exchange self-check of a multi-device handle reports a mismatch (tests the fall-back)
         no_pretest = false,     // CELLMAP_NO_PRETEST: skip the structure pre-test of the staged search (always sort, gate and pilot)
         no_preload = false;     // CELLMAP_NO_PRELOAD: cm_create does not load the hot path's code objects ahead of the first call
    int xfer_mode = 0;           // CELLMAP_XFER: 0 / unset = plain hipMemcpyAsync, "d2h" = large downloads through the pinned staging ring, 1 = both directions
    int xfer_threads = 0, xfer_slot_kb = 0, xfer_min_kb = 0;   // CELLMAP_XFER_THREADS / _SLOT_KB / _MIN_KB: staging ring geometry and the size from which it is used (0 = default: 8 threads, 4 MB slots, 8 MB)
};
const cm_knobs &cm_get_knobs();

#define CM_N_STAGE_EVENTS 8
struct cm_handle_s {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string err;
    // grow-only device workspace in SEGMENTS, carved by a bump allocator that is reset at the start of every API call
    // (cm_ws_reset).  Segment 0 is what a call reserves on entry for its common route (cm_ws_reserve); a route that needs more
    // (bucket pruning, the centred level, the f32 family's split lists ...) adds segments on demand inside cm_ws_alloc --
    // pointers handed out earlier stay valid, nothing is freed before cm_destroy.  (Round 5: one worst-case block used to be
    // reserved up front, 16 GB at BASELINE config 2, of which the structure-free route touches 3.)
    struct ws_seg { char *p; size_t cap; size_t off; };
    std::vector<ws_seg> ws_segs;
    size_t ws_cap = 0;    // sum of the segments' capacities
    size_t ws_used = 0;   // bytes carved since the last reset
    size_t ws_peak = 0;   // ... largest value of that over the handle's life
    int64_t knn_stats[4] = {0, 0, 0, 0};
    int64_t knn_sub_splits = 0;   // cm_knn_stats out[5]: most reference splits a subset stage of the last search used
    int64_t knn_wide_retries = 0; // cm_knn_stats out[6]: stages repeated with ordinary splits after a wide split overflowed the re-rank's capacity
    int n_cu = 0;   // compute units of the device (lazily queried)
    const void *a3_index = nullptr;   // last payload index whose header this handle has read (A3, wave-sliced kernel)
    int a3_max_slice = 0;             // ... its longest (payload row, column bucket) slice
    long long a3_nnz = 0, a3_rows = 0;
    // staged search: the filter level that certified most queries on the previous call with this (d, k, metric); lets
    // repeated calls on similar data skip the pilot (a stale hint only costs time: the ladder still escalates)
    int staged_hint_level = 0, staged_hint_d = 0, staged_hint_k = 0, staged_hint_metric = 0;
    int64_t staged_long_nref = 0, staged_long_nq = 0;   // hints 3, 4, 6 and -5 hold for the problem size they were learnt on
    int staged_ivf_skip = 0;   // calls for which bucket pruning is not tried again after it did not pay ...
    int64_t staged_ivf_nref = 0, staged_ivf_nq = 0;   // ... on a problem of these sizes
    int32_t *d_nflag_pinned = nullptr;  // host-pinned mirror for the fallback counter (read lazily)
    // host tier of A3: device copies of the mapping matrix / payload / output indptr kept between cm_spmm_csr_nnz and the
    // matching cm_spmm_csr call (opaque: owned by cm_api.hip, freed by cm_destroy)
    void *a3_session = nullptr;
    // device allocation pool of the host tier: blocks released by a call are reused by the next one instead of going
    // through hipMalloc / hipFree every time (freed by cm_destroy; bounded, see pool_release in cm_api.hip)
    std::vector<std::pair<size_t, void *>> pool;
    size_t pool_bytes = 0;
    size_t pool_cap = 0;   // bytes the pool may keep: min(an eighth of THIS device's memory, 32 GB), or CELLMAP_POOL_MB (cm_create)
    void *xfer = nullptr;   // pinned staging ring for large host <-> device copies (opaque: cm_api.hip, created on first use)
    hipStream_t copy_stream = nullptr;   // D2H of finished output tiles while the next tile is computed (host tier)
    hipEvent_t tile_ev[2] = {nullptr, nullptr};
    // multi-device handle (cm_create_multi): the OTHER GPUs' contexts, owned by this handle.  The handle itself is the
    // first device's context, so every single-device entry point keeps working on it unchanged; the graph tier
    // (cm_rep_* / cm_graph_*) and cm_knn spread over all of them (cm_api.hip, "multi-device").
    std::vector<cm_handle_s *> peers;
    bool multi_checked = false;        // the peer-copy exchange of this handle has been checked against the host-staged route (first sharded graph)
    bool multi_route_staged = false;   // ... and failed: every later graph exchanges through host memory
    // allocator clock (cm_alloc_stats): wall time this handle has spent inside the device allocator, by kind --
    // 0 workspace hipMalloc, 1 pooled blocks (representations, results, payloads) hipMalloc, 2 hipFree, 3 pinned host memory
    double alloc_ms[4] = {0, 0, 0, 0};
    double alloc_bytes[4] = {0, 0, 0, 0};
    double alloc_calls[4] = {0, 0, 0, 0};
    double create_ms = 0, preload_ms = 0;   // wall time of cm_create as a whole / of its code-object preload
    // optional per-stage timing (cm_set_profiling): events recorded on h->stream around the stages of the last call
    bool profiling = false;
    hipEvent_t ev[CM_N_STAGE_EVENTS] = {};
    bool ev_set[CM_N_STAGE_EVENTS] = {};
};
// stage boundaries: 0 start | 1 after row prep | 2 after phase-1 MFMA filter | 3 after re-rank | 4 after fallback |
//                   5 after bandwidth statistics | 6 after weights/normalise | 7 after label vote
void cm_mark(cm_handle_t h, int which);

#define CM_FAIL(h, code, ...)                                   \
    do {                                                        \
        char _b[512];                                           \
        snprintf(_b, sizeof(_b), __VA_ARGS__);                  \
        if (h) (h)->err = _b;                                   \
        return (code);                                          \
    } while (0)

#define CM_HIP(h, expr)                                                                         \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) CM_FAIL(h, CM_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

#define CM_TRY(expr)                     \
    do {                                 \
        cm_status _s = (expr);           \
        if (_s != CM_OK) return _s;      \
    } while (0)

// ---- code-object preload: one empty launch per hot-path translation unit (cm_create) ----
void cm_preload_knn(hipStream_t st);
void cm_preload_q4(hipStream_t st);
void cm_preload_f16(hipStream_t st);
void cm_preload_r16(hipStream_t st);
void cm_preload_graph(hipStream_t st);

// ---- workspace ----
cm_status cm_ws_reserve(cm_handle_t h, size_t bytes);  // total capacity >= bytes (adds a segment; never moves or frees one)
void *cm_ws_alloc(cm_handle_t h, size_t bytes);         // 256-B aligned carve; adds a segment when none has room; nullptr = out of device memory
void cm_ws_reset(cm_handle_t h);                        // start of a call: every segment is free again (stream order protects the previous call's kernels)
static inline size_t cm_align(size_t b) { return (b + 255) & ~size_t(255); }

// ---- k-NN (knn_kernels.hip) ----
constexpr int CM_MERGE_NCMAX = 768;   // most (parts x k) entries the shard merge ranks per query (knn_merge's LDS lists)
size_t cm_knn_ws_bytes(int64_t n_ref, int64_t nq, int32_t d, int32_t k, int32_t dtype, int32_t metric = CM_METRIC_EUCLIDEAN);
size_t cm_knn_ws_bytes_all_routes(int64_t n_ref, int64_t nq, int32_t d, int32_t k, int32_t dtype, int32_t metric);
cm_status cm_knn_run(cm_handle_t h, const void *d_ref, int64_t n_ref, const void *d_qry, int64_t nq, int32_t d,
                     int32_t dtype, int32_t k, int32_t metric, int32_t form, int64_t idx_offset, int32_t flags,
                     void *d_out_dist, int64_t *d_out_idx);
cm_status cm_knn_merge_run(cm_handle_t h, const double *d_dist, const int64_t *d_idx, int32_t nparts, int64_t part_stride,
                           int64_t nq, int32_t k, int32_t sqrt_out, double *d_out_dist, int64_t *d_out_idx);

// ---- graph kernel + transfers (graph_kernels.hip) ----
cm_status cm_moments_run(cm_handle_t h, const double *d_dist, const int64_t *d_idx, int64_t n, const double *d_center,
                         double *d_out3);
cm_status cm_param_run(cm_handle_t h, const double *d_m3, int32_t kernel, double *d_param);
cm_status cm_widen_f32_run(cm_handle_t h, const float *d_in, int64_t n, double *d_out);   // float32 distances -> float64
cm_status cm_weights_run(cm_handle_t h, const double *d_dist, const int64_t *d_idx, int64_t nq, int32_t k,
                         int64_t n_ref, int32_t kernel, const double *d_param, double eps, int32_t *d_cols,
                         float *d_vals, int32_t *d_cnt, double *d_conn);
cm_status cm_ell_to_csr_run(cm_handle_t h, const int32_t *d_cols, const float *d_vals, const double *d_conn,
                            const int32_t *d_cnt, int64_t nq, int32_t k, int64_t *d_indptr, int32_t *d_out_cols,
                            float *d_out_vals, double *d_out_conn);
size_t cm_scan_ws_bytes(int64_t n);
cm_status cm_ingest_count_run(cm_handle_t h, const int64_t *d_indptr, const int32_t *d_indices, int64_t n, int include_self,
                              int *d_rowlen, int *d_kmax);
cm_status cm_ingest_fill_run(cm_handle_t h, const int64_t *d_indptr, const int32_t *d_indices, const double *d_data, int64_t n,
                             int include_self, int kmax, int64_t *d_out_idx, double *d_out_dist);
// rows longer than the in-LDS sorts of cm_ingest_fill_run hold (rank_kernels.hip: segmented radix sort)
constexpr int CM_INGEST_LDS_MAX = 12288;
size_t cm_ingest_huge_ws_bytes(int64_t n, int kmax);
cm_status cm_ingest_fill_huge_run(cm_handle_t h, const int64_t *d_indptr, const int32_t *d_indices, const double *d_data, int64_t n,
                                  int include_self, int kmax, void *d_ws, size_t ws_bytes, int64_t *d_out_idx, double *d_out_dist);
cm_status cm_expr_metrics_run(cm_handle_t h, const float *d_a, const float *d_b, int64_t n, int64_t G, const int32_t *d_group,
                              int n_groups, int method, float *d_out);
size_t cm_expr_metrics_ws_bytes(int64_t n, int64_t G);
// rank_kernels.hip: Spearman (Pearson of average ranks) per gene
cm_status cm_expr_spearman_run(cm_handle_t h, const float *d_a, const float *d_b, int64_t n, int64_t G, const int32_t *d_group,
                               int n_groups, float *d_out);
size_t cm_expr_spearman_ws_bytes(int64_t n, int64_t G);
cm_status cm_presence_run(cm_handle_t h, const double *d_dist, const int64_t *d_idx, int64_t nq, int32_t k,
                          const double *d_param, const int32_t *d_group, int32_t n_groups, double *d_out);
cm_status cm_csr_normalize_run(cm_handle_t h, const int64_t *d_indptr, const double *d_data, int64_t nq, float *d_out);
cm_status cm_vote_run(cm_handle_t h, const int64_t *d_indptr, const int32_t *d_cols, const float *d_vals, int64_t nq,
                      int32_t ell_k, const int32_t *d_cnt, const int32_t *d_codes, int32_t n_cat, int32_t *d_out_code,
                      float *d_out_conf);
cm_status cm_spmm_dense_run(cm_handle_t h, const int64_t *d_indptr, const int32_t *d_cols, const float *d_vals,
                            int64_t nq, const void *d_B, int64_t n_out, int32_t b_dtype, void *d_out);
cm_status cm_spmm_csr_count_run(cm_handle_t h, const int64_t *d_indptr, const int32_t *d_cols, const float *d_vals,
                                int64_t nq, const int64_t *d_bptr, const int32_t *d_bcols, const float *d_bvals,
                                int64_t n_out, int32_t *d_row_nnz, const int32_t *d_order = nullptr);
cm_status cm_spmm_csr_fill_run(cm_handle_t h, const int64_t *d_indptr, const int32_t *d_cols, const float *d_vals,
                               int64_t nq, const int64_t *d_bptr, const int32_t *d_bcols, const float *d_bvals,
                               int64_t n_out, const int64_t *d_out_indptr, int64_t out_base, int32_t *d_out_cols, float *d_out_vals,
                               const int32_t *d_order = nullptr);
bool cm_spmm_csr_single_pass_ok(int64_t n_out);
int cm_a3_bucket_width(int64_t n_out);
size_t cm_a3_chain_bytes(int64_t nq);   // scratch of the single-pass sparse product's chained scan
int64_t cm_payload_index_len_impl(int64_t n_ref, int64_t n_out);
cm_status cm_payload_index_header(cm_handle_t h, const int32_t *d_index, int *max_slice, long long *nnz);
cm_status cm_payload_index_run(cm_handle_t h, const int64_t *d_bptr, const int32_t *d_bcols, int64_t n_ref, int64_t n_out,
                               int32_t *d_index, int *d_flag);
cm_status cm_spmm_csr_sliced_run(cm_handle_t h, const int64_t *d_indptr, const int32_t *d_cols, const float *d_vals,
                                 int64_t nq, const int64_t *d_bptr, const int32_t *d_bcols, const float *d_bvals, int64_t n_ref,
                                 const int32_t *d_index, int64_t n_out, int64_t *d_out_indptr, int32_t *d_out_cols,
                                 float *d_out_vals, int64_t capacity, const int32_t *d_order, unsigned long long *d_chain,
                                 int32_t *d_out_rows);
cm_status cm_spmm_csr_fused_run(cm_handle_t h, const int64_t *d_indptr, const int32_t *d_cols, const float *d_vals,
                                int64_t nq, const int64_t *d_bptr, const int32_t *d_bcols, const float *d_bvals,
                                int64_t n_out, int64_t *d_out_indptr, int32_t *d_out_cols, float *d_out_vals, int64_t capacity,
                                const int32_t *d_order, unsigned long long *d_chain, int32_t *d_out_rows);
size_t cm_locality_order_ws_bytes(int64_t n_ref);
cm_status cm_locality_order_run(cm_handle_t h, const int64_t *d_indptr, const int32_t *d_cols, int64_t nq, int64_t n_ref, int32_t *d_order);
cm_status cm_exclusive_scan_i32_to_i64(cm_handle_t h, const int32_t *d_in, int64_t n, int64_t *d_out /* n+1 */);
