/*
 * cellmap.h -- C ABI of libcellmap_hip.so: the MI355X (gfx950) backend for CellMapper's
 * k-NN graph -> graph kernel -> neighbour-weighted aggregation hot path.
 *
 * This is the drop-in boundary.  Plain pointers and sizes only; no torch / numpy types.
 * Each entry point names the reference interface it replaces (file:line under
 * /root/reference/src/cellmapper/).  INTEGRATION.md shows the ctypes binding a maintainer of
 * the reference would add.
 *
 * Two tiers:
 *   host tier   (cm_knn, cm_mapping_matrix, cm_label_vote, cm_spmm_dense, cm_spmm_csr, ...):
 *               caller-owned HOST buffers in/out (row-major, C-contiguous, no strides), exactly what
 *               the reference hands its k-NN backends (np.ascontiguousarray, cellmapper.py:250) and
 *               gets back (newly allocated numpy arrays).  Synchronous.
 *   device tier (cm_*_dev): caller-owned DEVICE buffers on the handle's GPU, asynchronous on the
 *               handle's stream (cm_set_stream).  Used by bench.py and by the multi-GPU driver,
 *               which keeps device memory / streams / RCCL in torch.
 *
 * Conventions: every function returns cm_status (0 = ok); cm_last_error(handle) gives the message.
 * One handle = one caller thread at a time; several handles per process are fine.
 * Index arrays follow the reference: neighbour indices int64, -1 = padded / invalid entry, distances
 * +inf for padded entries (utils.py:178-179); CSR index arrays int32 (scipy's choice below 2^31 nnz),
 * indptr int64.
 */
#ifndef CELLMAP_H
#define CELLMAP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cm_handle_s *cm_handle_t;

typedef enum {
    CM_OK = 0,
    CM_E_INVALID = 1,   /* bad argument (the Python layer raises ValueError) */
    CM_E_NOMEM = 2,     /* host or device allocation failed */
    CM_E_HIP = 3,       /* HIP runtime error (message has the hipError string) */
    CM_E_NOFINITE = 4,  /* "No finite distances found in the neighborhood graph" (knn.py:191-192) */
    CM_E_UNSUPPORTED = 5,
    CM_E_INTERNAL = 6
} cm_status;

/* element type of an embedding / payload matrix */
#define CM_F32 0
#define CM_F64 1
/* metric (knn.py:343 `metric`, passed to sklearn.neighbors.NearestNeighbors, knn.py:429) */
#define CM_METRIC_EUCLIDEAN 0
#define CM_METRIC_COSINE 1
/* Metric pass-through of the reference (knn.py:343,429 hands `metric` to sklearn): the two other Minkowski-family metrics
 * sklearn's trees and brute force share.  Exact like the rest (float64 re-rank in sklearn's arithmetic -- for float32 data
 * brute force the coordinate difference is a float32 and the distance is rounded to float32 once, as DistanceMetric32
 * does -- certificate, fallback); the filter is the fp32 VALU direct-difference kernel up to d = 128 (two queries per lane up
 * to 64 floats per row, one beyond), so they are VALU-bound.  Wider rows (round 4) run the exact scan: every pair value in
 * float64 by the same arithmetic, a running top-k per query (knn_exact_scan) -- no limit on d. */
#define CM_METRIC_MANHATTAN 2
#define CM_METRIC_CHEBYSHEV 3
/* sklearn's "sqeuclidean": always brute force (no tree supports it), EuclideanArgKmin with squared=True -- the euclidean GEMM-form
 * search whose float64 ranking values are returned WITHOUT the final square root, whatever d is. */
#define CM_METRIC_SQEUCLIDEAN 4
/* sklearn hands "correlation" to scipy.spatial.distance.cdist (brute force): rows centred in float64 (x - mean(x)), then the
 * cosine distance of the centred rows, float64 out whatever the input type.  Searched as the COSINE metric over float64
 * centred copies of both sides (workspace: 8 (n_ref + nq) d bytes more).  Not a bit-for-bit restatement: scipy's C loop
 * and numpy's pairwise mean round differently in the last bits (distances agree to ~1e-15; neighbours equal except inside
 * ties at that level) -- tests/parity.py states the tolerance.  A constant row has no correlation distance (scipy: nan);
 * here it behaves like a zero row under cosine (distance 1 to everything). */
#define CM_METRIC_CORRELATION 5
/* Round 4: sklearn's own DistanceMetric implementations of "canberra" (sum_j |x_j - y_j| / (|x_j| + |y_j|), a 0 / 0 term counts 0)
 * and "braycurtis" (sum_j |x_j - y_j| / sum_j (|x_j| + |y_j|), 0 when the denominator is 0): ArgKmin brute force at d > 15
 * (float32 data: the coordinate difference is a float32, everything else float64, the RESULT rounded to float32 once:
 * DistanceMetric32), ball_tree on float64 copies below.  Same exact machinery as manhattan / chebyshev (fp32 VALU direct
 * filter up to d = 128, the exact scan beyond); inputs holding a non-zero magnitude below 2^-100 (which the filter's float32
 * image would flush, and which its reciprocal cannot divide by) are screened and take the exact scan too.  Goldens through the reference pin the arithmetic (tests/golden/ *canberra*, *braycurtis*). */
#define CM_METRIC_CANBERRA 6
#define CM_METRIC_BRAYCURTIS 7
/* Round 4 (late): the counting metrics sklearn accepts without parameters.  sklearn keeps them away from ArgKmin: brute force
 * (d > 15) is scipy.spatial.distance.cdist on the data -- converted to boolean (x != 0, DataConversionWarning upstream) for all
 * but hamming -- and below that the ball tree with sklearn's DistanceMetric classes; both are ratios of the same integer
 * counts in float64, so one restatement serves both.  With n = d, ntt / ntf / nft / nff the coordinate counts of
 * (true, true) ... (false, false) and R = ntf + nft:
 *   hamming #(x_j != y_j) / n (on the values)     jaccard R / (ntt + R), 0 when ntt + R = 0     dice R / (2 ntt + R)
 *   rogerstanimoto = sokalmichener 2R / (ntt + nff + 2R)     russellrao (n - ntt) / n     sokalsneath 2R / (ntt + 2R)
 *   yule 2 ntf nft / (ntt nff + ntf nft), 0 when ntf nft = 0 (scipy only: brute force at every d)
 * dice and sokalsneath are 0 / 0 = NaN between two all-false rows, upstream and here (such a pair ranks last).  No filter:
 * the exact scan (knn_exact_scan) at every width; float64 distances out.  Neighbours with equal distances are ordered by
 * index here, arbitrarily upstream (argpartition / tree order): parity is on the distances and on tie groups. */
#define CM_METRIC_HAMMING 8
#define CM_METRIC_JACCARD 9
#define CM_METRIC_DICE 10
#define CM_METRIC_ROGERSTANIMOTO 11
#define CM_METRIC_RUSSELLRAO 12
#define CM_METRIC_SOKALMICHENER 13
#define CM_METRIC_SOKALSNEATH 14
#define CM_METRIC_YULE 15
/* Round 5: the last two metric names the reference can hand to sklearn without parameters.  Both run the exact scan; both are
 * compared with the reference at a STATED tolerance, not bit for bit (tests/parity.py): their upstream values come out of libm /
 * numpy transcendental functions and of BLAS products, whose last bits no device restatement reproduces.
 *   haversine     rows are (latitude, longitude) in radians, d = 2 only (else CM_E_INVALID: sklearn's ValueError):
 *                 2 asin(sqrt(sin^2((x0 - y0) / 2) + cos(x0) cos(y0) sin^2((x1 - y1) / 2))) -- sklearn's HaversineDistance.dist in float64
 *                 on the ball tree (k < n_ref / 2); ArgKmin brute force otherwise, which ranks by the bracket (rounded to float32, the
 *                 difference taken in float32, for float32 data: HaversineDistance32.rdist) and converts it at the end; float64 out.
 *   nan_euclidean sklearn.metrics.pairwise.nan_euclidean_distances (brute force at every width): NaN marks a missing coordinate;
 *                 sqrt(d / #present * sum over the coordinates present in BOTH rows of (x_j - y_j)^2), NaN when no coordinate is shared
 *                 (such a pair ranks last); float32 out for float32 data.  The only metric whose inputs may hold NaN: upload them
 *                 with cm_rep_create_ex(..., CM_REP_ALLOW_NAN) (cm_knn allows them by itself under this metric); infinities are
 *                 refused as everywhere. */
#define CM_METRIC_HAVERSINE 16
#define CM_METRIC_NAN_EUCLIDEAN 17
/* which exact float64 formula the returned distances follow (see DESIGN.md "K0"):
 *   AUTO   = what sklearn's algorithm="auto" would pick: GEMM form if d > 15 or k >= n_ref/2, else DIRECT
 *   GEMM   = sklearn brute / EuclideanArgKmin: (|q|^2 + (-2 q.y)) + |y|^2, clamp, sqrt
 *   DIRECT = sklearn kd_tree: sqrt(sum (q_j - y_j)^2)                                                   */
#define CM_FORM_AUTO 0
#define CM_FORM_GEMM 1
#define CM_FORM_DIRECT 2
/* graph kernels (knn.py:166-226; `mapping_method` of CellMapper.compute_mapping_matrix, cellmapper.py:253-305) */
#define CM_KERNEL_GAUSSIAN 0
#define CM_KERNEL_SCARCHES 1
#define CM_KERNEL_INVERSE_DISTANCE 2
#define CM_KERNEL_EQUAL 3
#define CM_KERNEL_IDENTITY 4 /* weight = the distance itself: NeighborsResults.knn_graph_distances (knn.py:113-132) */
/* cm_knn_dev flags */
#define CM_KNN_SQUARED 1      /* return squared (euclidean) distances; used by the sharded merge */
#define CM_KNN_FORCE_FALLBACK 2 /* testing: send every query through the exact float64 fallback */
#define CM_KNN_F64_OUT 4      /* cosine on float32 input: write float64 distances (the values the search ranked by) instead
                                 of sklearn's float32; the sharded merge needs them to reproduce the single-GPU order */

int cm_version(void);
int cm_device_count(void);

/* Create / destroy a backend context bound to one GPU (replaces faiss.StandardGpuResources /
 * index_cpu_to_gpu(res, 0, ...) bookkeeping of knn.py:410-413). */
cm_status cm_create(int device, cm_handle_t *out);
/* The same over SEVERAL GPUs of one node (SURVEY.md 8b `cm_create(devices[], n)`; the reference is pinned to one device,
 * knn.py:412,420): one stream + workspace per listed device, owned by the handle.  The graph tier (cm_rep_create,
 * cm_graph_knn, cm_graph_mapping, cm_graph_label_vote, cm_graph_spmm_dense, cm_graph_spmm_csr*) and cm_knn then use all of
 * them from ONE call of ONE caller thread: the representations are uploaded to every device, the REFERENCE ROWS are
 * partitioned (device p searches all queries against rows [lo_p, hi_p) with global indices), the per-shard lists are
 * exchanged by query slice with peer copies (hipMemcpyPeerAsync: xGMI between GPUs) and merged by (distance, index) with the
 * kernel of cm_knn_merge_strided_dev -- exact, bit-identical to one device -- after which device p owns query slice p of the
 * graph, of the mapping matrix (global bandwidth from per-device moments summed on the host) and of every transfer
 * (payloads replicated, outputs written to the caller's arrays by row range).  The library runs one host thread per device
 * inside the call.  Everything else (host tier, device tier) runs on devices[0].  n = 1 is cm_create(devices[0]).  A device
 * may be listed more than once (tests of the choreography on a one-GPU box).  Small problems (fewer than 4 096 reference
 * rows per device or fewer than 1 024 queries) and cosine on float32 input (whose float32 distances are a property of the
 * one-device arithmetic) run on devices[0] alone. */
cm_status cm_create_multi(const int *devices, int n, cm_handle_t *out);
cm_status cm_destroy(cm_handle_t h);
/* Housekeeping (no reference counterpart).  cm_pool_trim returns the handle's cached device blocks (the allocation pool of
 * the host / graph tiers, capped at min(1/8 of the device memory, 32 GB) or CELLMAP_POOL_MB) to the HIP allocator, so that
 * other users of the GPU (torch, another handle) see the memory again; it also drops a pending cm_spmm_csr_nnz session.
 * cm_reload_knobs re-reads the CELLMAP_* diagnostics / test variables: the library reads them in cm_create only, never on
 * a call path. */
cm_status cm_pool_trim(cm_handle_t h);
cm_status cm_reload_knobs(void);
/* Allocation ahead of the first call (no reference counterpart: the reference allocates nothing it could pre-size).
 * cm_workspace_bytes = device workspace the k-NN -> kernel -> vote pipeline needs for a problem of this shape (the figure
 * cm_knn_dev / cm_map_labels_dev / cm_graph_knn reserve on entry; -1 on a bad shape); cm_reserve_workspace grows the handle's
 * workspace to at least `bytes` NOW, so that the first call of a problem finds it (the workspace is grow-only, in segments --
 * a route that needs more than was reserved adds a segment inside the call -- and is never returned before cm_destroy).  cm_alloc_stats: wall time this handle has spent inside the device allocator since cm_create --
 * out[0..3] milliseconds, out[4..7] bytes, out[8..11] calls, each by kind (0 workspace hipMalloc, 1 pooled blocks of the
 * host / graph tiers, 2 hipFree, 3 pinned host memory); out[12] = milliseconds cm_create took, out[13] of which the
 * code-object preload; out[14] = workspace capacity (all segments), out[15] = most workspace bytes any one call has carved.  (bench.py: first_call_breakdown_ms) */
int64_t cm_workspace_bytes(int64_t n_ref, int64_t nq, int32_t d, int32_t dtype, int32_t k, int32_t metric);
/* ... plus what the routes of clustered embeddings add on demand (bucket pruning's lists, the bucket-centred level's side values:
 * 7 GB more at 1.5 M x 1.5 M x 50): reserve this to keep even the first call on clustered data free of allocations. */
int64_t cm_workspace_bytes_all_routes(int64_t n_ref, int64_t nq, int32_t d, int32_t dtype, int32_t k, int32_t metric);
cm_status cm_reserve_workspace(cm_handle_t h, int64_t bytes);
cm_status cm_alloc_stats(cm_handle_t h, double out[16]);
const char *cm_last_error(cm_handle_t h); /* h may be NULL: message of the last failed cm_create */
cm_status cm_set_stream(cm_handle_t h, void *hip_stream); /* NULL = the handle's own stream */
cm_status cm_synchronize(cm_handle_t h);
/* Per-stage device timing of the LAST device-tier call, from HIP events recorded on the handle's stream (used by
 * bench.py for the roofline line).  out_ms[i] = time between stage boundary i-1 and i:
 * 1 row prep | 2 phase-1 MFMA distance+filter kernel | 3 float64 re-rank | 4 exact fallback |
 * 5 bandwidth statistics | 6 weights+normalise | 7 label vote.  cm_stage_times synchronises the stream. */
cm_status cm_set_profiling(cm_handle_t h, int32_t on);
cm_status cm_stage_times(cm_handle_t h, float out_ms[8]);
/* counters of the last cm_knn / cm_knn_dev call: out[0] = queries sent to the exact fallback,
 * out[1] = candidate-list size K', out[2] = reference splits + 100 * (filter level the full set started at)
 *          + 10000 * (filter kernel of the main stage: 1 two-accumulator split-fp16 (512-query blocks), 2 four-accumulator (1 024-query
 *          blocks), 3 its 16x16x32 re-tile, 4 the K = 128 three-product kernel, 5 direct VALU; 0 = grid / other families),
 * out[3] = filter family (1 f32-input MFMA, 2 split-fp16 MFMA K <= 64, 3 direct VALU / grid, 4 split-fp16 MFMA K = 128)
 *          + 100 * (per cent of the (query block, reference tile) pairs the main stage visited; 0 = no bucket pruning)
 *          + 100000 * (form of the pruned main stage: 1 = one pass from sample-scan seeds, 2 = pass A + pass B; 0 = not pruned),
 * out[4] = 1 if the exact fallback overflowed (more than 16 384 queries of one call each had more than 4 096 rows
 *          within their bound -- up to that many take a selection pass and are exact; beyond it lists are truncated:
 *          cm_synchronize and the host tier turn this into CM_E_UNSUPPORTED, device-tier callers that synchronise the
 *          stream themselves must check it), out[5] = most reference splits a subset stage (uncertified queries on a
 *          later filter level) used -- beyond 16: a small seeded subset split to fill the chip; out[6] = such stages repeated with ordinary
 *          splits because their lists overflowed the re-rank's capacity (very tight clusters); out[7] reserved (0).
 *          Read after the stream is idle. */
cm_status cm_knn_stats(cm_handle_t h, int64_t out[8]);

/* ---------------------------------------------------------------- K0: exact k-NN ------------------
 * Replaces Neighbors.compute_neighbors' backend call (knn.py:428-440, sklearn branch;
 * faiss / cuML branches knn.py:379-426): for every query row the k nearest reference rows, rows sorted
 * ascending by (distance, index), self included when present.
 *   ref [n_ref, d], qry [nq, d]  dtype CM_F32 / CM_F64 (both the same)
 *   out_dist: float64 [nq, k] for euclidean; for cosine the input dtype (sklearn keeps float32), [nq, k]
 *   out_idx : int64 [nq, k]
 * Errors: k > n_ref -> CM_E_INVALID (sklearn raises ValueError, neighbors/_base.py:848-859).  Any k <= n_ref is served (round 5: beyond
 * ~4 200 the exact scan runs in passes of 4 096 ranks, each pass ranking only the pairs after the previous pass's last entry).
 * No limit on d (d <= 61: split-fp16 MFMA filter K = 64; <= 128: K = 128; beyond: f32 MFMA chunked over d -- past 2047 columns
 * the re-rank reads its query rows from global memory instead of LDS).  k beyond the filters' candidate buffers (about 190
 * on the split-fp16 paths, 350 on the f32 path): the exact scan, as for the term-wise metrics beyond 128 columns. */
cm_status cm_knn(cm_handle_t h, const void *ref, int64_t n_ref, const void *qry, int64_t nq, int32_t d,
                 int32_t dtype, int32_t k, int32_t metric, int32_t form, void *out_dist, int64_t *out_idx);

/* Device tier of the same.  idx_offset is added to every returned index (reference shard offset,
 * SURVEY.md 8e).  out_dist is float64 [nq,k] for euclidean (squared if CM_KNN_SQUARED), input dtype for cosine
 * (float64 with CM_KNN_F64_OUT). */
cm_status cm_knn_dev(cm_handle_t h, const void *d_ref, int64_t n_ref, const void *d_qry, int64_t nq, int32_t d,
                     int32_t dtype, int32_t k, int32_t metric, int32_t form, int64_t idx_offset, int32_t flags,
                     void *d_out_dist, int64_t *d_out_idx);

/* k-way merge of per-shard results (the RCCL exchange delivers nparts blocks of [nq,k]):
 * d_dist [nparts, nq, k] float64 (squared distances if sqrt_out != 0), d_idx [nparts, nq, k] int64.
 * Keeps the k smallest by (distance, index) -- the same tie rule as a single-GPU search. */
cm_status cm_knn_merge_dev(cm_handle_t h, const double *d_dist, const int64_t *d_idx, int32_t nparts, int64_t nq,
                           int32_t k, int32_t sqrt_out, double *d_out_dist, int64_t *d_out_idx);
/* The same on the PACKED exchange buffer of the multi-GPU path (one collective carries distances and indices): part p's
 * rows start at d_dist + p * part_stride and d_idx + p * part_stride (strides in 8-byte elements, >= nq*k), e.g. a
 * received all-to-all buffer [nparts][2][nq][k] has d_idx = d_dist + nq*k and part_stride = 2*nq*k. */
cm_status cm_knn_merge_strided_dev(cm_handle_t h, const double *d_dist, const int64_t *d_idx, int32_t nparts,
                                   int64_t part_stride, int64_t nq, int32_t k, int32_t sqrt_out, double *d_out_dist,
                                   int64_t *d_out_idx);

/* ------------------------------------------- K1-K3: graph kernel -> row-normalised mapping matrix ---
 * Replaces NeighborsResults.knn_graph_connectivities (knn.py:134-164, _compute_kernel_values :166-226,
 * _create_sparse_matrix :79-111) followed by CellMapper._validate_and_normalize_mapping_matrix
 * (cellmapper.py:99-137): valid = (idx != -1) & isfinite(dist); kernel weights with a GLOBAL bandwidth;
 * canonical CSR (columns ascending, duplicates summed); float64 row sums; multiply by reciprocal; float32.
 *   dist float64 [nq,k], idx int64 [nq,k]
 *   out_indptr int64 [nq+1], out_cols int32 [<= nq*k], out_vals float32 [<= nq*k] (normalised),
 *   out_conn float64 [<= nq*k] or NULL (un-normalised connectivities = knn_graph_connectivities().data)
 *   *out_nnz, *out_param (sigma for gaussian, (2/std)^2 for scarches; may be NULL) */
cm_status cm_mapping_matrix(cm_handle_t h, const double *dist, const int64_t *idx, int64_t nq, int32_t k,
                            int64_t n_ref, int32_t kernel, double epsilon, int64_t *out_indptr, int32_t *out_cols,
                            float *out_vals, double *out_conn, int64_t *out_nnz, double *out_param);

/* K3 alone, for a caller-supplied matrix (CellMapper.mapping_matrix setter, cellmapper.py:83-137; also the
 * jaccard / hnoca matrices): canonical CSR with float64 data -> row-normalised float32 data (zero rows unchanged). */
cm_status cm_csr_normalize(cm_handle_t h, const int64_t *indptr, const double *data, int64_t nq, float *out_vals);

/* Device tier, split so that the global bandwidth can be all-reduced across reference shards:
 *   cm_dist_moments_dev: d_out[0] += sum(d - c), d_out[1] += sum((d - c)^2), d_out[2] += count over valid
 *                        entries (c = *d_center or 0 when NULL).  d_out must be zeroed by the caller.
 *   cm_kernel_param_dev: d_param[0] = sigma (gaussian: m[0]/m[2]) or s = (2/sqrt(m[1]/m[2]))^2 (scarches).
 *   cm_weights_dev:      padded "ELL" form of the mapping matrix: d_cols/d_vals [nq,k] with each row's valid
 *                        entries first, columns ascending; d_cnt[nq] valid entries per row;
 *                        d_conn [nq,k] float64 or NULL. */
cm_status cm_dist_moments_dev(cm_handle_t h, const double *d_dist, const int64_t *d_idx, int64_t n,
                              const double *d_center, double *d_out3);
cm_status cm_kernel_param_dev(cm_handle_t h, const double *d_moments3, int32_t kernel, double *d_param);
cm_status cm_weights_dev(cm_handle_t h, const double *d_dist, const int64_t *d_idx, int64_t nq, int32_t k,
                         int64_t n_ref, int32_t kernel, const double *d_param, double epsilon, int32_t *d_cols,
                         float *d_vals, int32_t *d_cnt, double *d_conn);
/* ELL -> CSR: d_indptr int64 [nq+1] (exclusive scan of d_cnt); compacts cols / vals when rows are ragged. */
cm_status cm_ell_to_csr_dev(cm_handle_t h, const int32_t *d_cols, const float *d_vals, const int32_t *d_cnt,
                            int64_t nq, int32_t k, int64_t *d_indptr, int32_t *d_out_cols, float *d_out_vals);

/* ---------------------------------------------------------- A0: categorical label transfer ---------
 * Replaces CellMapper._map_obs_categorical (cellmapper.py:589-623): OneHotEncoder + (M @ onehot) +
 * sparse argmax / max, fused into a per-row weighted vote.  codes[n_ref] = category code of each
 * reference cell in sorted-unique-category order (what OneHotEncoder.categories_ yields).
 * float32 sums accumulated in ascending column order; ties -> lowest code; empty row -> (0, 0.0). */
cm_status cm_label_vote(cm_handle_t h, const int64_t *indptr, const int32_t *cols, const float *vals, int64_t nq,
                        const int32_t *codes, int64_t n_ref, int32_t n_cat, int32_t *out_code, float *out_conf);
cm_status cm_label_vote_dev(cm_handle_t h, const int64_t *d_indptr, const int32_t *d_cols, const float *d_vals,
                            int64_t nq, const int32_t *d_codes, int64_t n_ref, int32_t n_cat, int32_t *d_out_code,
                            float *d_out_conf);

/* ------------------------------------- A0n / A1 / A2: mapping matrix x dense reference payload -------
 * Replaces `self.mapping_matrix @ dense` at cellmapper.py:338 (map_obsm), :373 (map_layers, dense .X /
 * layer) and :628 (_map_obs_numerical, n_out = 1).  B [n_ref, n_out] row-major, b_dtype CM_F32 / CM_F64;
 * out [nq, n_out] has B's dtype (scipy upcast(float32, B.dtype)); accumulation in that dtype in
 * ascending column order, multiply and add rounded separately (scipy csr_matvecs). */
cm_status cm_spmm_dense(cm_handle_t h, const int64_t *indptr, const int32_t *cols, const float *vals, int64_t nq,
                        const void *B, int64_t n_ref, int64_t n_out, int32_t b_dtype, void *out);
cm_status cm_spmm_dense_dev(cm_handle_t h, const int64_t *d_indptr, const int32_t *d_cols, const float *d_vals,
                            int64_t nq, const void *d_B, int64_t n_ref, int64_t n_out, int32_t b_dtype, void *d_out);

/* ------------------------------------------ A3: mapping matrix x sparse reference payload ------------
 * Replaces `self.mapping_matrix @ csr` at cellmapper.py:373 (sparse .X / layer; scipy csr_matmat).
 * Two calls, like scipy's csr_matmat_maxnnz / csr_matmat: cm_spmm_csr_nnz computes out_indptr[nq+1]
 * (exact zeros dropped, as csr_matmat does); the caller allocates out_cols / out_vals of out_indptr[nq]
 * entries and calls cm_spmm_csr.  Output columns are ascending within a row. */
cm_status cm_spmm_csr_nnz(cm_handle_t h, const int64_t *indptr, const int32_t *cols, const float *vals, int64_t nq,
                          const int64_t *b_indptr, const int32_t *b_cols, const float *b_vals, int64_t n_ref,
                          int64_t n_out, int64_t *out_indptr);
cm_status cm_spmm_csr(cm_handle_t h, const int64_t *indptr, const int32_t *cols, const float *vals, int64_t nq,
                      const int64_t *b_indptr, const int32_t *b_cols, const float *b_vals, int64_t n_ref,
                      int64_t n_out, const int64_t *out_indptr, int32_t *out_cols, float *out_vals);
/* Host-tier streaming ("wide aggregation outputs"; cellmapper.py:372-373 materialises the whole
 * product): cm_spmm_csr_nnz keeps its device copies of both matrices and of out_indptr on the handle until the matching
 * cm_spmm_csr call (same pointers and sizes), so the pair uploads everything once; cm_spmm_csr and cm_spmm_dense produce
 * the output in query tiles (cm_spmm_dense additionally in column slabs of the payload) sized to the free device memory and
 * copy tile i to the caller's arrays on a second stream while tile i+1 is computed -- outputs larger than HBM are fine.
 * Any number of output columns: the on-chip accumulator of the sparse product covers 32 768 columns per pass.
 *
 * Device tier of the same (asynchronous on the handle's stream).  cm_spmm_csr_nnz_dev writes d_out_indptr [nq+1] (exclusive
 * scan of the per-row counts, starting at 0); cm_spmm_csr_dev fills d_out_cols / d_out_vals [d_out_indptr[nq]].  For a
 * query TILE pass d_indptr + q0 (its entries index the full d_cols / d_vals) and the tile's row count as nq.
 * d_row_order (may be NULL): output row r is the product row of mapping-matrix row d_row_order[r]; d_indptr is then the
 * FULL indptr and a tile is a slice of the ORDER (d_row_order + q0).  cm_locality_order_dev fills such an order (int32
 * [nq], a permutation of the rows): rows grouped by their smallest neighbour column, so consecutive rows gather mostly the
 * same payload rows and those stay in L2 (config 5: every reference row is gathered by ~k different queries).  The order
 * within a group is unspecified. */
cm_status cm_locality_order_dev(cm_handle_t h, const int64_t *d_indptr, const int32_t *d_cols, int64_t nq, int64_t n_ref,
                                int32_t *d_order);
cm_status cm_spmm_csr_nnz_dev(cm_handle_t h, const int64_t *d_indptr, const int32_t *d_cols, const float *d_vals, int64_t nq,
                              const int64_t *d_b_indptr, const int32_t *d_b_cols, const float *d_b_vals, int64_t n_ref,
                              int64_t n_out, int64_t *d_out_indptr, const int32_t *d_row_order);
cm_status cm_spmm_csr_dev(cm_handle_t h, const int64_t *d_indptr, const int32_t *d_cols, const float *d_vals, int64_t nq,
                          const int64_t *d_b_indptr, const int32_t *d_b_cols, const float *d_b_vals, int64_t n_ref,
                          int64_t n_out, const int64_t *d_out_indptr, int32_t *d_out_cols, float *d_out_vals,
                          const int32_t *d_row_order);
/* (A3, `M @ reference.X` for a sparse layer: /root/reference/src/cellmapper/model/cellmapper.py:372-373, scipy csr_matmat.)
 * Single-pass form for callers that own a tile buffer with room for the worst case (n_out <= 32 768 columns, else
 * CM_E_INVALID: use the pair above): counts, offsets and entries come from ONE gather of the payload rows -- the rows'
 * places in the output are found by a chained scan inside the kernel.  d_out_indptr [nq+1] is always written in full
 * (exact counts); the entries of rows that would end beyond `capacity` (entries available in d_out_cols / d_out_vals) are
 * NOT written: the caller compares d_out_indptr[nq] with its capacity.  Same result bits as the pair above.
 * d_out_rows == NULL: output row r is input row r (of the order); the chained scan makes a row wait for the rows in front of
 * it.  d_out_rows != NULL (int32 [nq], nq < 2^24): rows are emitted in COMPLETION order -- output row j is the product row of
 * input row d_out_rows[j] (a permutation of 0..nq-1 that differs from run to run; the content of a row does not), and no
 * workgroup waits for another: the fastest form, for consumers that address rows through d_out_rows. */
cm_status cm_spmm_csr_fused_dev(cm_handle_t h, const int64_t *d_indptr, const int32_t *d_cols, const float *d_vals, int64_t nq,
                                const int64_t *d_b_indptr, const int32_t *d_b_cols, const float *d_b_vals, int64_t n_ref,
                                int64_t n_out, int64_t capacity, int64_t *d_out_indptr, int32_t *d_out_cols, float *d_out_vals,
                                const int32_t *d_row_order, int32_t *d_out_rows, const int32_t *d_payload_index);
/* Payload index for cm_spmm_csr_fused_dev (d_payload_index, may be NULL there): for every payload row the positions where
 * its ascending columns cross the multiples of the bucket width (n_out / 8 rounded up to 64), int32
 * [cm_payload_index_len(n_ref, n_out)].  With it every wave of a workgroup owns one eighth of the output columns and takes
 * exactly its entries from each neighbour row: the adds of a column stay in neighbour order without the workgroup barrier
 * per neighbour -- same bits, ~1.4 x faster.  Build it once per payload (a few ms at 4 M x 900 non-zeros) and reuse it for
 * every query tile.  cm_payload_index_dev is SYNCHRONOUS: it returns CM_E_INVALID if a payload row is not canonical CSR
 * (strictly ascending columns inside [0, n_out)) -- the barrier kernel (NULL index) only needs distinct columns per row.
 * cm_payload_index_len: -1 if n_out is outside 1 .. 32 768. */
int64_t cm_payload_index_len(int64_t n_ref, int64_t n_out);
cm_status cm_payload_index_dev(cm_handle_t h, const int64_t *d_b_indptr, const int32_t *d_b_cols, int64_t n_ref, int64_t n_out,
                               int32_t *d_index);

/* ---------------------------------------------------------------- graph tier: device-resident staged flow ----
 * Upstream, compute_neighbors -> compute_mapping_matrix -> map_obs / map_obsm / map_layers is ONE in-process flow over
 * objects that stay in memory (cellmapper.py:426-491; Neighbors / NeighborsResults knn.py:14-77,269-465).  These objects
 * are its device-side counterpart: the representations, the k-NN result and the row-normalised mapping matrix stay in HBM
 * between the staged calls of the drop-in classes; host arrays are produced only when the caller reads them
 * (NeighborsResults.distances / .indices, CellMapper.mapping_matrix -> the *_fetch calls).  Host pointers in, host pointers
 * out, synchronous, like the host tier; arithmetic and results are those of cm_knn / cm_mapping_matrix / cm_label_vote /
 * cm_spmm_dense / cm_spmm_csr.
 *   cm_rep_t   = an embedding matrix uploaded once (xrep / yrep of Neighbors; the four-graph mode searches each twice)
 *   cm_graph_t = one k-NN graph (query rows x reference rows) and, after cm_graph_mapping, its mapping matrix */
typedef struct cm_rep_s *cm_rep_t;
typedef struct cm_graph_s *cm_graph_t;
cm_status cm_rep_create(cm_handle_t h, const void *X, int64_t n, int32_t d, int32_t dtype, cm_rep_t *out);
/* cm_rep_create with flags.  CM_REP_ALLOW_NAN: NaN entries are kept (missing coordinates of CM_METRIC_NAN_EUCLIDEAN: sklearn validates
 * with ensure_all_finite="allow-nan" under that metric only, neighbors/_base.py); a graph of any other metric over a representation
 * that holds a NaN fails with CM_E_INVALID like an upload without the flag.  Infinities are always refused. */
#define CM_REP_ALLOW_NAN 1
cm_status cm_rep_create_ex(cm_handle_t h, const void *X, int64_t n, int32_t d, int32_t dtype, int32_t flags, cm_rep_t *out);
cm_status cm_rep_destroy(cm_handle_t h, cm_rep_t r);
/* K0 (knn.py:428-440): same arguments and errors as cm_knn */
cm_status cm_graph_knn(cm_handle_t h, cm_rep_t ref, cm_rep_t qry, int32_t k, int32_t metric, int32_t form, cm_graph_t *out);
cm_status cm_graph_destroy(cm_handle_t h, cm_graph_t g);
/* out[0] = query rows, [1] = reference rows, [2] = k, [3] = metric, [4] = 1 if distances are float32 (cosine on float32
 * input), [5] = 1 once the mapping matrix exists, [6] = its nnz, [7] = 0 */
/* One part of a graph (multi-device handle: part p = the query slice device p owns; single device: part 0):
 * out = { device, first query row, query rows, first reference row searched on that device, reference rows searched,
 *         bytes of the reference representation resident on the device, bytes of the query representation resident,
 *         exchange route (0 peer copies, 1 staged through host memory) + 10 x self-check (0 not run for this graph, 1 passed,
 *         2 failed: the lists were re-exchanged through host memory) }.
 * Round 4: a representation is uploaded ONCE (to devices[0], cm_rep_create); device p > 0 receives reference rows
 * [lo_p, hi_p) and every query row by a peer copy when a graph first shards ("partition reference rows
 * contiguously across G GPUs"; the reference itself is pinned to one device, knn.py:412,420). */
cm_status cm_graph_part_info(cm_handle_t h, cm_graph_t g, int32_t part, int64_t out[8]);
cm_status cm_graph_info(cm_handle_t h, cm_graph_t g, int64_t out[8]);
/* distances float64 [nq,k] (float32 when info[4]), indices int64 [nq,k] */
cm_status cm_graph_knn_fetch(cm_handle_t h, cm_graph_t g, void *out_dist, int64_t *out_idx);
/* K1-K3 (knn.py:134-226,79-111; cellmapper.py:99-137) on the resident graph: same arithmetic as cm_mapping_matrix */
cm_status cm_graph_mapping(cm_handle_t h, cm_graph_t g, int32_t kernel, double epsilon, int64_t *out_nnz, double *out_param);
cm_status cm_graph_mapping_fetch(cm_handle_t h, cm_graph_t g, int64_t *out_indptr, int32_t *out_cols, float *out_vals);
/* A0 / A0n / A1 / A2 / A3 with the resident mapping matrix (cellmapper.py:589-637, :338, :372-373) */
cm_status cm_graph_label_vote(cm_handle_t h, cm_graph_t g, const int32_t *codes, int64_t n_ref, int32_t n_cat, int32_t *out_code,
                              float *out_conf);
cm_status cm_graph_spmm_dense(cm_handle_t h, cm_graph_t g, const void *B, int64_t n_ref, int64_t n_out, int32_t b_dtype, void *out);
cm_status cm_graph_spmm_csr_nnz(cm_handle_t h, cm_graph_t g, const int64_t *b_indptr, const int32_t *b_cols, const float *b_vals,
                                int64_t n_ref, int64_t n_out, int64_t *out_indptr);
cm_status cm_graph_spmm_csr(cm_handle_t h, cm_graph_t g, const int64_t *b_indptr, const int32_t *b_cols, const float *b_vals,
                            int64_t n_ref, int64_t n_out, const int64_t *out_indptr, int32_t *out_cols, float *out_vals);

/* ------------------------------------- shared-neighbour counts: jaccard / hnoca mapping methods ------
 * Replaces `(yx @ xx.T) + (yy @ xy.T)` on the 0/1 adjacency matrices of the four k-NN graphs and the following
 * transform (cellmapper.py:287-301; adjacencies: knn.py:228-266,467-483).  yx, yy: int64 [nq,k]; xx, xy: int64 [nr,k]
 * neighbour indices (-1 = padded).  J[i,j] = |N_yx(i) & N_xx(j)| + |N_yy(i) & N_xy(j)|.
 * transform: 0 = raw counts, 1 = jaccard  J/(4k - J), 2 = hnoca  (J/(2k - J))^2.   Output: canonical CSR, float64
 * values (feed cm_csr_normalize for the mapping matrix).  Call once with out_cols = NULL to obtain out_indptr and
 * *out_nnz, allocate, call again with capacity >= nnz. */
cm_status cm_shared_neighbors(cm_handle_t h, const int64_t *yx, const int64_t *xx, const int64_t *yy, const int64_t *xy,
                              int64_t nq, int64_t nr, int32_t k, int32_t transform, int64_t *out_indptr,
                              int32_t *out_cols, double *out_vals, int64_t capacity, int64_t *out_nnz);

/* ------------------------------------------------- presence scores -----------
 * Replaces the column sums in EvaluationMixin.estimate_presence_score (evaluate.py:426-480): raw score of reference
 * cell r = sum over query cells of the un-normalised gaussian connectivity (global sigma, as in K1), overall and per
 * query group.  group int32 [nq] in [0, n_groups) or NULL.  out float64 [n_ref, n_groups + 1]: column 0 = all queries. */
cm_status cm_presence_scores(cm_handle_t h, const double *dist, const int64_t *idx, int64_t nq, int32_t k, int64_t n_ref,
                             const int32_t *group, int32_t n_groups, double *out);

/* ---------------------------------------------------------------- fused device pipeline -------------
 * k-NN -> kernel -> normalise -> label vote with every intermediate resident in HBM (BASELINE config 2;
 * CellMapper.map(obs_keys=label), cellmapper.py:426-491).  Outputs (device): d_out_code int32 [nq],
 * d_out_conf float32 [nq]; optional d_out_dist float64 [nq,k] / d_out_idx int64 [nq,k] (may be NULL). */
cm_status cm_map_labels_dev(cm_handle_t h, const void *d_ref, int64_t n_ref, const void *d_qry, int64_t nq, int32_t d,
                            int32_t dtype, int32_t k, int32_t metric, int32_t kernel, double epsilon,
                            const int32_t *d_codes, int32_t n_cat, int32_t *d_out_code, float *d_out_conf,
                            double *d_out_dist, int64_t *d_out_idx);

/* ------------------------------------------------- precomputed-distance ingest ----
 * Replaces extract_neighbors_from_distances (utils.py:129-219; called from knn.py:296-337 via
 * CellMapper.load_precomputed_distances, cellmapper.py:493-532): square CSR distance matrix [n, n] (int64 indptr,
 * int32 column indices, float64 data) -> padded rows int64 indices / float64 distances [n, kmax] with -1 / inf padding.
 * include_self: -1 = None (keep rows as stored), 0 = False (drop diagonal entries), 1 = True (append (i, 0.0) to rows
 * without a diagonal entry).  Rows whose distances are non-decreasing keep their stored order, others are sorted by
 * distance (ties by stored position).  Call once with out_idx = out_dist = NULL to obtain *out_kmax (the longest row
 * after the include_self rule), then with buffers of kmax = *out_kmax columns.  Any row length (utils.py:129-219 sets no limit):
 * up to 1 023 entries one wave per row, up to 12 287 one workgroup per row (rank sort in LDS), beyond that a staged stable
 * segmented radix sort (round 4; device memory: 2 x n x kmax x 16 bytes). */
cm_status cm_ingest_distances(cm_handle_t h, const int64_t *indptr, const int32_t *indices, const double *data, int64_t n,
                              int32_t include_self, int32_t kmax, int64_t *out_idx, double *out_dist, int32_t *out_kmax);

/* ------------------------------------------------- expression-transfer evaluation --
 * Replaces the per-gene Python loop of EvaluationMixin.evaluate_expression_transfer (evaluate.py:236-323, helpers
 * :20-63): agreement between measured (orig) and imputed expression, both float32 row-major [n, n_genes], per gene over
 * all cells (output row 0) and over the cells of each group (rows 1..n_groups; group int32 [n] in [0, n_groups) or
 * NULL).  method: 0 = Pearson r (scipy.stats.pearsonr; NaN for a constant column), 1 = RMSE of the z-scored vectors
 * (std 0 -> 1), 2 = Jensen-Shannon distance base 10 of the clipped, normalised vectors (NaN if either sums to 0).
 * 3 = Spearman rho (scipy.stats.spearmanr, evaluate.py:276-277: Pearson r of the average ranks; NaN for a constant
 * column; device sort of every gene column).  out float32 [n_groups + 1, n_genes].  The _dev form takes device pointers (the imputed
 * matrix never has to leave the GPU). */
#define CM_EXPR_PEARSON 0
#define CM_EXPR_RMSE 1
#define CM_EXPR_JS 2
#define CM_EXPR_SPEARMAN 3
cm_status cm_expression_metrics(cm_handle_t h, const float *orig, const float *imputed, int64_t n, int64_t n_genes,
                                const int32_t *group, int32_t n_groups, int32_t method, float *out);
cm_status cm_expression_metrics_dev(cm_handle_t h, const float *d_orig, const float *d_imputed, int64_t n, int64_t n_genes,
                                    const int32_t *d_group, int32_t n_groups, int32_t method, float *d_out);

/* ------------------------------------------------- joint representations -----------
 * The dense linear algebra behind `use_rep=None` (cellmapper.py:211-235): fast CCA = truncated SVD of the cross-covariance
 * (X - rowmean)(Y - rowmean)^T / (G - 1) of query and reference expression (EmbeddingMixin.compute_fast_cca,
 * embedding.py:50-185; truncated_svd_cross_covariance, utils.py:230-398) and joint PCA (embedding.py:14-48).  Two
 * primitives over a tall cell x gene matrix A [n, p] -- dense row-major (dtype CM_F32 / CM_F64) or, with X = NULL,
 * canonical CSR (int64 indptr, int32 cols, float32 vals) -- in float64 on the f64 matrix cores:
 *   cm_gram    : out [p, p] = A'^T A'                 A' = A (center_rows = 0) or every row centred at its mean (1)
 *   cm_project : out [n, nc] = A' P                   P [p, nc] float64, nc <= 64; center_mode 0 none, 1 row means,
 *                                                      2 subtract col_shift [p] from every row (PCA's column means)
 * The p x p eigen-problems and the small SVD that turn these into singular vectors are the host's (embedding.py). */
cm_status cm_gram(cm_handle_t h, const void *X, const int64_t *indptr, const int32_t *cols, const float *vals, int64_t n, int32_t p,
                  int32_t dtype, int32_t center_rows, double *out);
cm_status cm_project(cm_handle_t h, const void *X, const int64_t *indptr, const int32_t *cols, const float *vals, int64_t n, int32_t p,
                     int32_t dtype, int32_t center_mode, const double *col_shift, const double *P, int32_t nc, double *out);

/* ---------------------------------------------------------------- diagnostics -----------------------
 * n independent products D = A*B + C on one v_mfma_f32_32x32x16_f16 chain each (chain k-steps of 16): a fp16 bits
 * [n, chain, 32, 16] row-major, b fp16 bits [n, chain, 16, 32], c / d float32 [n, 32, 32].  Test aid: pins the
 * accumulation arithmetic the phase-1 error bound assumes (no reference counterpart). */
cm_status cm_debug_mfma_f16(cm_handle_t h, const uint16_t *a, const uint16_t *b, const float *c, float *d, int32_t n,
                            int32_t chain);
/* the same for v_mfma_f32_16x16x32_f16 (the 16x16x32 re-tile of the hi*hi filter): n problems D = A(16x32) * B(32x16) + C(16x16),
 * `chain` k-steps accumulated; a: [n][chain][16][32] fp16 bits, b: [n][chain][32][16], c / d: [n][16][16] float32 */
cm_status cm_debug_mfma_f16_k32(cm_handle_t h, const uint16_t *a, const uint16_t *b, const float *c, float *d, int32_t n,
                                int32_t chain);

#ifdef __cplusplus
}
#endif
#endif /* CELLMAP_H */
