/*
 * cm_oracle.c -- CPU restatement of the CellMapper hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load this
 * library.  The product path (cellmapper_amd/) never links, imports or calls it.
 *
 * Every function restates -- from the algorithm's published behaviour, not from copied
 * source -- what the reference computes on the path
 *     Neighbors.compute_neighbors (sklearn branch)      /root/reference/src/cellmapper/model/knn.py:428-440
 *     NeighborsResults._compute_kernel_values            knn.py:166-226
 *     NeighborsResults._create_sparse_matrix             knn.py:79-111
 *     CellMapper._validate_and_normalize_mapping_matrix  cellmapper.py:99-137
 *     CellMapper._map_obs_categorical / _numerical       cellmapper.py:589-637
 *     CellMapper.map_obsm / map_layers                   cellmapper.py:307-383
 * The arithmetic of k-NN and SpMM lives in third-party code that is NOT under
 * /root/reference: scikit-learn 1.7.2 (EuclideanArgKmin, KDTree, cosine brute force) and
 * scipy 1.15.3 (csr_matvecs / csr_matmat / sparse argmax); their algorithms are restated
 * here and the restatement is pinned against golden vectors produced by running the
 * reference itself in the build container (oracle/make_golden.py writes the npz files in tests/golden).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).  -ffp-contract=off
 * matters: the operation order and the (non-)fusion of multiply-adds below is deliberate.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define CMO_OK 0
#define CMO_EINVAL 1
#define CMO_ENOMEM 2
#define CMO_ENOFINITE 3

/* kernel ids shared with include/cellmap.h */
enum { K_GAUSSIAN = 0, K_SCARCHES = 1, K_INVERSE_DISTANCE = 2, K_EQUAL = 3 };

int cmo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* Thread count of every later parallel region.  OMP_NUM_THREADS is read when libgomp is loaded -- in a process that has
 * already imported torch (which loads libgomp) the environment variable comes too late, and omp_get_max_threads() stays at
 * the host's core count while the process may run on a 16-core share: bench.py calls this with the cores it may use. */
void cmo_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

static inline double ld(const void *p, int is_f64, int64_t i) {
    return is_f64 ? ((const double *)p)[i] : (double)((const float *)p)[i];
}

/* ---------- (value, index) max-heap: keeps the k lexicographically smallest pairs ----------
 * sklearn keeps a fixed-size max-heap per query and rejects a candidate when
 * `val >= heap_max` (sklearn/utils/_heap.pyx); with reference rows scanned in ascending
 * index order that means "on an exact tie the lower index stays".  We make that rule
 * total by ordering on (value, index), so the result is independent of heap layout.   */
typedef struct { double v; int64_t i; } pair_t;
static inline int pair_less(pair_t a, pair_t b) { return a.v < b.v || (a.v == b.v && a.i < b.i); }

static void heap_sift_down(pair_t *h, int64_t n, int64_t pos) {
    pair_t x = h[pos];
    for (;;) {
        int64_t l = 2 * pos + 1, r = l + 1, m;
        if (l >= n) break;
        m = (r < n && pair_less(h[l], h[r])) ? r : l;
        if (!pair_less(x, h[m])) break;
        h[pos] = h[m];
        pos = m;
    }
    h[pos] = x;
}
static inline void heap_push(pair_t *h, int64_t k, pair_t x) {
    if (!pair_less(x, h[0])) return;
    h[0] = x;
    heap_sift_down(h, k, 0);
}
static int pair_cmp(const void *a, const void *b) {
    pair_t x = *(const pair_t *)a, y = *(const pair_t *)b;
    return pair_less(x, y) ? -1 : (pair_less(y, x) ? 1 : 0);
}

/* Exact Euclidean k-NN, restating sklearn's two CPU algorithms.
 *   form 0 ("gemm", sklearn brute / EuclideanArgKmin, used when d > 15 or k >= n_ref/2):
 *       everything in float64 (float32 inputs are upcast first);
 *       d2 = (|q|^2 + (-2 * q.y)) + |y|^2, clamped at 0; returned distance = sqrt(d2).
 *       (_argkmin.pyx.tp `_compute_and_reduce_distances_on_chunks`, _middle_term_computer GEMM alpha=-2)
 *   form 1 ("direct", sklearn kd_tree for d <= 15): float64 sum_j (q_j - y_j)^2, sqrt.
 * Output rows are sorted ascending by (distance, index).  qry is "X" and ref is "Y" in sklearn's naming. */
/* The scan is cache-blocked (round 2; the per-pair arithmetic and the per-query visiting order are unchanged, so the
 * golden vectors still pin it): the reference is re-laid once into feature-major tiles Rt[tile][c][CMO_RB] (set-up,
 * timed separately: cmo_last_setup_seconds), a thread owns a block of CMO_QB queries and walks the tiles in ascending
 * order, so a tile is read from DRAM once per CMO_QB queries instead of once per query; inside a tile the dot products
 * of one query with CMO_RB rows advance together, one fused multiply-add per (row, feature) in ascending feature order
 * -- the same chain as before, vectorised ACROSS rows by the compiler (target_clones: AVX-512 / AVX2+FMA / baseline). */
#define CMO_RB 256
#define CMO_QB 32
static double g_last_setup_s = 0.0;
double cmo_last_setup_seconds(void) { return g_last_setup_s; }
static double now_s(void) {
#ifdef _OPENMP
    return omp_get_wtime();
#else
    return 0.0;
#endif
}

__attribute__((target_clones("avx512f", "avx2,fma", "default")))
static void tile_gemm_form(const double *restrict q, const double *restrict Rt, const double *restrict yn, int32_t d,
                           double xn, double *restrict d2) {
    double acc[CMO_RB];
    for (int j = 0; j < CMO_RB; j++) acc[j] = 0.0;
    for (int32_t c = 0; c < d; c++) {
        const double qc = q[c];
        const double *restrict y = Rt + (size_t)c * CMO_RB;
        for (int j = 0; j < CMO_RB; j++) acc[j] = fma(qc, y[j], acc[j]);
    }
    for (int j = 0; j < CMO_RB; j++) {
        double v = (xn + (-2.0 * acc[j])) + yn[j];
        d2[j] = v > 0.0 ? v : 0.0; /* max(0., d2), also squashes -0. and NaN */
    }
}

/* Four queries against one tile (round 3): the same fma chains, but a loaded vector of reference values now feeds four
 * queries' accumulators -- 16 vector accumulators (4 queries x 32 rows) stay in registers over the feature loop, where the
 * one-query form kept its 256 partial sums in memory (three memory operations per fused multiply-add).  Per-pair
 * arithmetic and order unchanged: the same bits (the golden vectors pin it). */
#define CMO_JB 32
__attribute__((target_clones("avx512f", "avx2,fma", "default")))
static void tile_gemm_form4(const double *restrict q, const double *restrict Rt, const double *restrict yn, int32_t d,
                            const double *restrict xn, double *restrict d2 /* [4][CMO_RB] */) {
    const double *restrict q0 = q, *restrict q1 = q + d, *restrict q2 = q + 2 * (size_t)d, *restrict q3 = q + 3 * (size_t)d;
    for (int j0 = 0; j0 < CMO_RB; j0 += CMO_JB) {
        double a0[CMO_JB], a1[CMO_JB], a2[CMO_JB], a3[CMO_JB];
        for (int j = 0; j < CMO_JB; j++) { a0[j] = 0.0; a1[j] = 0.0; a2[j] = 0.0; a3[j] = 0.0; }
        for (int32_t c = 0; c < d; c++) {
            const double *restrict y = Rt + (size_t)c * CMO_RB + j0;
            const double c0 = q0[c], c1 = q1[c], c2 = q2[c], c3 = q3[c];
            for (int j = 0; j < CMO_JB; j++) {
                const double yj = y[j];
                a0[j] = fma(c0, yj, a0[j]);
                a1[j] = fma(c1, yj, a1[j]);
                a2[j] = fma(c2, yj, a2[j]);
                a3[j] = fma(c3, yj, a3[j]);
            }
        }
        for (int j = 0; j < CMO_JB; j++) {
            const double ynj = yn[j0 + j];
            double v0 = (xn[0] + (-2.0 * a0[j])) + ynj, v1 = (xn[1] + (-2.0 * a1[j])) + ynj;
            double v2 = (xn[2] + (-2.0 * a2[j])) + ynj, v3 = (xn[3] + (-2.0 * a3[j])) + ynj;
            d2[j0 + j] = v0 > 0.0 ? v0 : 0.0;
            d2[CMO_RB + j0 + j] = v1 > 0.0 ? v1 : 0.0;
            d2[2 * CMO_RB + j0 + j] = v2 > 0.0 ? v2 : 0.0;
            d2[3 * CMO_RB + j0 + j] = v3 > 0.0 ? v3 : 0.0;
        }
    }
}

__attribute__((target_clones("avx512f", "avx2,fma", "default")))
static void tile_direct_form(const double *restrict q, const double *restrict Rt, int32_t d, double *restrict d2) {
    for (int j = 0; j < CMO_RB; j++) d2[j] = 0.0;
    for (int32_t c = 0; c < d; c++) {
        const double qc = q[c];
        const double *restrict y = Rt + (size_t)c * CMO_RB;
        for (int j = 0; j < CMO_RB; j++) { double t = qc - y[j]; d2[j] = d2[j] + t * t; }
    }
}

/* sklearn's metric="sqeuclidean" (EuclideanArgKmin, squared=True: _argkmin.pyx.tp skips the final sqrt): set before the call */
static int g_squared_out = 0;
void cmo_set_squared_output(int on) { g_squared_out = on != 0; }

int cmo_knn_euclidean(const void *ref, int64_t n_ref, const void *qry, int64_t nq, int32_t d, int32_t is_f64,
                      int32_t k, int32_t form, double *out_dist, int64_t *out_idx) {
    if (k < 1 || k > n_ref || d < 1 || nq < 0) return CMO_EINVAL;
    const double t_setup0 = now_s();
    /* float32 input is upcast to float64 first, as sklearn does per chunk (_middle_term_computer.pyx.tp); padded rows
     * of the last tile are never pushed (j < n_ref below) */
    const int64_t n_tiles = (n_ref + CMO_RB - 1) / CMO_RB;
    double *Rt = (double *)malloc(sizeof(double) * (size_t)n_tiles * CMO_RB * (size_t)d);
    double *yn = (double *)malloc(sizeof(double) * (size_t)n_tiles * CMO_RB);
    if (!Rt || !yn) { free(Rt); free(yn); return CMO_ENOMEM; }
#pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < n_tiles; t++) {
        double *T = Rt + (size_t)t * CMO_RB * d;
        for (int j = 0; j < CMO_RB; j++) {
            const int64_t r = t * CMO_RB + j;
            double s = 0.0;
            for (int32_t c = 0; c < d; c++) {
                double v = r < n_ref ? ld(ref, is_f64, r * d + c) : 0.0;
                T[(size_t)c * CMO_RB + j] = v;
                s = fma(v, v, s);
            }
            yn[r] = s;
        }
    }
    g_last_setup_s = now_s() - t_setup0;
    int err = 0;
    const int64_t n_qblocks = (nq + CMO_QB - 1) / CMO_QB;
#pragma omp parallel
    {
        pair_t *h = (pair_t *)malloc(sizeof(pair_t) * (size_t)k * CMO_QB);
        double *q = (double *)malloc(sizeof(double) * (size_t)d * CMO_QB);
        double xn[CMO_QB], d2[4 * CMO_RB];
        if (!h || !q) {
#pragma omp atomic write
            err = CMO_ENOMEM;
        } else {
#pragma omp for schedule(dynamic, 1)
            for (int64_t b = 0; b < n_qblocks; b++) {
                const int64_t i0 = b * CMO_QB;
                const int nb = (int)((nq - i0) < CMO_QB ? (nq - i0) : CMO_QB);
                for (int u = 0; u < nb; u++) {
                    double s = 0.0;
                    for (int32_t c = 0; c < d; c++) { double v = ld(qry, is_f64, (i0 + u) * d + c); q[u * d + c] = v; s = fma(v, v, s); }
                    xn[u] = s;
                    for (int32_t t = 0; t < k; t++) { h[(size_t)u * k + t].v = INFINITY; h[(size_t)u * k + t].i = INT64_MAX; }
                }
                for (int64_t t = 0; t < n_tiles; t++) {
                    const double *T = Rt + (size_t)t * CMO_RB * d;
                    const int64_t j0 = t * CMO_RB;
                    const int nr = (int)((n_ref - j0) < CMO_RB ? (n_ref - j0) : CMO_RB);
                    int u = 0;
                    if (form == 0)
                        for (; u + 4 <= nb; u += 4) {   /* four queries share every loaded reference vector */
                            tile_gemm_form4(q + (size_t)u * d, T, yn + j0, d, xn + u, d2);
                            for (int w = 0; w < 4; w++) {
                                pair_t *hu = h + (size_t)(u + w) * k;
                                const double *dw = d2 + (size_t)w * CMO_RB;
                                for (int j = 0; j < nr; j++)   /* rows in ascending index order, as sklearn scans them */
                                    if (dw[j] <= hu[0].v) { pair_t x = { dw[j], j0 + j }; heap_push(hu, k, x); }
                            }
                        }
                    for (; u < nb; u++) {
                        pair_t *hu = h + (size_t)u * k;
                        if (form == 0) tile_gemm_form(q + (size_t)u * d, T, yn + j0, d, xn[u], d2);
                        else tile_direct_form(q + (size_t)u * d, T, d, d2);
                        for (int j = 0; j < nr; j++)   /* rows in ascending index order, as sklearn scans them */
                            if (d2[j] <= hu[0].v) { pair_t x = { d2[j], j0 + j }; heap_push(hu, k, x); }
                    }
                }
                for (int u = 0; u < nb; u++) {
                    pair_t *hu = h + (size_t)u * k;
                    qsort(hu, (size_t)k, sizeof(pair_t), pair_cmp);
                    for (int32_t t = 0; t < k; t++) { out_dist[(i0 + u) * k + t] = g_squared_out ? hu[t].v : sqrt(hu[t].v); out_idx[(i0 + u) * k + t] = hu[t].i; }
                }
            }
        }
        free(h);
        free(q);
    }
    free(yn);
    free(Rt);
    return err;
}

/* ---------- numpy-style pairwise summation (so that sigma matches np.mean / np.std bit for bit) ---------- */
static double pairwise_sum(const double *a, int64_t n) {
    if (n < 8) {
        double r = 0.0;
        for (int64_t i = 0; i < n; i++) r += a[i];
        return r;
    } else if (n <= 128) {
        double r[8];
        int64_t i;
        for (int j = 0; j < 8; j++) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return pairwise_sum(a, n2) + pairwise_sum(a + n2, n - n2);
    }
}

/* Kernel weights, knn.py:166-226.  valid = (idx != -1) & isfinite(dist) (knn.py:68-77).
 * gaussian: sigma = mean(valid d); w = exp(-(d*d) / (2*sigma*sigma))
 * scarches: s = (2/std(valid d))^2;  w = exp(-d / s)
 * inverse_distance: w = 1/(d + eps);  equal: w = 1.   Invalid entries -> 0.
 * param_out receives sigma (gaussian) / s (scarches). */
int cmo_kernel_weights(const double *dist, const int64_t *idx, int64_t nq, int32_t k, int32_t kernel, double eps,
                       double *out_w, double *param_out) {
    int64_t n = nq * (int64_t)k, nv = 0;
    double *fin = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    if (!fin) return CMO_ENOMEM;
    for (int64_t t = 0; t < n; t++)
        if (idx[t] != -1 && isfinite(dist[t])) fin[nv++] = dist[t];
    if (nv == 0) { free(fin); return CMO_ENOFINITE; }
    double param = 0.0;
    if (kernel == K_GAUSSIAN) {
        param = pairwise_sum(fin, nv) / (double)nv;
    } else if (kernel == K_SCARCHES) {
        double mean = pairwise_sum(fin, nv) / (double)nv;
        for (int64_t t = 0; t < nv; t++) { double c = fin[t] - mean; fin[t] = c * c; }
        double sd = sqrt(pairwise_sum(fin, nv) / (double)nv);
        double r = 2.0 / sd;
        param = r * r;
    } else if (kernel != K_INVERSE_DISTANCE && kernel != K_EQUAL) {
        free(fin);
        return CMO_EINVAL;
    }
    free(fin);
    double two_s2 = 2.0 * (param * param);
#pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < n; t++) {
        double w = 0.0, dd = dist[t];
        if (idx[t] != -1 && isfinite(dd)) {
            switch (kernel) {
            case K_GAUSSIAN: w = exp(-(dd * dd) / two_s2); break;
            case K_SCARCHES: w = exp(-dd / param); break;
            case K_INVERSE_DISTANCE: w = 1.0 / (dd + eps); break;
            default: w = 1.0;
            }
        }
        out_w[t] = w;
    }
    if (param_out) *param_out = param;
    return CMO_OK;
}

/* COO -> canonical CSR (knn.py:97-111: columns ascending in a row, duplicates summed) followed by
 * row normalisation (cellmapper.py:125-135: float64 row sums in CSR order, zero sums -> 1,
 * multiply by the reciprocal, cast to float32).  If `normalize` is 0 only the packing is done and
 * out_val_f64 (required) receives the un-normalised float64 values.
 * out_indptr has nq+1 entries; out_cols/out_val* need room for nq*k entries.  Returns nnz in *nnz_out. */
typedef struct { int64_t c; double w; int32_t pos; } ent_t;
static int ent_cmp(const void *a, const void *b) {
    const ent_t *x = (const ent_t *)a, *y = (const ent_t *)b;
    if (x->c != y->c) return x->c < y->c ? -1 : 1;
    return x->pos - y->pos;
}
int cmo_mapping_matrix(const double *w, const double *dist, const int64_t *idx, int64_t nq, int32_t k, int64_t n_ref,
                       int32_t normalize, int64_t *out_indptr, int32_t *out_cols, float *out_val_f32,
                       double *out_val_f64, int64_t *nnz_out) {
    if (!out_val_f64) return CMO_EINVAL;
    ent_t *e = (ent_t *)malloc(sizeof(ent_t) * (size_t)(k > 0 ? k : 1));
    if (!e) return CMO_ENOMEM;
    int64_t nnz = 0;
    out_indptr[0] = 0;
    for (int64_t i = 0; i < nq; i++) {
        int32_t m = 0;
        for (int32_t t = 0; t < k; t++) {
            int64_t c = idx[i * k + t];
            if (c == -1 || !isfinite(dist[i * k + t])) continue;
            if (c < 0 || c >= n_ref) { free(e); return CMO_EINVAL; }
            e[m].c = c; e[m].w = w[i * k + t]; e[m].pos = t; m++;
        }
        qsort(e, (size_t)m, sizeof(ent_t), ent_cmp);
        int64_t start = nnz;
        for (int32_t t = 0; t < m; t++) {
            if (nnz > start && out_cols[nnz - 1] == (int32_t)e[t].c) {
                out_val_f64[nnz - 1] += e[t].w;
            } else {
                out_cols[nnz] = (int32_t)e[t].c;
                out_val_f64[nnz] = e[t].w;
                nnz++;
            }
        }
        out_indptr[i + 1] = nnz;
    }
    free(e);
    if (normalize) {
        if (!out_val_f32) return CMO_EINVAL;
        for (int64_t i = 0; i < nq; i++) {
            double s = 0.0;
            for (int64_t p = out_indptr[i]; p < out_indptr[i + 1]; p++) s += out_val_f64[p] * 1.0;
            if (s == 0.0) s = 1.0;
            double r = 1.0 / s;
            for (int64_t p = out_indptr[i]; p < out_indptr[i + 1]; p++) out_val_f32[p] = (float)(out_val_f64[p] * r);
        }
    }
    if (nnz_out) *nnz_out = nnz;
    return CMO_OK;
}

/* Label vote, cellmapper.py:589-623 == (M @ onehot) then sparse argmax / max.
 * float32 per-category sums accumulated in ascending column order (scipy csr_matmat), zero sums are
 * dropped by csr_matmat; argmax ties -> lowest category (scipy _arg_min_or_max_axis on sorted columns);
 * empty row -> category 0, confidence 0. */
int cmo_label_vote(const int64_t *indptr, const int32_t *cols, const float *vals, int64_t nq, const int32_t *codes,
                   int32_t n_cat, int32_t *out_code, float *out_conf) {
    int err = 0;
#pragma omp parallel
    {
        float *sums = (float *)malloc(sizeof(float) * (size_t)(n_cat > 0 ? n_cat : 1));
        if (!sums) {
#pragma omp atomic write
            err = CMO_ENOMEM;
        } else {
#pragma omp for schedule(static)
            for (int64_t i = 0; i < nq; i++) {
                for (int32_t c = 0; c < n_cat; c++) sums[c] = 0.0f;
                for (int64_t p = indptr[i]; p < indptr[i + 1]; p++) {
                    int32_t c = codes[cols[p]];
                    sums[c] = sums[c] + vals[p] * 1.0f;
                }
                int32_t best = 0; float bv = 0.0f; int found = 0;
                for (int32_t c = 0; c < n_cat; c++) {
                    if (sums[c] != 0.0f && (!found || sums[c] > bv)) { best = c; bv = sums[c]; found = 1; }
                }
                if (!found) { best = 0; bv = 0.0f; }
                else if (!(bv > 0.0f)) { /* negative max with implicit zeros present: zero wins */
                    int32_t z = 0; while (z < n_cat && sums[z] != 0.0f) z++;
                    if (z < n_cat) { best = z; bv = 0.0f; }
                }
                out_code[i] = best; out_conf[i] = bv;
            }
        }
        free(sums);
    }
    return err;
}

/* CSR (float32 data) x dense, scipy csr_matvecs order: y[i,:] += a * x[j,:] over the row's entries in
 * ascending column order; separate multiply and add (no FMA); accumulation dtype = upcast(f32, B dtype). */
int cmo_spmm_dense(const int64_t *indptr, const int32_t *cols, const float *vals, int64_t nq, const void *B,
                   int64_t n_out, int32_t b_is_f64, void *out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nq; i++) {
        if (b_is_f64) {
            double *y = (double *)out + i * n_out;
            for (int64_t c = 0; c < n_out; c++) y[c] = 0.0;
            for (int64_t p = indptr[i]; p < indptr[i + 1]; p++) {
                double a = (double)vals[p];
                const double *x = (const double *)B + (int64_t)cols[p] * n_out;
                for (int64_t c = 0; c < n_out; c++) y[c] = y[c] + a * x[c];
            }
        } else {
            float *y = (float *)out + i * n_out;
            for (int64_t c = 0; c < n_out; c++) y[c] = 0.0f;
            for (int64_t p = indptr[i]; p < indptr[i + 1]; p++) {
                float a = vals[p];
                const float *x = (const float *)B + (int64_t)cols[p] * n_out;
                for (int64_t c = 0; c < n_out; c++) y[c] = y[c] + a * x[c];
            }
        }
    }
    return CMO_OK;
}

/* CSR (f32) x CSR (f32) -> dense float32 [nq, n_out] (scipy csr_matmat accumulation order: for each entry of
 * the left row in ascending column order, add a*b over the right row's entries).  The caller converts to CSR
 * dropping exact zeros, as csr_matmat does. */
int cmo_spmm_csr_to_dense(const int64_t *indptr, const int32_t *cols, const float *vals, int64_t nq,
                          const int64_t *b_indptr, const int32_t *b_cols, const float *b_vals, int64_t n_out,
                          float *out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nq; i++) {
        float *y = out + i * n_out;
        for (int64_t c = 0; c < n_out; c++) y[c] = 0.0f;
        for (int64_t p = indptr[i]; p < indptr[i + 1]; p++) {
            float a = vals[p];
            int64_t j = cols[p];
            for (int64_t t = b_indptr[j]; t < b_indptr[j + 1]; t++) y[b_cols[t]] = y[b_cols[t]] + a * b_vals[t];
        }
    }
    return CMO_OK;
}
