// K0: exact k-NN on gfx950 -- row preparation, phase-2 float64 re-rank + certification, exact fallback, shard merge,
// and the host-side orchestration (cm_knn_run).  Phase 1 (MFMA candidate filter) lives in knn_phase1.h.
//
// Reference interface replaced: Neighbors.compute_neighbors, sklearn branch, /root/reference/src/cellmapper/model/
// knn.py:428-440 (sklearn.neighbors.NearestNeighbors(...).fit(xrep).kneighbors(yrep)).
#include <algorithm>
#include <type_traits>
#include "cm_internal.h"
#include "knn_phase1.h"
#include "knn_phase1_f16.h"
#include "knn_phase1_q4.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

cm_status cm_p1_wide_launch(cm_handle_t h, int KP, int NW, dim3 grid, size_t lds, const float *Q, const float *R, int tps, int n_tiles,
                            int Kc, int C, int *cand, float *cthr);   // knn_phase1_inst.hip (group 3): K-chunked kernel, d >= 260
cm_status cm_p1f16q_launch(cm_handle_t h, int KSTEPS, int grid, size_t lds, const cmk2::P1Args &a);
#ifdef CM_ABLATE
cm_status cm_p1f16r_launch(cm_handle_t h, int KSTEPS, int grid, size_t lds, const cmk2::P1Args &a);   // 16x16x32 re-tile (knn_phase1_r16.h): experiment builds only
#endif
cm_status cm_p1f16_launch(cm_handle_t h, int KSTEPS, int grid, size_t lds, const cmk2::P1Args &a);
cm_status cm_p1f16_seed10_launch(cm_handle_t h, int KSTEPS, int grid, size_t lds, const cmk2::P1Args &a);
cm_status cm_p1lowd_launch(cm_handle_t h, int D, int grid, size_t lds, const cmk2::P1Args &a);
cm_status cm_p1lowd_metric_launch(cm_handle_t h, int D, int op, int grid, size_t lds, const cmk2::P1Args &a);
cm_status cm_p1direct_wide_launch(cm_handle_t h, int D, int op, int grid, size_t lds, const cmk2::P1Args &a);
// grid_knn.hip: uniform-grid candidate search for d <= 3
size_t cm_grid_ws_bytes(int64_t n_ref, int64_t ncell_max);
int64_t cm_grid_ncell_max(int64_t n_ref, int k);
template <typename T>
cm_status cm_grid_build(cm_handle_t h, const T *d_ref, int64_t n_ref, int d, int k, void **gi_out, int64_t **cell_start_out,
                        float **sxyz_out, int **sidx_out);
template <typename T>
cm_status cm_grid_query(cm_handle_t h, const T *d_qry, int64_t nq, int d, const int *rowmap, const void *gi,
                        const int64_t *cell_start, const float *sxyz, const int *sidx, int k, int Kc, const unsigned *gmax,
                        int *cand, float *cthr);
cm_status cm_p1f16w_launch(cm_handle_t h, int grid, size_t lds, const cmk2::P1Args &a);

// phase-1 instantiation groups (knn_phase1_inst.hip)
#define CM_DECL_P1(g)                                                                                                  \
    bool cm_p1_launch_g##g(cm_handle_t h, int KP, int NW, dim3 grid, size_t lds, const float *Q, const float *R,       \
                           int tps, int n_tiles, int Kc, int C, int *cand, float *cthr, cm_status *st);
CM_DECL_P1(0) CM_DECL_P1(1) CM_DECL_P1(2) CM_DECL_P1(3)

namespace {

constexpr int kSupportedKP[] = {8, 12, 16, 20, 24, 28, 32, 36, 40, 44, 48, 52, 56, 60, 64, 68, 84, 100, 132, 164, 196, 260};
constexpr int RR_NCMAX = 1024;  // max candidates per query handled by the re-rank / merge kernels
constexpr int RR_DMAX = 260;   // LDS stride (doubles) of a staged query row for d <= 259; wider rows: q_stride(d)
constexpr int KNN_DMAX = 2047;  // widest row whose float64 copy is staged in LDS by the re-rank / fallback kernels (wider: their GQ form)
__host__ __device__ constexpr int q_stride(int d) { return d < RR_DMAX ? RR_DMAX : ((d + 4) & ~3); }
constexpr int MG_NCMAX = CM_MERGE_NCMAX;   // max nparts*k in the shard merge
constexpr int FB_CAP = 4096;    // fallback collection capacity per query (a query with more rows inside its bound goes to knn_fallback_select)
constexpr int OVF_CAP = 16384;  // queries the selection kernel takes per call (more: the call fails loudly)
constexpr int FB_BLOCKS = 1024;
constexpr int FBS_SLICES = 512;  // sliced fallback: reference slices scanned in parallel per flagged query ...
constexpr int FBS_NQ = 1024;     // ... for the first FBS_NQ flagged queries (the rest: one workgroup per query)
constexpr int FBS_FY = 8;
constexpr int FBS_QG = 4;       // flagged queries served by one staged chunk of reference rows

int pick_kp(int d) {
    for (int kp : kSupportedKP)
        if (kp >= d + 1) return kp;
    return -1;
}

// ------------------------------------------------------------------ row preparation -----------------
// X [n,d] (T) -> P [n_pad, KP] float32:  reference rows [-2y, |y|^2, 0..], query rows [q, 1, 0..];
// cosine: rows are L2-normalised first (zero rows stay zero, like sklearn.preprocessing.normalize).
// Padded reference rows carry |y|^2 = +inf so they never pass a threshold.
// gmax[0] = max |y|^2, gmax[1] = max |y| over reference rows (float bits; valid for non-negative floats).
template <typename T>
__global__ void prep_rows(const T *__restrict__ X, int64_t n, int d, int KP, int64_t n_pad, int is_ref, int metric,
                          float *__restrict__ P, float *__restrict__ rownorm, unsigned *__restrict__ gmax) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float yn_f = 0.f;
    if (i < n_pad) {
        float *out = P + i * KP;
        if (i < n) {
            const T *x = X + i * d;
            double scale = 1.0;
            if (metric == CM_METRIC_COSINE) {
                double s = 0.0;
                for (int c = 0; c < d; ++c) { double v = (double)x[c]; s = fma(v, v, s); }
                double nrm = sqrt(s);
                scale = nrm > 0.0 ? 1.0 / nrm : 1.0;
            }
            double s2 = 0.0;
            for (int c = 0; c < d; ++c) {
                float v = (float)((double)x[c] * scale);
                s2 = fma((double)v, (double)v, s2);
                out[c] = is_ref ? -2.0f * v : v;
            }
            yn_f = (float)s2;
            out[d] = is_ref ? yn_f : 1.0f;
            for (int c = d + 1; c < KP; ++c) out[c] = 0.f;
            if (rownorm) rownorm[i] = sqrtf(yn_f) * 1.0000002f;
        } else {
            for (int c = 0; c < KP; ++c) out[c] = 0.f;
            out[d] = is_ref ? INFINITY : 1.0f;
            yn_f = 0.f;
        }
    }
    if (is_ref && gmax) {
        float m = yn_f;
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if ((threadIdx.x & 63) == 0) {
            atomicMax(&gmax[0], __float_as_uint(m));
            atomicMax(&gmax[1], __float_as_uint(sqrtf(m) * 1.0000002f));
        }
    }
}

// ------------------------------------------------------------------ split-fp16 preparation ----------
// max |coordinate| over a matrix (after the optional cosine normalisation): decides the power-of-two scale that
// keeps the fp16 halves in range.  gmax[4] holds the float bits of the maximum.
// Column means of the (cosine: normalised) reference rows.  Distances are translation invariant, so phase 1 works on
// rows centred at the reference mean: the fp16 split and the error bound then scale with the SPREAD of the data, not
// with its offset from the origin.  (Phase 2 always uses the caller's original rows.)  Deterministic two-stage sum.
constexpr int CMEAN_BLOCKS = 1024;   // (round 4: 256 -> 1 024 blocks, four rows per wave in flight: 0.66 -> ~0.15 ms per 1.5 M x 50 rows)
constexpr int CMEAN_COLS = 128;   // the staged search handles d <= 128: a lane owns columns c and c + 64
template <typename T>
__global__ __launch_bounds__(256) void colmean_partial(const T *__restrict__ X, int64_t n, int d, int metric,
                                                        double *__restrict__ partial /*[CMEAN_BLOCKS][CMEAN_COLS]*/) {
    __shared__ double sh[4][CMEAN_COLS];
    const int c = threadIdx.x & 63, w = threadIdx.x >> 6;  // lane = column (pair), 4 waves stride over rows
    double acc0 = 0.0, acc1 = 0.0;
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * per, r1 = r0 + per < n ? r0 + per : n;
    int64_t i = r0 + w;
    if (metric != CM_METRIC_COSINE) {   // four rows per wave in flight (independent loads), added in row order
        for (; i + 12 < r1; i += 16) {
            T a0 = 0, a1 = 0, a2 = 0, a3 = 0, b0 = 0, b1 = 0, b2 = 0, b3 = 0;
            if (c < d) { a0 = X[i * d + c]; a1 = X[(i + 4) * d + c]; a2 = X[(i + 8) * d + c]; a3 = X[(i + 12) * d + c]; }
            if (c + 64 < d) { b0 = X[i * d + c + 64]; b1 = X[(i + 4) * d + c + 64]; b2 = X[(i + 8) * d + c + 64]; b3 = X[(i + 12) * d + c + 64]; }
            acc0 += (double)a0; acc0 += (double)a1; acc0 += (double)a2; acc0 += (double)a3;
            acc1 += (double)b0; acc1 += (double)b1; acc1 += (double)b2; acc1 += (double)b3;
        }
    }
    for (; i < r1; i += 4) {
        const double x0 = c < d ? (double)X[i * d + c] : 0.0, x1 = c + 64 < d ? (double)X[i * d + c + 64] : 0.0;
        double scale = 1.0;
        if (metric == CM_METRIC_COSINE) {   // (the row's norm by a wave reduction: every lane used to re-read the whole row)
            double s = x0 * x0 + x1 * x1;
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            const double nrm = sqrt(s);
            scale = nrm > 0.0 ? 1.0 / nrm : 1.0;
        }
        acc0 += x0 * scale;
        acc1 += x1 * scale;
    }
    sh[w][c] = acc0;
    sh[w][c + 64] = acc1;
    __syncthreads();
    if (w < 2) {
        const int cc = c + 64 * w;
        partial[blockIdx.x * CMEAN_COLS + cc] = (sh[0][cc] + sh[1][cc]) + (sh[2][cc] + sh[3][cc]);
    }
}
__global__ void colmean_final(const double *__restrict__ partial, int64_t n, int d, double *__restrict__ mean /*[CMEAN_COLS]*/) {
    const int c = threadIdx.x;
    if (c < CMEAN_COLS) {
        double s = 0.0;
        for (int b = 0; b < CMEAN_BLOCKS; ++b) s += partial[b * CMEAN_COLS + c];
        mean[c] = (c < d && n > 0) ? s / (double)n : 0.0;
    }
}

template <typename T>
__global__ void maxabs_rows(const T *__restrict__ X, int64_t n, int d, int metric, const double *__restrict__ mean,
                            unsigned *__restrict__ gmax) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float m = 0.f;
    if (i < n) {
        const T *x = X + i * d;
        double scale = 1.0;
        if (metric == CM_METRIC_COSINE) {
            double s = 0.0;
            for (int c = 0; c < d; ++c) { double v = (double)x[c]; s = fma(v, v, s); }
            double nrm = sqrt(s);
            scale = nrm > 0.0 ? 1.0 / nrm : 1.0;
        }
        for (int c = 0; c < d; ++c) m = fmaxf(m, fabsf((float)((double)x[c] * scale - (mean ? mean[c] : 0.0))));
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(&gmax[4], __float_as_uint(m));
}
// Euclidean rows need no per-row scale: the same maximum over the flat array, coalesced (one element per thread and step)
template <typename T>
__global__ __launch_bounds__(256) void maxabs_flat(const T *__restrict__ X, int64_t total, int d, const double *__restrict__ mean,
                                                    unsigned *__restrict__ gmax) {
    float m = 0.f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int c = (int)(e % d);
        m = fmaxf(m, fabsf((float)((double)X[e] - (mean ? mean[c] : 0.0))));
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(&gmax[4], __float_as_uint(m));
}
// sc = 2^(top_exp - e) with maxabs = m * 2^e, m in [0.5, 1): scaled coordinates land in [2^(top_exp-1), 2^top_exp) at most
// (top_exp = 12 for K <= 64, 11 for K = 128 so that sc^2 |y|^2 <= 2^30 still fits the first norm digit a1 = N * 2^-15)
__global__ void f16_scale_kernel(const unsigned *__restrict__ gmax, float *__restrict__ scale3, int top_exp) {
    if (threadIdx.x == 0) {
        float mx = __uint_as_float(gmax[4]);
        int e = 0;
        float sc = 1.f;
        if (mx > 0.f && isfinite(mx)) { (void)frexpf(mx, &e); sc = ldexpf(1.f, top_exp - e); }
        scale3[0] = sc;
        scale3[1] = sc * sc;
        scale3[2] = 1.f / (sc * sc);
    }
}
// X [n,d] -> split-fp16 prepared rows [hi(K) | lo(K)], K = 16*KSTEPS >= d + 3.
//   queries   : hi/lo of sc*q in slots [0,d); slots d..d+2 of hi hold the constants 2^15, 2^4, 2^-7; plain rows.
//   references: hi/lo of -2*sc*y in slots [0,d); slots d..d+2 of hi hold fp16 digits a1,a2,a3 of N = sc^2*|y|^2 with
//               N = a1*2^15 + a2*2^4 + a3*2^-7 (residual <= 2^-33 N), so q'.y' = sc^2 (|y|^2 - 2 q.y); padded rows carry
//               a1 = +inf.  Rows are grouped in 128-row tile images whose 16-byte granules are XOR-swizzled by row
//               (the LDS image knn_cand_f16 copies verbatim with LDS-DMA).
template <typename T>
__global__ void prep_rows_f16(const T *__restrict__ X, int64_t n, int d, int KSTEPS, int64_t n_pad, int is_ref, int metric,
                              const float *__restrict__ scale3, const double *__restrict__ mean, char *__restrict__ P,
                              float *__restrict__ rownorm, unsigned *__restrict__ gmax, int yn_block, int hi_only,
                              const int *__restrict__ rowmap, float *__restrict__ rowlo,
                              const double *__restrict__ cen = nullptr, const int *__restrict__ blk_bucket = nullptr,
                              float *__restrict__ rownorm_c = nullptr, float *__restrict__ rowcen = nullptr, int stage_lds = 0) {
    // cen != nullptr (bucket-centred queries, knn_cand_f16<.,1,true>): the row is prepared as q' - cen[bucket of its
    // 512-query block]; rownorm stays |q'|, rownorm_c = |q' - c|, rowcen = |c|; rowmap entries < 0 are padding (zero row).
    // rowlo (queries): |lo part of the row| in input units -- the certificate of filter level 1 bounds the dropped
    // lo(query)*hi(ref) product with the query's ACTUAL lo norm (typically a third of the worst case 2^-11 |q'|);
    // reference rows: gmax[5] = max over rows of |lo part| / |row| (same purpose, dropped hi(query)*lo(ref)).
    // yn_block != 0 (K = 128 kernel): no norm slots; the tile image ends with a block of 128 floats sc^2|y|^2
    // hi_only != 0 (reference image of filter levels 1/2): rows hold the hi halves only
    const int GRAN = cmk2::row_bytes(KSTEPS) / 16;
    const int ROWB = (is_ref && hi_only) ? KSTEPS * 32 : cmk2::row_bytes(KSTEPS);
    const int TILEB = cmk2::BR * ROWB + (yn_block ? 1024 : 0);
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // Source rows staged through LDS (round 2): a thread per row reading its row straight from global memory touches 64
    // cache lines per load instruction, twice (norm pass, conversion pass).  When the block's rows fit (the launcher passes
    // `stage_lds` = bytes of dynamic LDS, else 0), each wave copies whole rows with coalesced loads first; the arithmetic
    // below is unchanged.  Row stride d + 1 elements (odd for the usual even d: conflict-free thread-per-row reads).
    extern __shared__ __align__(16) unsigned char prep_smem[];
    T *sx = reinterpret_cast<T *>(prep_smem);
    const bool staged = stage_lds > 0;
    if (staged) {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwv = blockDim.x >> 6;
        const int64_t b0 = (int64_t)blockIdx.x * blockDim.x;
        for (int rr = wave; rr < (int)blockDim.x; rr += nwv) {
            const int64_t ii = b0 + rr;
            if (ii >= n_pad) break;
            const bool lv = ii < n && !(rowmap && rowmap[ii] < 0);
            if (!lv) continue;
            const T *src = X + (rowmap ? (int64_t)rowmap[ii] : ii) * d;
            for (int c = lane; c < d; c += 64) sx[(size_t)rr * (d + 1) + c] = src[c];
        }
        __syncthreads();
    }
    float yn_f = 0.f, lo_ratio = 0.f;
    if (i < n_pad) {
        const float sc = scale3[0];
        const int64_t tile = i / cmk2::BR;
        const int r = (int)(i - tile * cmk2::BR);
        char *rowp = is_ref ? P + tile * TILEB + (size_t)r * ROWB : P + i * (size_t)ROWB;
        const int f = is_ref ? cmk2::swz_b(r, ROWB) : 0;
        double nscale = 1.0;
        // rowmap (queries only): prepare row i from source row rowmap[i] (query subsets of the staged search)
        const bool live = i < n && !(rowmap && rowmap[i] < 0);
        const T *x = staged ? sx + (size_t)threadIdx.x * (d + 1)
                            : X + ((rowmap && live) ? (int64_t)rowmap[i] : (live ? i : 0)) * d;
        const double *cj = (cen && blk_bucket) ? cen + (size_t)blk_bucket[i / cmk2::QB] * d : nullptr;
        double s2 = 0.0;  // squared norm of the centred (cosine: normalised, then centred) row
        double s2c = 0.0, c2 = 0.0;   // centred mode: |q' - c|^2, |c|^2
        if (live) {
            if (metric == CM_METRIC_COSINE) {
                double s = 0.0;
                for (int c = 0; c < d; ++c) { double v = (double)x[c]; s = fma(v, v, s); }
                double nrm = sqrt(s);
                nscale = nrm > 0.0 ? 1.0 / nrm : 1.0;
            }
            for (int c = 0; c < d; ++c) {
                double v = (double)x[c] * nscale - mean[c];
                s2 = fma(v, v, s2);
                if (cj) { const double w = v - cj[c]; s2c = fma(w, w, s2c); c2 = fma(cj[c], cj[c], c2); }
            }
        }
        // norm digits / query constants for slots d, d+1, d+2
        float slot[3];
        if (is_ref) {
            if (live) {
                double N = s2 * (double)sc * (double)sc;
                _Float16 a1 = (_Float16)(float)(N * 0x1p-15);
                double r1 = N - (double)(float)a1 * 0x1p15;
                _Float16 a2 = (_Float16)(float)(r1 * 0x1p-4);
                double r2 = r1 - (double)(float)a2 * 0x1p4;
                _Float16 a3 = (_Float16)(float)(r2 * 0x1p7);
                slot[0] = (float)a1; slot[1] = (float)a2; slot[2] = (float)a3;
            } else {
                slot[0] = INFINITY; slot[1] = 0.f; slot[2] = 0.f;
            }
        } else {
            slot[0] = 0x1p15f; slot[1] = 0x1p4f; slot[2] = 0x1p-7f;
        }
        const float mul = (is_ref ? -2.0f : 1.0f) * sc;
        double lo2 = 0.0;   // squared norm of the lo parts (scaled units)
        for (int g = 0; g < GRAN / 2; ++g) {
            cmk2::half8 hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                int c = g * 8 + e;
                _Float16 h = (_Float16)0.f, l = (_Float16)0.f;
                if (c < d) {
                    if (live) {
                        float v = (float)((double)x[c] * nscale - mean[c] - (cj ? cj[c] : 0.0)) * mul;
                        h = (_Float16)v;
                        l = (_Float16)(v - (float)h);
                        lo2 = fma((double)(float)l, (double)(float)l, lo2);
                    }
                } else if (!yn_block && c < d + cmk2::NORM_SLOTS) {
                    h = (_Float16)slot[c - d];
                }
                hi[e] = h;
                lo[e] = l;
            }
            *reinterpret_cast<cmk2::half8 *>(rowp + ((g ^ f) * 16)) = hi;
            if (!(is_ref && hi_only)) *reinterpret_cast<cmk2::half8 *>(rowp + (((GRAN / 2 + g) ^ f) * 16)) = lo;
        }
        yn_f = live ? (float)s2 : 0.f;
        if (yn_block && is_ref) {
            float *ynp = reinterpret_cast<float *>(P + tile * TILEB + (size_t)cmk2::BR * ROWB);
            ynp[r] = live ? (float)(s2 * (double)sc * (double)sc) : INFINITY;
            if (r == 0) for (int e = cmk2::BR; e < 256; ++e) ynp[e] = 0.f;
        }
        if (rownorm && i < n) rownorm[i] = sqrtf(yn_f) * 1.0000002f;
        if (rowlo && i < n) rowlo[i] = (float)(sqrt(lo2) / (double)sc) * 1.000001f;
        if (rownorm_c && i < n) rownorm_c[i] = (float)sqrt(s2c) * 1.000001f;
        if (rowcen && i < n) rowcen[i] = (float)sqrt(c2) * 1.000001f;
        if (is_ref && live && s2 > 0.0) lo_ratio = (float)(sqrt(lo2) / (2.0 * (double)sc * sqrt(s2))) * 1.000001f;
    }
    if (is_ref && gmax) {
        float m = yn_f, lr = lo_ratio;
        for (int o = 32; o > 0; o >>= 1) { m = fmaxf(m, __shfl_xor(m, o)); lr = fmaxf(lr, __shfl_xor(lr, o)); }
        if ((threadIdx.x & 63) == 0) {
            atomicMax(&gmax[0], __float_as_uint(m * 1.0000002f));
            atomicMax(&gmax[1], __float_as_uint(sqrtf(m) * 1.0000002f));
            atomicMax(&gmax[5], __float_as_uint(lr));
        }
    }
}

// X [n,d] -> float32 rows [n_pad][d] for the low-d kernel (cosine: normalised); padded reference rows are +inf.
template <typename T>
__global__ void prep_rows_lowd(const T *__restrict__ X, int64_t n, int d, int64_t n_pad, int is_ref, int metric,
                               float *__restrict__ P, int dpad, const int *__restrict__ rowmap) {
    // dpad >= d: row stride of the prepared image (extra columns are zero on both sides); rowmap: prepare row i from
    // source row rowmap[i] (query subsets of the staged search)
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pad) return;
    float *out = P + i * dpad;
    for (int c = d; c < dpad; ++c) out[c] = 0.f;
    if (i < n) {
        const T *x = X + (rowmap ? (int64_t)rowmap[i] : i) * d;
        double nscale = 1.0;
        if (metric == CM_METRIC_COSINE) {
            double s = 0.0;
            for (int c = 0; c < d; ++c) { double v = (double)x[c]; s = fma(v, v, s); }
            double nrm = sqrt(s);
            nscale = nrm > 0.0 ? 1.0 / nrm : 1.0;
        }
        for (int c = 0; c < d; ++c) out[c] = (float)((double)x[c] * nscale);
    } else {
        for (int c = 0; c < d; ++c) out[c] = is_ref ? INFINITY : 0.f;
    }
}

// ------------------------------------------------------------------ exact float64 pair value --------
// The value that orders neighbours and (after sqrt) is returned, following sklearn's arithmetic exactly as the
// CPU oracle restates it (oracle/cm_oracle.c: cmo_knn_euclidean):
//   GEMM form   : (|q|^2 + (-2 * q.y)) + |y|^2 with fma chains, clamped at 0      (EuclideanArgKmin)
//   DIRECT form : sum_j (q_j - y_j)^2, multiply and add rounded separately        (KDTree rdist)
//   cosine      : clip(1 - q.y / (|q| |y|), 0, 2)                                 (pairwise cosine_distances)
//   manhattan / chebyshev : sum / max of |q_j - y_j| in coordinate order           (DistanceMetric dist, see below)
// PairAcc: NQ query rows against one reference row at once -- the same chain per pair, in the same order; the pairs' chains
// are independent, which is what lets the exact scan interleave them, and the state can be carried over column chunks (run is
// called with consecutive column ranges; yat(i) is the reference row's coordinate c0 + i).  pair_value is the NQ = 1 case
// over the whole row.
template <typename T, typename TQ, int NQ>
struct PairAcc {
    double a[NQ], b[NQ], c[NQ], yn;
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int g = 0; g < NQ; ++g) { a[g] = 0.0; b[g] = 0.0; c[g] = 0.0; }
        yn = 0.0;
    }
    // one loop per (metric family, float32-difference rule): MODE 0 q.y and |y|^2 (cosine, GEMM form), 1 direct squared
    // differences, 2 manhattan, 3 chebyshev, 4 canberra, 5 braycurtis, 6 hamming, 7 the boolean counts; F32: the difference is
    // taken in float32 (below)
    template <int MODE, bool F32, int UNR, typename YF>
    __device__ __forceinline__ void run_mode(const TQ *const *q, int c0, int n, YF yat) {
        auto step = [&](int i) {
            const T yv = yat(i);
#pragma unroll
            for (int g = 0; g < NQ; ++g) {
                const TQ qv = q[g][c0 + i];
                if constexpr (MODE == 0) {
                    a[g] = fma((double)qv, (double)yv, a[g]);
                } else if constexpr (MODE == 6) {   // hamming: coordinates that differ (counts are exact in float64)
                    a[g] += qv != yv ? 1.0 : 0.0;
                } else if constexpr (MODE == 7) {   // boolean metrics: ntt, ntf, nft of (x != 0, y != 0)
                    const bool tq = qv != (TQ)0, ty = yv != (T)0;
                    a[g] += (tq && ty) ? 1.0 : 0.0;
                    b[g] += (tq && !ty) ? 1.0 : 0.0;
                    c[g] += (!tq && ty) ? 1.0 : 0.0;
                } else if constexpr (MODE == 1) {
                    const double t = (double)qv - (double)yv;
                    a[g] = __dadd_rn(a[g], __dmul_rn(t, t));
                } else if constexpr (MODE == 8) {   // haversine: column 0 = latitude, 1 = longitude (HaversineDistance.rdist)
                    double df;
                    if constexpr (F32) df = (double)__fsub_rn((float)qv, (float)yv);
                    else df = (double)qv - (double)yv;
                    const double sh = sin(__dmul_rn(0.5, df));
                    if (c0 + i == 0) { a[g] = __dmul_rn(sh, sh); b[g] = __dmul_rn(cos((double)qv), cos((double)yv)); }
                    else if (c0 + i == 1) c[g] = sh;
                } else if constexpr (MODE == 9) {   // nan_euclidean: coordinates present (not NaN) in both rows
                    const double qd = (double)qv, yd = (double)yv;
                    if (qd == qd && yd == yd) { const double t = qd - yd; a[g] = __dadd_rn(a[g], __dmul_rn(t, t)); b[g] += 1.0; }
                } else {
                    double t;
                    if constexpr (F32) t = (double)fabsf(__fsub_rn((float)qv, (float)yv));
                    else t = fabs((double)qv - (double)yv);
                    if constexpr (MODE == 2) a[g] = __dadd_rn(a[g], t);
                    else if constexpr (MODE == 3) a[g] = fmax(a[g], t);
                    else {
                        const double dn = __dadd_rn(fabs((double)qv), fabs((double)yv));
                        if constexpr (MODE == 4) { if (dn > 0.0) a[g] = __dadd_rn(a[g], __ddiv_rn(t, dn)); }
                        else { a[g] = __dadd_rn(a[g], t); b[g] = __dadd_rn(b[g], dn); }
                    }
                }
            }
            if constexpr (MODE == 0) yn = fma((double)yv, (double)yv, yn);
        };
        if constexpr (UNR > 1) {
#pragma unroll UNR
            for (int i = 0; i < n; ++i) step(i);
        } else {
            for (int i = 0; i < n; ++i) step(i);   // (unrolling left to the compiler, as before the accumulator form)
        }
    }
    // manhattan / chebyshev -- sklearn's DistanceMetric: |q_j - y_j| joins a float64 accumulator in coordinate order.  Brute
    // force on float32 data (DistanceMetric32) takes the difference in float32 and returns a float32: one more rounding of the
    // result.  The kd-tree (DIRECT form) works on float64 copies of the data: everything in float64.
    // canberra / braycurtis -- sklearn's CanberraDistance / BrayCurtisDistance `dist` (metrics/_dist_metrics.pyx.tp): float64
    // accumulators in coordinate order; |x_j| + |y_j| in float64; the difference in the INPUT dtype under brute force on
    // float32 data (DistanceMetric32), whose result is rounded to float32 once more; everything float64 on the tree path.
    template <int UNR = 1, typename YF>
    __device__ __forceinline__ void run(const TQ *const *q, int c0, int n, YF yat, int metric, int form) {
        const bool f32 = sizeof(T) == 4 && form != CM_FORM_DIRECT;
        switch (metric) {
            case CM_METRIC_MANHATTAN: if (f32) run_mode<2, true, UNR>(q, c0, n, yat); else run_mode<2, false, UNR>(q, c0, n, yat); return;
            case CM_METRIC_CHEBYSHEV: if (f32) run_mode<3, true, UNR>(q, c0, n, yat); else run_mode<3, false, UNR>(q, c0, n, yat); return;
            case CM_METRIC_CANBERRA: if (f32) run_mode<4, true, UNR>(q, c0, n, yat); else run_mode<4, false, UNR>(q, c0, n, yat); return;
            case CM_METRIC_BRAYCURTIS: if (f32) run_mode<5, true, UNR>(q, c0, n, yat); else run_mode<5, false, UNR>(q, c0, n, yat); return;
            case CM_METRIC_HAMMING: run_mode<6, false, UNR>(q, c0, n, yat); return;
            case CM_METRIC_HAVERSINE: if (f32) run_mode<8, true, UNR>(q, c0, n, yat); else run_mode<8, false, UNR>(q, c0, n, yat); return;
            case CM_METRIC_NAN_EUCLIDEAN: run_mode<9, false, UNR>(q, c0, n, yat); return;
            case CM_METRIC_JACCARD: case CM_METRIC_DICE: case CM_METRIC_ROGERSTANIMOTO: case CM_METRIC_RUSSELLRAO:
            case CM_METRIC_SOKALMICHENER: case CM_METRIC_SOKALSNEATH: case CM_METRIC_YULE: run_mode<7, false, UNR>(q, c0, n, yat); return;
            default: break;
        }
        if (metric != CM_METRIC_COSINE && form == CM_FORM_DIRECT) run_mode<1, false, UNR>(q, c0, n, yat);
        else run_mode<0, false, UNR>(q, c0, n, yat);   // cosine and the GEMM form: q.y per pair, |y|^2 once
    }
    // d: the row width (the counting metrics' n)
    __device__ __forceinline__ void finish(const double *xn, int metric, int form, double *out, int d) const {
        const bool f32 = sizeof(T) == 4 && form != CM_FORM_DIRECT;
#pragma unroll
        for (int g = 0; g < NQ; ++g) {
            if (metric == CM_METRIC_HAVERSINE) {
                // sin_0 * sin_0 + cos(x0) * cos(y0) * sin_1 * sin_1, evaluated left to right as the C source is; brute force on
                // float32 data (HaversineDistance32.rdist returns its INPUT dtype) rounds the bracket to float32 before the
                // conversion 2 asin(sqrt(.)) -- which ArgKmin applies in float64 at the end, on either path
                double r = __dadd_rn(a[g], __dmul_rn(__dmul_rn(b[g], c[g]), c[g]));
                if (f32) r = (double)(float)r;
                out[g] = __dmul_rn(2.0, asin(sqrt(r)));
            } else if (metric == CM_METRIC_NAN_EUCLIDEAN) {
                // nan_euclidean_distances: clip at 0, / #present, * d, sqrt; no shared coordinate: NaN.  float32 data: float32 out
                const double r = b[g] > 0.0 ? sqrt(__dmul_rn(__ddiv_rn(a[g], b[g]), (double)d)) : __longlong_as_double(0x7ff8000000000000LL);
                out[g] = sizeof(T) == 4 ? (double)(float)r : r;
            } else if (metric >= CM_METRIC_HAMMING) {
                // scipy's cdist kernels (spatial/src/distance_impl.h) and sklearn's DistanceMetric classes: one float64 division
                // of exact integer counts
                const double n = (double)d, ntt = a[g], ntf = b[g], nft = c[g], R = ntf + nft, nff = n - ntt - R;
                double r;
                switch (metric) {
                    case CM_METRIC_HAMMING: r = __ddiv_rn(a[g], n); break;
                    case CM_METRIC_JACCARD: r = (ntt + R) > 0.0 ? __ddiv_rn(R, ntt + R) : 0.0; break;
                    case CM_METRIC_DICE: r = __ddiv_rn(R, 2.0 * ntt + R); break;   // (0 / 0 = NaN, as upstream)
                    case CM_METRIC_RUSSELLRAO: r = __ddiv_rn(n - ntt, n); break;
                    case CM_METRIC_SOKALSNEATH: r = __ddiv_rn(2.0 * R, ntt + 2.0 * R); break;
                    case CM_METRIC_YULE: r = (ntf * nft) > 0.0 ? __ddiv_rn(2.0 * (ntf * nft), ntt * nff + ntf * nft) : 0.0; break;
                    default: r = __ddiv_rn(2.0 * R, ntt + nff + 2.0 * R); break;   // rogerstanimoto, sokalmichener
                }
                out[g] = r;
            } else if (metric == CM_METRIC_COSINE) {
                double nx = sqrt(xn[g]), ny = sqrt(yn);
                if (!(nx > 0.0)) nx = 1.0;
                if (!(ny > 0.0)) ny = 1.0;
                const double s = 1.0 - a[g] / (nx * ny);
                out[g] = s < 0.0 ? 0.0 : (s > 2.0 ? 2.0 : s);
            } else if (metric == CM_METRIC_MANHATTAN || metric == CM_METRIC_CHEBYSHEV) {
                out[g] = f32 ? (double)(float)a[g] : a[g];
            } else if (metric == CM_METRIC_CANBERRA || metric == CM_METRIC_BRAYCURTIS) {
                const double r = metric == CM_METRIC_CANBERRA ? a[g] : (b[g] > 0.0 ? __ddiv_rn(a[g], b[g]) : 0.0);
                out[g] = f32 ? (double)(float)r : r;
            } else if (form == CM_FORM_DIRECT) {
                out[g] = a[g];
            } else {
                const double d2 = __dadd_rn(__dadd_rn(xn[g], __dmul_rn(-2.0, a[g])), yn);
                out[g] = d2 > 0.0 ? d2 : 0.0;
            }
        }
    }
};
template <typename T, typename TQ = double>
__device__ __forceinline__ double pair_value(const TQ *__restrict__ q, double xn, const T *__restrict__ y, int d,
                                             int metric, int form) {
    const TQ *const qa[1] = {q};
    PairAcc<T, TQ, 1> acc;
    acc.init();
    acc.run(qa, 0, d, [&](int c) { return y[c]; }, metric, form);
    double out[1];
    acc.finish(&xn, metric, form, out, d);
    return out[0];
}

__device__ __forceinline__ bool pair_before(double va, int64_t ia, double vb, int64_t ib) {
    return va < vb || (va == vb && ia < ib);
}

// ------------------------------------------------------------------ phase 2: re-rank + certify ------
// One wave per query.  Candidates of all reference splits are re-evaluated in float64, ranked by (value, index),
// the first k are written out.  Certification: every non-candidate has fp32 ranking value >= tau; with the fp32
// error bound eps_q, the float64 top-k is provably the exact one iff  s_k < tau - eps_q  (s = d^2 - |q|^2).
// GQ (rows wider than KNN_DMAX, round 4): the query row is read from global memory in its own type instead of a float64 copy
// in LDS (the conversion is exact, the arithmetic the same); the three kernels that stage query rows have this form.
template <typename T, bool GQ = false>
__global__ __launch_bounds__(256) void knn_rerank(const T *__restrict__ ref, const T *__restrict__ qry, int64_t nq,
                                                   int d, const int *__restrict__ cand, const float *__restrict__ cthr,
                                                   int S, int Kc, int k, int metric, int form,
                                                   const float *__restrict__ qnorm, const unsigned *__restrict__ gmax,
                                                   float c_eps, float c_split, const float *__restrict__ scale3,
                                                   const double *__restrict__ mean, int64_t idx_offset,
                                                   int flags, void *__restrict__ out_dist,
                                                   int64_t *__restrict__ out_idx, int *__restrict__ flag_list,
                                                   int *__restrict__ flag_cnt, double *__restrict__ bound,
                                                   const int *__restrict__ qmap, int pilot_kc,
                                                   const float *__restrict__ qlo, int level, const int *__restrict__ rmap,
                                                   const float *__restrict__ qnc = nullptr,
                                                   const float *__restrict__ qcen = nullptr, int ncap = 0,
                                                   float *__restrict__ seedb = nullptr) {
    // seedb != nullptr (staged search): `bound` is initialised (+inf) and kept MONOTONE over the stages (any stage's k-th
    // candidate value bounds the true k-th value from above), and a flagged query leaves a starting threshold for its next
    // stage in seedb[io]: bound - |q'|^2 + twice this stage's filter error, in filter units -- the k candidates behind the
    // bound have filter values below it on every level whose error does not exceed this one's, with room for that level's
    // certificate.
    // ncap > 0 (bucket pruning: many lists per query, most of them empty): the LDS holds ncap candidates; the valid
    // entries of the lists are compacted first, a query with more than ncap of them is left to the next stage.
    // qnc / qcen != nullptr (bucket-centred filter): |q' - c| and |c| of every packed row; qmap entries < 0 are padding.
    // qmap != nullptr: the nq rows of cand / cthr / qnorm belong to the query SUBSET qmap[0..nq) (staged search);
    // the query row, the outputs, the bound and the flag entry use the original index io.
    // pilot_kc > 0 (pilot of the staged search): a small sample runs with many reference splits, whose union of
    // candidate lists certifies far more easily than the single list of pilot_kc candidates the full run will keep.
    // The flag then answers "would a single list of pilot_kc candidates have been certified?": the threshold is
    // emulated by the exact value of the pilot_kc-th best candidate of the union, less the filter error.
    // dynamic LDS sized by the candidate count (rerank_lds_bytes): a fixed RR_NCMAX-sized layout holds the kernel to two
    // workgroups per CU, too few to hide the latency of the candidate gathers
    extern __shared__ __align__(16) unsigned char rr_smem[];
    const int cap = ncap > 0 ? ncap : S * Kc;
    const int ncp = (cap + 1) & ~1;
    const int qs = GQ ? 0 : q_stride(d);
    double *s_qb = reinterpret_cast<double *>(rr_smem);   // [4][qs]: the four waves' query rows
    double *s_vb = reinterpret_cast<double *>(rr_smem + sizeof(double) * 4 * qs);
    int *s_ib = reinterpret_cast<int *>(s_vb + 4 * ncp);
    double *s_kth = reinterpret_cast<double *>(s_ib + 4 * ncp);
    double *s_kc = s_kth + 4;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    struct Row2 { double *v; int *i; };
    const Row2 s_vw{s_vb + (size_t)w * ncp, s_ib + (size_t)w * ncp};
    const int64_t i = (int64_t)blockIdx.x * 4 + w;
    if (i >= nq) return;
    if (lane == 0) { s_kc[w] = INFINITY; s_kth[w] = INFINITY; }   // (fewer than k valid candidates: bound = inf)
    const int64_t io = qmap ? (int64_t)qmap[i] : i;
    if (io < 0) return;   // padding row of the bucket-aligned packing
    using TQ = typename std::conditional<GQ, T, double>::type;
    const TQ *q;
    if constexpr (GQ) {
        q = qry + io * d;
    } else {
        double *ql = s_qb + (size_t)w * qs;
        for (int c = lane; c < d; c += 64) ql[c] = (double)qry[io * d + c];
        q = ql;
    }
    __builtin_amdgcn_wave_barrier();
    double xn = 0.0;
    for (int c = 0; c < d; ++c) xn = fma((double)q[c], (double)q[c], xn);
    const int nc_all = S * Kc;
    const int *cq = cand + i * nc_all;
    int nc = 0;   // valid candidates, compacted to the front
    for (int s0 = 0; s0 < nc_all; s0 += 64) {
        const int s = s0 + lane;
        const int j = s < nc_all ? cq[s] : -1;
        const bool ok = j >= 0;
        const unsigned long long m = __ballot(ok);
        const int p = nc + __popcll(m & ((1ull << lane) - 1ull));
        if (ok && p < cap) s_vw.i[p] = j;
        nc += __popcll(m);
    }
    const bool over = nc > cap;
    if (over) nc = cap;
    __builtin_amdgcn_wave_barrier();
    for (int s = lane; s < nc; s += 64) {
        int j = s_vw.i[s];
        if (rmap) j = rmap[j];   // candidate positions refer to the locality-sorted tile image
        double v = INFINITY;
        if (j >= 0) v = pair_value<T, TQ>(q, xn, ref + (int64_t)j * d, d, metric, form);
        s_vw.v[s] = v;
        s_vw.i[s] = j >= 0 ? j : 0x7fffffff;
    }
    __builtin_amdgcn_wave_barrier();
    for (int s = lane; s < nc; s += 64) {
        double v = s_vw.v[s];
        int id = s_vw.i[s];
        int rk = 0;
        for (int j = 0; j < nc; ++j) rk += pair_before(s_vw.v[j], s_vw.i[j], v, id) ? 1 : 0;
        if (rk < k) {
            double o = v;
            if (metric == CM_METRIC_EUCLIDEAN && !(flags & CM_KNN_SQUARED)) o = sqrt(v);
            if (metric == CM_METRIC_COSINE && !(flags & CM_KNN_F64_OUT)) reinterpret_cast<T *>(out_dist)[io * k + rk] = (T)o;
            else reinterpret_cast<double *>(out_dist)[io * k + rk] = o;
            out_idx[io * k + rk] = (int64_t)id + idx_offset;
            if (rk == k - 1) s_kth[w] = v;
        }
        if (pilot_kc > 0 && rk == pilot_kc - 1) s_kc[w] = v;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        double kth = s_kth[w];
        float tau = INFINITY;
        for (int s = 0; s < S; ++s) tau = fminf(tau, cthr[i * S + s]);
        if (scale3) tau *= scale3[2];  // phase-1 values were computed on rows scaled by sc: back to input units
        bool certified = true;
        double seed_eps = -1.0, seed_qn2 = 0.0;   // (set by the MFMA-level branch: what the next stage's seed needs)
        if (tau < INFINITY && c_eps >= 0.f) {
            // phase 1 ranked by |y'|^2 - 2 q'.y' = D - |q'|^2 with D = d^2 (cosine: 2*dist) and q', y' the rows as
            // prepared for the filter (cosine: normalised; split-fp16 path: centred at the reference mean)
            double qn2 = xn;
            if (metric == CM_METRIC_COSINE || mean) {
                double ns = 1.0;
                if (metric == CM_METRIC_COSINE) { double nx = sqrt(xn); ns = nx > 0.0 ? 1.0 / nx : 1.0; }
                qn2 = 0.0;
                for (int c = 0; c < d; ++c) { double v = (double)q[c] * ns - (mean ? mean[c] : 0.0); qn2 = fma(v, v, qn2); }
            }
            double sk = ((metric == CM_METRIC_COSINE) ? 2.0 * kth : kth) - qn2;
            double ymax = (double)__uint_as_float(gmax[1]);
            // Only a non-candidate with true value <= s_k could invalidate the result; such a row lies within the k-th
            // candidate's distance of the query, so |y'| <= |q'| + sqrt(D_k): the filter error that matters is bounded
            // with that norm instead of the global maximum (much tighter for queries near the centre of the data).
            const double qn = (double)qnorm[i];
            const double dk = sqrt(fmax(sk + qn2, 0.0)) * (1.0 + 1e-7);
            if (qn + dk < ymax) ymax = qn + dk;
            const double ymax2 = ymax * ymax;
            // c_eps: accumulation / representation error on every term; c_split: split products that were dropped
            double eps = (double)c_eps * (ymax2 + 2.0 * qn * ymax + 1e-30);
            if (qnc) {
                // bucket-centred filter: the MFMA chain accumulates |y'|^2 digits, the side value -2c.y' (float32, one
                // rounding of a float64 dot product) and the products of q'' = q' - c with y': magnitudes
                // Y^2 + 2|c|Y + 2|q''|Y; the dropped lo products involve q'' only
                const double qc = (double)qnc[i], cn = (double)qcen[i];
                const double rho_y = (double)__uint_as_float(gmax[5]);
                eps = (double)c_eps * (ymax2 + 2.0 * (cn + qc) * ymax + 1e-30) + 2.4e-7 * cn * ymax +
                      2.0 * 1.001 * ymax * ((level == 1 ? (double)qlo[i] : 0.0) + (level <= 2 ? qc * rho_y : 0.0));
            } else if (qlo) {
                // dropped split products with the ACTUAL lo norms (Cauchy-Schwarz): level 1 drops lo(q).hi(y) and
                // hi(q).lo(y), level 2 only the latter; |hi(.)| <= (1 + 2^-11)|.|; gmax[5] = max |lo(y')| / |y'|
                const double rho_y = (double)__uint_as_float(gmax[5]);
                eps += 2.0 * 1.001 * ymax * ((level == 1 ? (double)qlo[i] : 0.0) + (level <= 2 ? qn * rho_y : 0.0));
            } else {
                eps += (double)c_split * 2.0 * qn * ymax;
            }
            seed_eps = eps; seed_qn2 = qn2;
            double tau_d = (double)tau;
            if (pilot_kc > 0) {
                const double vkc = s_kc[w];
                tau_d = vkc < INFINITY ? ((metric == CM_METRIC_COSINE) ? 2.0 * vkc : vkc) - qn2 - eps : INFINITY;
            }
            certified = sk < tau_d - eps;
        } else if (tau < INFINITY) {
            // low-d kernel: phase-1 value is the full direct squared distance sum (q-y)^2 in fp32; relative error
            // -c_eps, plus the rounding of float64 inputs to fp32 (2^-24 * max|coordinate| per coordinate)
            double sk = (metric == CM_METRIC_COSINE) ? 2.0 * kth : kth;
            // GEMM-form distances requested for d <= 15 (k >= n_ref/2): the float64 GEMM-form value differs from the
            // direct-form one by ~1e-16 (|q|^2+|y|^2); inflate conservatively
            if (metric != CM_METRIC_COSINE && form != CM_FORM_DIRECT) sk = kth * (1.0 + 1e-9) + 1e-9 * xn;
            double crel = (double)(-c_eps);
            double dlt = sizeof(T) == 8 ? 2.0 * 5.9604645e-8 * (double)__uint_as_float(gmax[4]) : 0.0;
            double big = fmax(sk, (double)tau);
            double eps = crel * big + 2.0 * dlt * sqrt((double)d) * sqrt(big) + (double)d * dlt * dlt;
            // sum / max of |q_j - y_j|: a coordinate rounded by dlt moves a term by at most dlt (two coordinates per term);
            // the float32 rounding of the reported value (DistanceMetric32) is inside crel
            if (metric == CM_METRIC_MANHATTAN) eps = crel * big + 2.0 * dlt * (double)d;
            if (metric == CM_METRIC_CHEBYSHEV) eps = crel * big + 2.0 * dlt;
            // Canberra: every term is a ratio in [0, 1]; float32 subtract, add and approximate divide cost <= 4 ulp per term and
            // the running sum d more: covered by crel = (d + 4) 2^-23 relative to the value (terms are non-negative).  A float64
            // coordinate rounded to float32 (relative 2^-24; magnitudes below 2^-100 are refused by the host) moves a term by at
            // most ~2^-23 ABSOLUTE: d 2^-22 in all.  Bray-Curtis: numerator and denominator each move by 2^-24 of the denominator
            // under that rounding, the ratio by <= 2^-22 absolute; the two float32 sums are inside crel.
            if (metric == CM_METRIC_CANBERRA) eps = crel * big + (sizeof(T) == 8 ? (double)d * 2.4e-7 : 0.0);
            if (metric == CM_METRIC_BRAYCURTIS) eps = crel * big + (sizeof(T) == 8 ? 2.4e-7 : 0.0);
            certified = sk + eps < (double)tau - eps;
        }
        if (!certified || over || (flags & CM_KNN_FORCE_FALLBACK)) {
            int pos = atomicAdd(flag_cnt, 1);
            flag_list[pos] = (int)io;
            if (over) atomicAdd(flag_cnt + 21, 1);            // (words[23], [24]: why -- diagnostics, CELLMAP_DBG=16)
            else if (nc < k) atomicAdd(flag_cnt + 22, 1);
            if (seedb) {
                const double b = fmin(bound[io], kth);
                bound[io] = b;
                if (!(b < INFINITY)) atomicAdd(flag_cnt + 20, 1);   // (words[22]: flagged without a finite bound -- fewer than k candidates so far)
                float sv = INFINITY;
                if (b < INFINITY && seed_eps >= 0.0) {
                    // (2 x the filter error: the next stage's certificate needs s_k + its own error below the threshold, and the
                    //  threshold cannot exceed the seed -- its error is at most this stage's on the levels that are seeded)
                    double x = (((metric == CM_METRIC_COSINE) ? 2.0 * b : b) - seed_qn2 + 2.0 * seed_eps);
                    x += 1e-6 * fabs(x) + 1e-30;
                    if (scale3) x /= (double)scale3[2];
                    sv = (float)x;
                    if ((double)sv < x) sv = nextafterf(sv, INFINITY);
                }
                seedb[io] = sv;
            } else {
                bound[io] = kth;
                if (!(kth < INFINITY)) atomicAdd(flag_cnt + 20, 1);
            }
        }
    }
}

static inline size_t rerank_lds_bytes(int S, int Kc, int d = 0) {
    const int ncp = (S * Kc + 1) & ~1;
    return sizeof(double) * 4 * (d > KNN_DMAX ? 0 : q_stride(d)) + (size_t)4 * ncp * (sizeof(double) + sizeof(int)) + 64;
}

// a query whose collection overflowed FB_CAP: handed to knn_fallback_select (ovf = words + 3: ovf[0] = flag "list
// full", ovf[OVF_SLOT] = words[20] = count; words[4..] up to the scales belong to the row preparation)
constexpr int OVF_SLOT = 17;
__device__ __forceinline__ void push_overflow(int *ovf, int *ovf_list, int64_t i) {
    const int pos = atomicAdd(ovf + OVF_SLOT, 1);
    if (pos < OVF_CAP) ovf_list[pos] = (int)i;
    else atomicExch(ovf, 1);
}

// ------------------------------------------------------------------ exact fallback ------------------
// For queries that could not be certified: one workgroup scans ALL reference rows in float64 (same arithmetic as
// phase 2), collects every row whose value is <= the k-th candidate value (an upper bound of the true k-th),
// ranks the collection and overwrites the query's output row.  Rare by construction.
template <typename T, bool GQ = false>
__global__ __launch_bounds__(256) void knn_exact_fallback(const T *__restrict__ ref, int64_t n_ref,
                                                           const T *__restrict__ qry, int d, int k, int metric, int form,
                                                           int64_t idx_offset, int flags, const int *__restrict__ flag_list,
                                                           const int *__restrict__ flag_cnt,
                                                           const double *__restrict__ bound, void *__restrict__ out_dist,
                                                           int64_t *__restrict__ out_idx, int *__restrict__ overflow,
                                                           int *__restrict__ ovf_list, int f_start) {
    extern __shared__ __align__(16) unsigned char fb_smem[];
    double *s_q = reinterpret_cast<double *>(fb_smem);   // [q_stride(d)] (dynamic: wide rows)
    __shared__ double s_v[FB_CAP];
    __shared__ int s_i[FB_CAP];
    __shared__ int s_n;
    const int nf = *flag_cnt;
    for (int f = f_start + blockIdx.x; f < nf; f += gridDim.x) {
        const int64_t i = flag_list[f];
        __syncthreads();
        using TQ = typename std::conditional<GQ, T, double>::type;
        const TQ *q;
        if constexpr (GQ) {
            q = qry + i * d;
        } else {
            for (int c = threadIdx.x; c < d; c += blockDim.x) s_q[c] = (double)qry[i * d + c];
            q = s_q;
        }
        if (threadIdx.x == 0) s_n = 0;
        __syncthreads();
        double xn = 0.0;
        for (int c = 0; c < d; ++c) xn = fma((double)q[c], (double)q[c], xn);
        const double b = bound[i];
        for (int64_t j = threadIdx.x; j < n_ref; j += blockDim.x) {
            double v = pair_value<T, TQ>(q, xn, ref + j * d, d, metric, form);
            if (v <= b) {
                int pos = atomicAdd(&s_n, 1);
                if (pos < FB_CAP) { s_v[pos] = v; s_i[pos] = (int)j; }
            }
        }
        __syncthreads();
        int m = s_n;
        if (m > FB_CAP) { if (threadIdx.x == 0) push_overflow(overflow, ovf_list, i); continue; }   // (uniform)
        for (int s = threadIdx.x; s < m; s += blockDim.x) {
            double v = s_v[s];
            int id = s_i[s];
            int rk = 0;
            for (int j = 0; j < m; ++j) rk += pair_before(s_v[j], s_i[j], v, id) ? 1 : 0;
            if (rk < k) {
                double o = v;
                if (metric == CM_METRIC_EUCLIDEAN && !(flags & CM_KNN_SQUARED)) o = sqrt(v);
                if (metric == CM_METRIC_COSINE && !(flags & CM_KNN_F64_OUT)) reinterpret_cast<T *>(out_dist)[i * k + rk] = (T)o;
                else reinterpret_cast<double *>(out_dist)[i * k + rk] = o;
                out_idx[i * k + rk] = (int64_t)id + idx_offset;
            }
        }
    }
}

// Sliced variant for the first FBS_NQ flagged queries.  One workgroup per query (above) needs ~50 ms for a 1.5M-row
// reference, whatever the number of flagged queries; here workgroup (x, y) scans reference slice x for the flagged
// queries y, y + gridDim.y, ... and appends the rows within the bound to the query's global candidate list;
// knn_fallback_rank then ranks each list exactly as knn_exact_fallback does.  Same arithmetic (pair_value), same
// (value, index) order, same overflow rule.
template <typename T>
__global__ __launch_bounds__(256) void knn_fallback_scan(const T *__restrict__ ref, int64_t n_ref, const T *__restrict__ qry,
                                                         int d, int metric, int form, const int *__restrict__ flag_list,
                                                         const int *__restrict__ flag_cnt, const double *__restrict__ bound,
                                                         int *__restrict__ gcount, double *__restrict__ gv,
                                                         int *__restrict__ gi, int rc) {
    // Rows are staged through LDS in chunks of rc rows (coalesced copy; one thread per row afterwards, which keeps
    // pair_value's sequential summation order) and every staged chunk serves FBS_QG flagged queries.
    extern __shared__ __align__(16) unsigned char fbs_smem[];
    const int qs_ = q_stride(d);
    double *s_qb = reinterpret_cast<double *>(fbs_smem);                          // [FBS_QG][qs_]: the flagged queries' rows
    T *s_rows = reinterpret_cast<T *>(fbs_smem + sizeof(double) * FBS_QG * qs_);  // [rc][d]
    int nf = *flag_cnt;
    if (nf > FBS_NQ) nf = FBS_NQ;
    const int64_t per = (n_ref + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < n_ref ? lo + per : n_ref;
    for (int f0 = blockIdx.y * FBS_QG; f0 < nf; f0 += gridDim.y * FBS_QG) {
        const int ng = nf - f0 < FBS_QG ? nf - f0 : FBS_QG;
        __syncthreads();
        for (int t = threadIdx.x; t < ng * d; t += blockDim.x) {
            const int g = t / d, c = t - g * d;
            s_qb[(size_t)g * qs_ + c] = (double)qry[(int64_t)flag_list[f0 + g] * d + c];
        }
        __syncthreads();
        double xn[FBS_QG], b[FBS_QG];
#pragma unroll
        for (int g = 0; g < FBS_QG; ++g) {
            xn[g] = 0.0;
            b[g] = -1.0;
            if (g < ng) {
                for (int c = 0; c < d; ++c) xn[g] = fma(s_qb[(size_t)g * qs_ + c], s_qb[(size_t)g * qs_ + c], xn[g]);
                b[g] = bound[flag_list[f0 + g]];
            }
        }
        for (int64_t base = lo; base < hi; base += rc) {
            const int nr = hi - base < rc ? (int)(hi - base) : rc;
            __syncthreads();
            for (int t = threadIdx.x; t < nr * d; t += blockDim.x) s_rows[t] = ref[base * d + t];
            __syncthreads();
            for (int r = threadIdx.x; r < nr; r += blockDim.x) {
#pragma unroll
                for (int g = 0; g < FBS_QG; ++g) {
                    if (g < ng) {
                        const double v = pair_value<T>(s_qb + (size_t)g * qs_, xn[g], s_rows + (size_t)r * d, d, metric, form);
                        if (v <= b[g]) {
                            const int pos = atomicAdd(&gcount[f0 + g], 1);
                            if (pos < FB_CAP) {
                                gv[(size_t)(f0 + g) * FB_CAP + pos] = v;
                                gi[(size_t)(f0 + g) * FB_CAP + pos] = (int)(base + r);
                            }
                        }
                    }
                }
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void knn_fallback_rank(int k, int metric, int64_t idx_offset, int flags,
                                                         const int *__restrict__ flag_list, const int *__restrict__ flag_cnt,
                                                         const int *__restrict__ gcount, const double *__restrict__ gv,
                                                         const int *__restrict__ gi, void *__restrict__ out_dist,
                                                         int64_t *__restrict__ out_idx, int *__restrict__ overflow,
                                                         int *__restrict__ ovf_list) {
    __shared__ double s_v[FB_CAP];
    __shared__ int s_i[FB_CAP];
    int nf = *flag_cnt;
    if (nf > FBS_NQ) nf = FBS_NQ;
    for (int f = blockIdx.x; f < nf; f += gridDim.x) {
        const int64_t i = flag_list[f];
        int m = gcount[f];
        if (m > FB_CAP) { if (threadIdx.x == 0) push_overflow(overflow, ovf_list, i); continue; }   // (uniform)
        __syncthreads();
        for (int s = threadIdx.x; s < m; s += blockDim.x) { s_v[s] = gv[(size_t)f * FB_CAP + s]; s_i[s] = gi[(size_t)f * FB_CAP + s]; }
        __syncthreads();
        for (int s = threadIdx.x; s < m; s += blockDim.x) {
            double v = s_v[s];
            int id = s_i[s];
            int rk = 0;
            for (int j = 0; j < m; ++j) rk += pair_before(s_v[j], s_i[j], v, id) ? 1 : 0;
            if (rk < k) {
                double o = v;
                if (metric == CM_METRIC_EUCLIDEAN && !(flags & CM_KNN_SQUARED)) o = sqrt(v);
                if (metric == CM_METRIC_COSINE && !(flags & CM_KNN_F64_OUT)) reinterpret_cast<T *>(out_dist)[i * k + rk] = (T)o;
                else reinterpret_cast<double *>(out_dist)[i * k + rk] = o;
                out_idx[i * k + rk] = (int64_t)id + idx_offset;
            }
        }
    }
}

// More than FB_CAP reference rows inside a query's bound (thousands of rows tied with its k-th neighbour: duplicated
// cells): no collection at all -- k passes over the reference, pass r keeps the smallest (value, index) pair that comes
// after the pair of pass r - 1.  Same arithmetic (pair_value) and the same order as everywhere else; k x n_ref x d
// operations per query, for the handful of queries that get here.  One workgroup per query.
template <typename T, bool GQ = false>
__global__ __launch_bounds__(256) void knn_fallback_select(const T *__restrict__ ref, int64_t n_ref, const T *__restrict__ qry, int d,
                                                           int k, int metric, int form, int64_t idx_offset, int flags,
                                                           const int *__restrict__ ovf, const int *__restrict__ ovf_list,
                                                           void *__restrict__ out_dist, int64_t *__restrict__ out_idx) {
    extern __shared__ __align__(16) unsigned char fsel_smem[];
    double *s_q = reinterpret_cast<double *>(fsel_smem);   // [q_stride(d)]
    __shared__ double s_rv[4];
    __shared__ long long s_ri[4];
    int n = ovf[OVF_SLOT];
    if (n > OVF_CAP) n = OVF_CAP;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int f = blockIdx.x; f < n; f += gridDim.x) {
        const int64_t i = ovf_list[f];
        __syncthreads();
        using TQ = typename std::conditional<GQ, T, double>::type;
        const TQ *q;
        if constexpr (GQ) {
            q = qry + i * d;
        } else {
            for (int c = threadIdx.x; c < d; c += blockDim.x) s_q[c] = (double)qry[i * d + c];
            q = s_q;
        }
        __syncthreads();
        double xn = 0.0;
        for (int c = 0; c < d; ++c) xn = fma((double)q[c], (double)q[c], xn);
        double pv = 0.0;
        long long pj = -1;
        for (int rk = 0; rk < k; ++rk) {
            double bv = INFINITY;
            long long bj = 0x7fffffffffffffffLL;
            for (int64_t j = threadIdx.x; j < n_ref; j += blockDim.x) {
                const double v = pair_value<T, TQ>(q, xn, ref + j * d, d, metric, form);
                if ((rk == 0 || pair_before(pv, pj, v, j)) && pair_before(v, j, bv, bj)) { bv = v; bj = j; }
            }
#pragma unroll
            for (int sft = 32; sft > 0; sft >>= 1) {
                const double ov = __shfl_xor(bv, sft);
                const long long oj = __shfl_xor(bj, sft);
                if (pair_before(ov, oj, bv, bj)) { bv = ov; bj = oj; }
            }
            __syncthreads();   // (the previous pass's minima have been read)
            if (lane == 0) { s_rv[w] = bv; s_ri[w] = bj; }
            __syncthreads();
            bv = s_rv[0]; bj = s_ri[0];
#pragma unroll
            for (int u = 1; u < 4; ++u)
                if (pair_before(s_rv[u], s_ri[u], bv, bj)) { bv = s_rv[u]; bj = s_ri[u]; }
            if (threadIdx.x == 0) {
                double o = bv;
                if (metric == CM_METRIC_EUCLIDEAN && !(flags & CM_KNN_SQUARED)) o = sqrt(bv);
                if (metric == CM_METRIC_COSINE && !(flags & CM_KNN_F64_OUT)) reinterpret_cast<T *>(out_dist)[i * k + rk] = (T)o;
                else reinterpret_cast<double *>(out_dist)[i * k + rk] = o;
                out_idx[i * k + rk] = (int64_t)bj + idx_offset;
            }
            pv = bv; pj = bj;
        }
    }
}

// ------------------------------------------------------------------ exact scan (no filter) ----------
// Shapes no filter family serves (round 4: the term-wise metrics beyond 128 columns; k beyond the candidate buffers of the
// filters): every (query, reference row) value in float64 with pair_value -- the arithmetic and the
// (value, index) order of the re-rank, so nothing needs certifying -- and a running top-k per query.  A workgroup owns QG
// queries; thread t evaluates reference row base + t against each of them (query rows straight from global memory: uniform
// addresses, any width), a value within the query's current k-th value joins its LDS list, a list within 256 entries of its
// capacity is cut back to its k smallest (rank by counting, written in order), which also gives the next k-th value.
// Reference rows reach the threads through LDS: XS_T rows x 128 bytes of columns per step, loaded along the rows (coalesced)
// and read back one row per thread; the queries' columns of the step sit next to them (broadcast reads); the pair accumulators
// (PairAcc) carry over the column steps.  The cut's scratch shares that buffer (a cut never runs inside a row block).
constexpr int XS_T = 256;
template <typename T> __host__ __device__ constexpr int xs_dc() { return 128 / (int)sizeof(T); }   // columns per step
template <typename T> __host__ __device__ constexpr int xs_dcp() { return xs_dc<T>() + 16 / (int)sizeof(T); }   // row stride: 36 dwords (16-byte reads, no bank conflicts)
template <typename T> __host__ __device__ constexpr size_t xs_chunk_bytes() { return (size_t)(XS_T * xs_dcp<T>() + 4 * xs_dc<T>()) * sizeof(T); }   // rows + the queries' columns
template <typename T>
static inline size_t exact_scan_lds_bytes(int QG, int CAP) {
    size_t xb = (size_t)CAP * 12 > xs_chunk_bytes<T>() ? (size_t)CAP * 12 : xs_chunk_bytes<T>();
    xb = (xb + 15) & ~(size_t)15;
    return (size_t)QG * CAP * 12 + 128 + xb + 32;
}
template <typename T, int QG, bool VEC>
__global__ __launch_bounds__(XS_T, 2) void knn_exact_scan(const T *__restrict__ ref, int64_t n_ref, const T *__restrict__ qry,
                                                       int64_t nq, int d, int k, int metric, int form, int64_t idx_offset,
                                                       int flags, int CAP, void *__restrict__ out_dist,
                                                       int64_t *__restrict__ out_idx, int k_total, int k_off,
                                                       double *__restrict__ lb_v, int *__restrict__ lb_i, int have_lb) {
    // k_total / k_off / lb_* (round 5: any k): a k beyond what the on-chip lists hold is served in PASSES of k entries each --
    // this launch writes ranks [k_off, k_off + k) of rows of k_total entries, and ranks only pairs strictly after the query's
    // lower bound (lb_v, lb_i) in the (value, index) order, i.e. after the last entry the previous pass wrote; it leaves its
    // own last entry there for the next pass.  One pass (k_total = k, k_off = 0, no bound) is the round-4 kernel.
    constexpr int DC = xs_dc<T>(), DCP = xs_dcp<T>();
    extern __shared__ __align__(16) unsigned char xs_smem[];
    size_t xb = (size_t)CAP * 12 > xs_chunk_bytes<T>() ? (size_t)CAP * 12 : xs_chunk_bytes<T>();
    xb = (xb + 15) & ~(size_t)15;
    double *s_v = reinterpret_cast<double *>(xs_smem);   // [QG][CAP]: the lists' values
    double *s_thr = s_v + (size_t)QG * CAP;              // [8] current k-th value per query
    double *s_xn = s_thr + 8;                            // [8] |q|^2 (sequential fma chain, as the re-rank computes it)
    unsigned char *s_b = reinterpret_cast<unsigned char *>(s_xn + 8);   // the row block's column step / the cut's scratch
    T *s_y = reinterpret_cast<T *>(s_b);                 // [XS_T][DCP]
    T *s_qc = s_y + XS_T * DCP;                          // [QG][DC]
    double *t_v = reinterpret_cast<double *>(s_b);       // [CAP]
    int *t_i = reinterpret_cast<int *>(t_v + CAP);       // [CAP]
    int *s_i = reinterpret_cast<int *>(s_b + xb);        // [QG][CAP]
    int *s_n = s_i + (size_t)QG * CAP;                   // [8]
    const int tid = threadIdx.x;
    const int64_t q0 = (int64_t)blockIdx.x * QG;
    const int ng = nq - q0 < QG ? (int)(nq - q0) : QG;
    if (tid < 8) { s_n[tid] = 0; s_thr[tid] = INFINITY; }
    if (tid < ng) {
        const T *q = qry + (q0 + tid) * d;
        double xn = 0.0;
        for (int c = 0; c < d; ++c) xn = fma((double)q[c], (double)q[c], xn);
        s_xn[tid] = xn;
    }
    __syncthreads();
    // cut list g back to its k smallest by (value, index), in order (uniform over the workgroup)
    auto cut = [&](int g) {
        double *v = s_v + (size_t)g * CAP;
        int *ix = s_i + (size_t)g * CAP;
        const int m = s_n[g];
        for (int s = tid; s < m; s += XS_T) {
            const double a = v[s];
            const int id = ix[s];
            int rk = 0;
#pragma unroll 8
            for (int j = 0; j < m; ++j) rk += pair_before(v[j], ix[j], a, id) ? 1 : 0;   // (eight broadcast reads in flight)
            if (rk < k) { t_v[rk] = a; t_i[rk] = id; }
        }
        __syncthreads();
        const int keep = m < k ? m : k;
        for (int s = tid; s < keep; s += XS_T) { v[s] = t_v[s]; ix[s] = t_i[s]; }
        if (tid == 0) { s_n[g] = keep; if (m >= k) s_thr[g] = t_v[k - 1]; }
        __syncthreads();
    };
    const T *qa[QG], *ql[QG];
    double xa[QG], lbv[QG];
    int lbi[QG];
#pragma unroll
    for (int g = 0; g < QG; ++g) {
        const int gg = g < ng ? g : ng - 1;   // (a short last group repeats its last row)
        qa[g] = qry + (q0 + gg) * d; ql[g] = s_qc + g * DC; xa[g] = s_xn[gg];
        lbv[g] = have_lb ? lb_v[q0 + gg] : -(double)INFINITY;
        lbi[g] = have_lb ? lb_i[q0 + gg] : -1;
    }
    // the next step's rows and query columns travel to registers while the current step is evaluated
    // (16-byte pieces when every row segment is aligned -- d a multiple of 4 floats / 2 doubles: a quarter of the load and store
    //  instructions; single elements otherwise)
    constexpr int E = 16 / (int)sizeof(T), NP = DC / E;   // elements per piece, pieces per row and step (8)
    struct alignas(16) P16 { T v[E]; };
    constexpr bool vec = VEC;   // (host: d a multiple of E and the reference 16-byte aligned)
    T pre[DC], preq = (T)0;
    auto fetch = [&](int64_t base, int c0) {
        const int nv = d - c0 < DC ? d - c0 : DC;
        const T *rb = ref + base * d + c0;
        if constexpr (vec) {
#pragma unroll
            for (int it = 0; it < NP; ++it) {
                const int idx = tid + it * XS_T;
                const int r = idx / NP, pc = idx - r * NP;
                P16 x;
#pragma unroll
                for (int e = 0; e < E; ++e) x.v[e] = (T)0;
                if (base + r < n_ref && (pc + 1) * E <= nv) x = *reinterpret_cast<const P16 *>(rb + (r * d + pc * E));   // (32-bit offsets from a uniform base)
#pragma unroll
                for (int e = 0; e < E; ++e) pre[it * E + e] = x.v[e];
            }
        } else {
#pragma unroll
            for (int it = 0; it < DC; ++it) {
                const int idx = tid + it * XS_T;
                const int r = idx / DC, c = idx - r * DC;
                pre[it] = (base + r < n_ref && c < nv) ? rb[r * d + c] : (T)0;
            }
        }
        if (tid < QG * DC) {
            const int g = tid / DC, c = tid - g * DC;
            preq = c < nv ? qa[g][c0 + c] : (T)0;
        }
    };
    fetch(0, 0);
    for (int64_t base = 0; base < n_ref; base += XS_T) {
        const int64_t j = base + tid;
        PairAcc<T, T, QG> acc;
        acc.init();
        for (int c0 = 0; c0 < d; c0 += DC) {
            const int nv = d - c0 < DC ? d - c0 : DC;
            __syncthreads();   // (the previous step's readers, or a cut, are done with the buffer)
            if constexpr (vec) {
#pragma unroll
                for (int it = 0; it < NP; ++it) {
                    const int idx = tid + it * XS_T;
                    const int r = idx / NP, pc = idx - r * NP;
                    P16 x;
#pragma unroll
                    for (int e = 0; e < E; ++e) x.v[e] = pre[it * E + e];
                    *reinterpret_cast<P16 *>(s_y + r * DCP + pc * E) = x;
                }
            } else {
#pragma unroll
                for (int it = 0; it < DC; ++it) {
                    const int idx = tid + it * XS_T;
                    const int r = idx / DC, c = idx - r * DC;
                    s_y[r * DCP + c] = pre[it];
                }
            }
            if (tid < QG * DC) s_qc[tid] = preq;
            __syncthreads();
            if (c0 + DC < d) fetch(base, c0 + DC);
            else if (base + XS_T < n_ref) fetch(base + XS_T, 0);
            if (j < n_ref) {
                const T *yrow = s_y + tid * DCP;
                if (nv == DC) acc.template run<8>(ql, 0, DC, [&](int i) { return yrow[i]; }, metric, form);   // (eight columns in flight: a full unroll spills)
                else acc.run(ql, 0, nv, [&](int i) { return yrow[i]; }, metric, form);
            }
        }
        if (j < n_ref) {
            double va[QG];
            acc.finish(xa, metric, form, va, d);
#pragma unroll
            for (int g = 0; g < QG; ++g) {
                const double v = va[g] == va[g] ? va[g] : (double)INFINITY;   // (a NaN value -- dice / sokalsneath of two all-false rows -- ranks last)
                if (g < ng && v <= s_thr[g] && (!have_lb || pair_before(lbv[g], (int64_t)lbi[g], v, (int64_t)j))) {
                    const int pos = atomicAdd(&s_n[g], 1);   // (< CAP: every list entered this pass with 256 free slots)
                    s_v[(size_t)g * CAP + pos] = v;
                    s_i[(size_t)g * CAP + pos] = (int)j;
                }
            }
        }
        __syncthreads();
        for (int g = 0; g < ng; ++g)
            if (s_n[g] > CAP - XS_T) cut(g);
    }
    for (int g = 0; g < ng; ++g) {
        cut(g);
        const int m = s_n[g];
        const int64_t io = q0 + g;
        for (int rk = tid; rk < m; rk += XS_T) {
            const double v = s_v[(size_t)g * CAP + rk];
            double o = v;
            if (metric == CM_METRIC_EUCLIDEAN && !(flags & CM_KNN_SQUARED)) o = sqrt(v);
            if ((metric == CM_METRIC_DICE || metric == CM_METRIC_SOKALSNEATH || metric == CM_METRIC_NAN_EUCLIDEAN) && !(v < (double)INFINITY)) o = __longlong_as_double(0x7ff8000000000000LL);
            if (metric == CM_METRIC_COSINE && !(flags & CM_KNN_F64_OUT)) reinterpret_cast<T *>(out_dist)[io * k_total + k_off + rk] = (T)o;
            else reinterpret_cast<double *>(out_dist)[io * k_total + k_off + rk] = o;
            out_idx[io * k_total + k_off + rk] = (int64_t)s_i[(size_t)g * CAP + rk] + idx_offset;
        }
        if (lb_v && tid == 0 && m > 0) { lb_v[io] = s_v[(size_t)g * CAP + m - 1]; lb_i[io] = s_i[(size_t)g * CAP + m - 1]; }
    }
}

// ------------------------------------------------------------------ shard merge ---------------------
// parts [nparts, nq, k] -> the k smallest by (distance, global index).  One wave per query.
// Ranks are UNIQUE whatever the lists hold, so every output slot is written exactly once: entries are ordered by (value, index,
// position in the concatenated lists) -- two shards never report the same reference row, but a caller that hands in
// overlapping lists gets both copies, not an unwritten slot -- and a NaN value (dice / sokalsneath between two all-false rows:
// the exact scan writes the NaN sklearn returns into the per-shard lists) ranks as +inf, after every number, like the
// single-device scan ranks it; it is written out as NaN again (round-4 advisor finding: NaN compared false against everything,
// took rank 0 and raced with the true nearest neighbour).
__global__ __launch_bounds__(256) void knn_merge(const double *__restrict__ pd, const int64_t *__restrict__ pi,
                                                  int nparts, int64_t part_stride, int64_t nq, int k, int sqrt_out,
                                                  double *__restrict__ out_dist, int64_t *__restrict__ out_idx) {
    __shared__ double s_v[4][MG_NCMAX];
    __shared__ int64_t s_i[4][MG_NCMAX];
    constexpr int64_t NAN_TAG = (int64_t)1 << 62;   // (global row numbers stay far below 2^62)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 4 + w;
    if (i >= nq) return;
    const int nc = nparts * k;
    for (int s = lane; s < nc; s += 64) {
        int p = s / k, t = s - p * k;
        const double v = pd[(int64_t)p * part_stride + i * k + t];
        const int64_t id = pi[(int64_t)p * part_stride + i * k + t];
        const bool is_nan = v != v && id >= 0;
        s_v[w][s] = v != v ? INFINITY : v;
        s_i[w][s] = is_nan ? (id | NAN_TAG) : id;
    }
    __builtin_amdgcn_wave_barrier();
    for (int s = lane; s < nc; s += 64) {
        double v = s_v[w][s];
        int64_t id = s_i[w][s];
        int rk = 0;
        for (int j = 0; j < nc; ++j) {
            const double vj = s_v[w][j];
            const int64_t ij = s_i[w][j];
            rk += (vj < v || (vj == v && (ij < id || (ij == id && j < s)))) ? 1 : 0;
        }
        if (rk < k) {
            const bool was_nan = (id & NAN_TAG) != 0;
            out_dist[i * k + rk] = was_nan ? __longlong_as_double(0x7ff8000000000000LL) : (sqrt_out ? sqrt(v) : v);
            out_idx[i * k + rk] = id & ~NAN_TAG;
        }
    }
}

// bound[] = +inf, seed[] = +inf (staged search: monotone k-th-value bounds and next-stage threshold seeds per query)
__global__ void init_bounds(double *bound, float *seedb, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { bound[i] = INFINITY; seedb[i] = INFINITY; }
}
// thr0[pos] = seed of the query the packed position holds (positions beyond the subset: +inf).
// qnorm != nullptr (round 5: seeds left by the bucket-centred level): the seed carries twice the error of the level that left
// it, which bounds the next level's only where that level is the less precise one -- here the seed gets twice the TARGET
// level's own bound (knn_rerank's, with the global |y'| maximum) on top, so the k rows behind the bound stay below it and
// the certificate has its room whatever level the seed came from.
__global__ void seed_gather(const float *__restrict__ seedb, const int *__restrict__ qmap, int64_t nsub, int64_t n_pad,
                            float *__restrict__ thr0, const float *__restrict__ qnorm = nullptr, const float *__restrict__ qlo = nullptr,
                            const unsigned *__restrict__ gmax = nullptr, const float *__restrict__ scale3 = nullptr, float c_eps = 0.f,
                            int level = 0) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pad) return;
    float v = INFINITY;
    if (i < nsub) { const int q = qmap[i]; if (q >= 0) v = seedb[q]; }
    if (qnorm && v < INFINITY) {
        const double ymax = (double)__uint_as_float(gmax[1]), rho_y = (double)__uint_as_float(gmax[5]);
        const double qn = (double)qnorm[i];
        const double eps = (double)c_eps * (ymax * ymax + 2.0 * qn * ymax + 1e-30) +
                           2.0 * 1.001 * ymax * ((level == 1 ? (double)qlo[i] : 0.0) + (level <= 2 ? qn * rho_y : 0.0));
        const double w = (double)v + 2.0 * eps / (double)scale3[2];
        v = (float)w;
        if ((double)v < w) v = nextafterf(v, INFINITY);
    }
    thr0[i] = v;
}
// the same into slot 0 of the bucket-pruned stage's per-query threshold slots (stride S): pass A starts from it.  direct != 0:
// the direct-difference filter ranks by the distance value itself (cosine rows are normalised: 2 x the cosine distance), so
// its seed is the query's bound plus that filter's relative error.
__global__ void seed_scatter_slot0(const float *__restrict__ seedb, const double *__restrict__ bound, const int *__restrict__ qmap,
                                   int64_t nsub, int S, int direct, int is_cosine, double crel, float *__restrict__ cthr,
                                   const float *__restrict__ qnorm = nullptr, const float *__restrict__ qlo = nullptr,
                                   const unsigned *__restrict__ gmax = nullptr, const float *__restrict__ scale3 = nullptr,
                                   float c_eps = 0.f, int level = 0) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nsub) return;
    const int q = qmap[i];
    if (q < 0) return;
    float v = INFINITY;
    if (!direct) {
        v = seedb[q];
        if (qnorm && v < INFINITY) {   // (seeds of the bucket-centred level: + twice this level's own bound, as in seed_gather)
            const double ymax = (double)__uint_as_float(gmax[1]), rho_y = (double)__uint_as_float(gmax[5]);
            const double qn = (double)qnorm[i];
            const double eps = (double)c_eps * (ymax * ymax + 2.0 * qn * ymax + 1e-30) +
                               2.0 * 1.001 * ymax * ((level == 1 ? (double)qlo[i] : 0.0) + (level <= 2 ? qn * rho_y : 0.0));
            const double w = (double)v + 2.0 * eps / (double)scale3[2];
            v = (float)w;
            if ((double)v < w) v = nextafterf(v, INFINITY);
        }
    } else {
        const double b = bound[q];
        if (b < INFINITY) {
            const double x = (is_cosine ? 2.0 * b : b) * (1.0 + crel) + 1e-30;
            v = (float)x;
            if ((double)v < x) v = nextafterf(v, INFINITY);
        }
    }
    cthr[i * S] = v;
}
__global__ void zero_words(unsigned *p, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0u;
}

struct KnnPlan {
    int KP, NW, Kc, C, S, n_tiles, tps;
    int64_t nq_pad, nref_pad;
    size_t lds;
};

// 0: planned; 1: k too large for the on-chip candidate buffers; 2: k too large for the re-rank's lists
int make_plan_quiet(int64_t n_ref, int64_t nq, int d, int k, KnnPlan *p) {
    p->KP = pick_kp(d);
    // d > 128 (every width the split-fp16 families do not serve): the K-chunked kernel, prepared rows padded to a multiple of its
    // chunk width -- beyond KP = 260 nothing else fits on chip, and from 129 on it is several times faster than the one-piece
    // kernel, whose 128 x KP tile leaves LDS for a single wave of queries (d = 256: 881 -> see DESIGN.md 7b)
    const bool wide = p->KP < 0 || d > 128;
    if (wide) p->KP = (d + 1 + cmk::KCW - 1) / cmk::KCW * cmk::KCW;
    int slack = k / 4 > 8 ? k / 4 : 8;
    p->Kc = k + slack;
    p->C = p->Kc + (wide ? 40 : 64);   // (wide rows: room for four waves next to the 34-KB chunk)
    p->NW = 0;
    for (int nw : {4, 2, 1}) {
        if ((wide ? cmk::phase1_wide_lds_bytes(nw, p->C) : cmk::phase1_lds_bytes(p->KP, nw, p->C)) <= 160 * 1024) { p->NW = nw; break; }
    }
    if (!p->NW) return 1;
    p->lds = wide ? cmk::phase1_wide_lds_bytes(p->NW, p->C) : cmk::phase1_lds_bytes(p->KP, p->NW, p->C);
    int QB = p->NW * 32;
    p->nq_pad = (nq + QB - 1) / QB * QB;
    p->n_tiles = (int)((n_ref + cmk::BR - 1) / cmk::BR);
    p->nref_pad = (int64_t)p->n_tiles * cmk::BR;
    int64_t nqb = p->nq_pad / QB;
    int S = 1;
    if (nqb < 256) {
        S = (int)((512 + nqb - 1) / nqb);
        if (S > 8) S = 8;
        if (S > p->n_tiles) S = p->n_tiles;
        while (S > 1 && S * p->Kc > RR_NCMAX) --S;
        if (S < 1) S = 1;
    }
    p->tps = (p->n_tiles + S - 1) / S;
    p->S = (p->n_tiles + p->tps - 1) / p->tps;
    if (p->S * p->Kc > RR_NCMAX) return 2;
    return 0;
}
cm_status make_plan(cm_handle_t h, int64_t n_ref, int64_t nq, int d, int k, KnnPlan *p) {
    const int rc = make_plan_quiet(n_ref, nq, d, k, p);
    if (rc == 1) CM_FAIL(h, CM_E_UNSUPPORTED, "k-NN: n_neighbors=%d too large for the on-chip candidate buffers", k);
    if (rc == 2) CM_FAIL(h, CM_E_UNSUPPORTED, "k-NN: n_neighbors=%d too large (max %d)", k, RR_NCMAX - 64);
    return CM_OK;
}

unsigned long long *g_dbg_ctr = nullptr;  // CELLMAP_DBG=4 diagnostics only

struct KnnPlan2 {
    int KSTEPS, Kc, S, n_tiles, tps, nqb, grid;
    int64_t nq_pad;
    size_t lds, tileb, rowb;
};

int knn_slack(int k) { return k / 4 > 8 ? k / 4 : 8; }
// row stride of the direct-difference (VALU) kernel: d itself up to 15, else padded to the instantiated widths
int lowd_dpad(int d) { return d <= 15 ? d : (d + 3) / 4 * 4; }
// candidate slack of the split-fp16 kernels by number of products evaluated (coarser filter -> wider safety margin)
int knn_slack_level(int k, int nprod) {
    { const int v = cm_get_knobs().slack; if (v >= 0 && v <= 256) return v; }  // tuning only
    if (nprod >= 3) return knn_slack(k);
    if (nprod == 2) return k / 3 > 12 ? k / 3 : 12;
    return k / 2 > 16 ? k / 2 : 16;
}

// phase-1 kernel choice: 2 = low-d VALU direct form (d <= 15), 1 = split-fp16 MFMA K <= 64 (d <= 61),
// 3 = split-fp16 MFMA K = 128 (d <= 125), 0 = f32-input MFMA
int phase1_path(int d, int k) {
    const int e = cm_get_knobs().phase1;   // CELLMAP_PHASE1: 1 f32, 2 f16, 3 direct
    if (e == 1) return 0;
    if (e == 3 && d <= 64) return 2;   // diagnostics: the VALU direct-difference kernel for any d <= 64
    if (k + knn_slack_level(k, 1) + 2 * cmk2::CL > 512) return 0;  // the in-register merge handles 512 entries per query
    if (d <= 15 && e != 2) return 2;
    if (d + cmk2::NORM_SLOTS <= 64) return 1;             // needs three spare feature slots for the norm digits
    return d <= 128 ? 3 : 0;  // the K = 128 kernel carries |y|^2 in a side block, all 128 slots hold features
}
bool use_f16_path(int d, int k) { return phase1_path(d, k) != 0; }

cm_status make_plan2(cm_handle_t h, int64_t n_ref, int64_t nq, int d, int k, int path, int nprod, KnnPlan2 *p) {
    p->KSTEPS = path == 3 ? cmk2::KSTEPS_W : (d + cmk2::NORM_SLOTS <= 32 ? 2 : 4);
    p->Kc = k + ((path == 1 || path == 3) ? knn_slack_level(k, nprod) : knn_slack(k));
    const int dl = lowd_dpad(d);   // path 2: row stride of the direct-difference kernel
    const int tile_rows = path == 2 ? cmk2::lowd_rows(dl) : cmk2::BR;
    p->tileb = path == 2 ? (size_t)cmk2::lowd_tile_bytes(dl)
                         : (path == 3 ? (size_t)cmk2::wide_tile_bytes() : (size_t)cmk2::tile_bytes(p->KSTEPS));
    p->rowb = path == 2 ? (size_t)dl * 4 : (size_t)cmk2::row_bytes(p->KSTEPS);
    p->lds = (path == 1 ? cmk2::ring_depth(false) : 2) * p->tileb + 64;   // knn_cand_f16: see ring_depth (set per level)
    if (p->lds > 160 * 1024) CM_FAIL(h, CM_E_UNSUPPORTED, "k-NN: n_neighbors=%d too large for the on-chip merge scratch", k);
    // queries per workgroup pass of the chosen kernel (the direct kernel takes one query per lane beyond 64 floats per row)
    const int qb = path == 3 ? cmk2::QBW : (path == 2 ? cmk2::lowd_qb(dl) : cmk2::QB);
    p->nq_pad = (nq + qb - 1) / qb * qb;
    p->nqb = (int)(p->nq_pad / qb);
    p->n_tiles = (int)((n_ref + tile_rows - 1) / tile_rows);
    if (h->n_cu <= 0) {   // (queried once per handle: the call costs a fraction of a millisecond, a stage of the search runs it)
        int v = 0;
        h->n_cu = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, h->device) == hipSuccess && v > 0) ? v : 256;
    }
    const int ncu = h->n_cu;
    const int resident = path == 3 ? ncu : ncu * (8 / cmk2::NW);  // 8 waves per CU: one 8-wave or two 4-wave workgroups
    int S = 1;
    if (p->nqb < resident) {
        S = (resident + p->nqb - 1) / p->nqb;
        if (S > 16) S = 16;
        if (S > p->n_tiles) S = p->n_tiles;
        while (S > 1 && S * p->Kc > RR_NCMAX) --S;
        if (S < 1) S = 1;
    }
    p->tps = (p->n_tiles + S - 1) / S;
    p->S = (p->n_tiles + p->tps - 1) / p->tps;
    if (p->S * p->Kc > RR_NCMAX) CM_FAIL(h, CM_E_UNSUPPORTED, "k-NN: n_neighbors=%d too large (max %d)", k, RR_NCMAX - 64);
    int64_t n_work = (int64_t)p->nqb * p->S;
    p->grid = (int)(n_work < resident ? n_work : resident);
    return CM_OK;
}

// ------------------------------------------------------------------ locality sort ---------------------
// Rows are bucketed by their nearest of C pivots (reference rows sampled at a fixed stride) and both the reference tile
// image and the order in which queries are packed into waves follow that bucketing.  A wave then holds 64 queries of one
// neighbourhood and a tile holds reference rows of one neighbourhood: candidates arrive in bursts that one append event
// handles together, instead of one event per stray value.  Pure scheduling: which rows share a tile or a wave has no
// influence on the result.
constexpr int LS_CMAX = 256;   // pivots
constexpr int LS_ROWS = 4096;  // rows per workgroup of the bucketing kernels
constexpr int LS_DSUB = 16;    // leading coordinates used for the bucketing (one 64-byte line per row; for PCA embeddings
                               // these carry most of the cluster structure -- any partition is valid, this one is cheap)
__global__ void strided_rows(int *__restrict__ out, int n, int64_t stride);
// Bucket numbering: a greedy nearest-neighbour chain through the pivots, so that buckets with consecutive numbers are
// neighbours in space.  Query blocks and reference tiles that straddle two buckets then still cover one neighbourhood
// (bucket pruning bounds every block and bucket by a ball).  One wave; order[p] = pivot at chain position p.
template <typename T>
__global__ __launch_bounds__(64) void ls_chain(const T *__restrict__ pivsrc, int64_t piv_stride, int d, int C,
                                               int *__restrict__ order) {
    __shared__ float piv[LS_CMAX][LS_DSUB];
    const int lane = threadIdx.x;
    const int ds = d < LS_DSUB ? d : LS_DSUB;
    for (int t = lane; t < C * LS_DSUB; t += 64) {
        const int c = t / LS_DSUB, e = t % LS_DSUB;
        piv[c][e] = e < ds ? (float)pivsrc[(int64_t)c * piv_stride * d + e] : 0.f;
    }
    __syncthreads();
    bool used[LS_CMAX / 64];   // lane owns pivots lane + 64*i
#pragma unroll
    for (int i = 0; i < LS_CMAX / 64; ++i) used[i] = lane + 64 * i >= C;
    int cur = 0;
    if (lane == 0) { order[0] = 0; used[0] = true; }
    for (int p = 1; p < C; ++p) {
        float best = INFINITY;
        int bc = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < LS_CMAX / 64; ++i) {
            const int c = lane + 64 * i;
            if (used[i]) continue;
            float sacc = 0.f;
#pragma unroll
            for (int e = 0; e < LS_DSUB; ++e) { const float t = piv[c][e] - piv[cur][e]; sacc = fmaf(t, t, sacc); }
            if (sacc < best || (sacc == best && c < bc)) { best = sacc; bc = c; }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o);
            const int oc = __shfl_xor(bc, o);
            if (ob < best || (ob == best && oc < bc)) { best = ob; bc = oc; }
        }
        cur = bc;
        if (lane == (cur & 63)) used[cur >> 6] = true;
        if (lane == 0) order[p] = cur;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void ls_assign(const T *__restrict__ X, int64_t n, int d, const T *__restrict__ pivsrc,
                                                 int64_t piv_stride, int C, int *__restrict__ label, int *__restrict__ counts,
                                                 const int *__restrict__ order) {
    __shared__ float ls_piv[LS_CMAX][LS_DSUB];
    const int ds = d < LS_DSUB ? d : LS_DSUB;
    for (int t = threadIdx.x; t < C * LS_DSUB; t += blockDim.x) {
        const int c = t / LS_DSUB, e = t % LS_DSUB;
        ls_piv[c][e] = e < ds ? (float)pivsrc[(int64_t)order[c] * piv_stride * d + e] : 0.f;
    }
    __shared__ int ls_hist[LS_CMAX];
    if (threadIdx.x < LS_CMAX) ls_hist[threadIdx.x] = 0;
    __syncthreads();
    // LS_ROWS rows per workgroup: the per-bucket counts go through an LDS histogram (64 global atomics per workgroup)
    for (int rr = 0; rr < LS_ROWS / 256; ++rr) {
        const int64_t i = (int64_t)blockIdx.x * LS_ROWS + rr * 256 + threadIdx.x;
        if (i >= n) break;
        float x[LS_DSUB];
#pragma unroll
        for (int e = 0; e < LS_DSUB; ++e) x[e] = e < ds ? (float)X[i * d + e] : 0.f;
        float best = INFINITY;
        int bc = 0;
        for (int c = 0; c < C; ++c) {
            float sacc = 0.f;
#pragma unroll
            for (int e = 0; e < LS_DSUB; ++e) { const float t = x[e] - ls_piv[c][e]; sacc = fmaf(t, t, sacc); }
            if (sacc < best) { best = sacc; bc = c; }
        }
        label[i] = bc;
        atomicAdd(&ls_hist[bc], 1);
    }
    __syncthreads();
    if (threadIdx.x < C && ls_hist[threadIdx.x]) atomicAdd(&counts[threadIdx.x], ls_hist[threadIdx.x]);
}
__global__ void ls_offsets(const int *__restrict__ counts, int C, int *__restrict__ cursor) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int run = 0;
        for (int c = 0; c < C; ++c) { cursor[c] = run; run += counts[c]; }
    }
}
__global__ void ls_starts(const int *__restrict__ counts, int C, int *__restrict__ start /* C + 1 */) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int run = 0;
        for (int c = 0; c < C; ++c) { start[c] = run; run += counts[c]; }
        start[C] = run;
    }
}
__global__ __launch_bounds__(256) void ls_scatter(const int *__restrict__ label, int64_t n, int *__restrict__ cursor,
                                                  int *__restrict__ perm) {
    __shared__ int hist[LS_CMAX], base[LS_CMAX];
    if (threadIdx.x < LS_CMAX) hist[threadIdx.x] = 0;
    __syncthreads();
    const int64_t r0 = (int64_t)blockIdx.x * LS_ROWS;
    for (int rr = 0; rr < LS_ROWS / 256; ++rr) {
        const int64_t i = r0 + rr * 256 + threadIdx.x;
        if (i < n) atomicAdd(&hist[label[i]], 1);
    }
    __syncthreads();
    if (threadIdx.x < LS_CMAX) { base[threadIdx.x] = hist[threadIdx.x] ? atomicAdd(&cursor[threadIdx.x], hist[threadIdx.x]) : 0; hist[threadIdx.x] = 0; }
    __syncthreads();
    for (int rr = 0; rr < LS_ROWS / 256; ++rr) {
        const int64_t i = r0 + rr * 256 + threadIdx.x;
        if (i < n) { const int c = label[i]; perm[base[c] + atomicAdd(&hist[c], 1)] = (int)i; }
    }
}

// ------------------------------------------------------------------ structure pre-test (round 4) -----
// The locality sort, the bucket-pruning gate and the pilot only pay on embeddings with cluster structure.  Whether there is
// any is decided INSIDE the call, before the sort, on a strided sample of both sides (the round-3 library learned it from
// the pruning gate of the FIRST call and remembered it on the handle, so a first call was 5 % slower than the twentieth):
// SG_N rows per side are assigned to the sort's own pivots (same 16 leading coordinates), every bucket of either side gets
// a bounding ball of its sample (float64 centre over all d coordinates, max distance), and the (query bucket, reference
// bucket) pairs whose balls overlap are counted, weighted by the buckets' sizes -- the bucket-level version of the
// pruning gate's (query block, reference tile) count.  Sample radii can only be smaller than the buckets' own, so "the
// balls overlap for 15/16 of the pairs" here implies it for the full buckets: such data takes the plain route (no
// sort, no pruning attempt, no pilot: the hi*hi level starts at once, its certificate and the ladder behind it as
// before).  Pure scheduling: the result never depends on the verdict.
constexpr int SG_N = 16384;   // sample rows per side
constexpr int SG_STR = 128;   // doubles per centre (d <= 128 on the staged path)
template <typename T>
__global__ __launch_bounds__(256) void sg_assign(const T *__restrict__ X, int64_t stride, int ns, int d, const T *__restrict__ pivsrc,
                                                 int64_t piv_stride, int C, int *__restrict__ label, int *__restrict__ counts) {
    __shared__ float s_piv[LS_CMAX][LS_DSUB];
    __shared__ int s_hist[LS_CMAX];
    const int ds = d < LS_DSUB ? d : LS_DSUB;
    for (int t = threadIdx.x; t < C * LS_DSUB; t += blockDim.x) {
        const int c = t / LS_DSUB, e = t % LS_DSUB;
        s_piv[c][e] = e < ds ? (float)pivsrc[(int64_t)c * piv_stride * d + e] : 0.f;
    }
    if (threadIdx.x < LS_CMAX) s_hist[threadIdx.x] = 0;
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ns) {
        float x[LS_DSUB];
#pragma unroll
        for (int e = 0; e < LS_DSUB; ++e) x[e] = e < ds ? (float)X[(int64_t)i * stride * d + e] : 0.f;
        float best = INFINITY;
        int bc = 0;
        for (int c = 0; c < C; ++c) {
            float sacc = 0.f;
#pragma unroll
            for (int e = 0; e < LS_DSUB; ++e) { const float t = x[e] - s_piv[c][e]; sacc = fmaf(t, t, sacc); }
            if (sacc < best) { best = sacc; bc = c; }
        }
        label[i] = bc;
        atomicAdd(&s_hist[bc], 1);
    }
    __syncthreads();
    if (threadIdx.x < C && s_hist[threadIdx.x]) atomicAdd(&counts[threadIdx.x], s_hist[threadIdx.x]);
}
// one workgroup per bucket: centre and radius of the sample rows carrying its label (cosine: rows normalised first)
template <typename T>
__global__ __launch_bounds__(256) void sg_ball(const T *__restrict__ X, int64_t stride, int ns, int d, int metric,
                                               const int *__restrict__ label, double *__restrict__ cen, double *__restrict__ rad) {
    __shared__ double s_acc[4][SG_STR];
    __shared__ double s_cen[SG_STR];
    __shared__ double s_r[4];
    __shared__ int s_n[4];
    const int g = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int pass = 0; pass < 2; ++pass) {
        double a0 = 0.0, a1 = 0.0, rmax = 0.0;
        int n = 0;
        for (int c0 = w * 64; c0 < ns; c0 += 256) {
            const int i = c0 + lane;
            unsigned long long m = __ballot(i < ns && label[i] == g);
            while (m) {
                const int b = __ffsll((long long)m) - 1;
                m &= m - 1;
                const T *x = X + (int64_t)(c0 + b) * stride * d;
                double v0 = lane < d ? (double)x[lane] : 0.0, v1 = lane + 64 < d ? (double)x[lane + 64] : 0.0;
                if (metric == CM_METRIC_COSINE) {
                    double s2 = v0 * v0 + v1 * v1;
                    for (int o = 32; o > 0; o >>= 1) s2 += __shfl_xor(s2, o);
                    const double nrm = sqrt(s2), inv = nrm > 0.0 ? 1.0 / nrm : 1.0;
                    v0 *= inv; v1 *= inv;
                }
                if (pass == 0) { a0 += v0; a1 += v1; ++n; }
                else {
                    const double t0 = lane < d ? v0 - s_cen[lane] : 0.0, t1 = lane + 64 < d ? v1 - s_cen[lane + 64] : 0.0;
                    double s2 = t0 * t0 + t1 * t1;
                    for (int o = 32; o > 0; o >>= 1) s2 += __shfl_xor(s2, o);
                    rmax = fmax(rmax, s2);
                }
            }
        }
        if (pass == 0) {
            s_acc[w][lane] = a0; s_acc[w][lane + 64] = a1;
            if (lane == 0) s_n[w] = n;
            __syncthreads();
            if (threadIdx.x < SG_STR) {
                const int nn = s_n[0] + s_n[1] + s_n[2] + s_n[3];
                const double c = nn ? (s_acc[0][threadIdx.x] + s_acc[1][threadIdx.x] + s_acc[2][threadIdx.x] + s_acc[3][threadIdx.x]) / (double)nn : 0.0;
                s_cen[threadIdx.x] = c;
                cen[(size_t)g * SG_STR + threadIdx.x] = c;
            }
            __syncthreads();
        } else {
            if (lane == 0) s_r[w] = rmax;
            __syncthreads();
            if (threadIdx.x == 0) rad[g] = sqrt(fmax(fmax(s_r[0], s_r[1]), fmax(s_r[2], s_r[3])));
        }
    }
}
// out[0] = sum of cq[jq] * cr[jr] over the pairs whose balls overlap, out[1] = over all pairs
__global__ __launch_bounds__(256) void sg_overlap(const double *__restrict__ qcen, const double *__restrict__ qrad, const int *__restrict__ qcnt,
                                                  const double *__restrict__ rcen, const double *__restrict__ rrad, const int *__restrict__ rcnt,
                                                  int C, int d, unsigned long long *__restrict__ out) {
    __shared__ unsigned long long s_o[256], s_t[256];
    const int jq = blockIdx.x, jr = threadIdx.x;
    unsigned long long ov = 0, tot = 0;
    if (jr < C && qcnt[jq] > 0 && rcnt[jr] > 0) {
        double s2 = 0.0;
        for (int c = 0; c < d; ++c) { const double t = qcen[(size_t)jq * SG_STR + c] - rcen[(size_t)jr * SG_STR + c]; s2 = fma(t, t, s2); }
        tot = (unsigned long long)qcnt[jq] * (unsigned long long)rcnt[jr];
        if (!(sqrt(s2) > qrad[jq] + rrad[jr])) ov = tot;
    }
    s_o[jr] = ov; s_t[jr] = tot;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (jr < o) { s_o[jr] += s_o[jr + o]; s_t[jr] += s_t[jr + o]; }
        __syncthreads();
    }
    if (jr == 0) { atomicAdd(&out[0], s_o[0]); atomicAdd(&out[1], s_t[0]); }
}

// ------------------------------------------------------------------ bucket-centred queries -----------
// (knn_cand_f16<., 1, true>)  Buckets are those of the locality sort.  cen_centroid: mean of the bucket's prepared
// (centred, cosine: normalised) query rows in float64.  cen_pack: packed query order in which no 512-query block
// straddles two buckets (-1 = padding).  cen_side: for every bucket and every row of the (sorted, padded) reference
// image the float32 value -2 sc^2 c_bucket . y'_row, accumulated in float64.
// (round 5: a bucket's rows are summed in CEN_SLICES slices by as many workgroups -- one workgroup per bucket left a 40-cluster set
//  with 40 busy workgroups, 4.5 ms for 1.5 M rows -- and combined in slice order by cen_centroid_final: deterministic)
constexpr int CEN_SLICES = 32;
template <typename T>
__global__ __launch_bounds__(256) void cen_centroid(const T *__restrict__ Q, int d, int metric, const double *__restrict__ mean,
                                                    const int *__restrict__ qperm, const int *__restrict__ bstart,
                                                    double *__restrict__ part /* [C][CEN_SLICES][64] */) {
    __shared__ double s_acc[4][64];
    const int j = blockIdx.x, sl = blockIdx.y, e = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int lo = bstart[j], hi = bstart[j + 1];
    const int per = (hi - lo + CEN_SLICES - 1) / CEN_SLICES;
    const int r0 = lo + sl * per, r1 = r0 + per < hi ? r0 + per : hi;
    double acc = 0.0;
    if (e < d)
        for (int r = r0 + rl; r < r1; r += 4) {
            const T *x = Q + (int64_t)qperm[r] * d;
            double ns = 1.0;
            if (metric == CM_METRIC_COSINE) {
                double s = 0.0;
                for (int c = 0; c < d; ++c) { double v = (double)x[c]; s = fma(v, v, s); }
                const double nrm = sqrt(s);
                ns = nrm > 0.0 ? 1.0 / nrm : 1.0;
            }
            acc += (double)x[e] * ns - mean[e];
        }
    s_acc[rl][e] = acc;
    __syncthreads();
    if (rl == 0) part[((size_t)j * CEN_SLICES + sl) * 64 + e] = e < d ? (s_acc[0][e] + s_acc[1][e] + s_acc[2][e] + s_acc[3][e]) : 0.0;
}
__global__ __launch_bounds__(64) void cen_centroid_final(const double *__restrict__ part, const int *__restrict__ bstart, int d,
                                                        double *__restrict__ cen) {
    const int j = blockIdx.x, e = threadIdx.x;
    if (e >= d) return;
    double s = 0.0;
    for (int sl = 0; sl < CEN_SLICES; ++sl) s += part[((size_t)j * CEN_SLICES + sl) * 64 + e];
    const int n = bstart[j + 1] - bstart[j];
    cen[(size_t)j * d + e] = n > 0 ? s / (double)n : 0.0;
}

__global__ void cen_pack(const int *__restrict__ qperm, const int *__restrict__ bstart, const int *__restrict__ pstart, int C,
                         int64_t n_packed, int *__restrict__ qpack, int *__restrict__ blk_bucket) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_packed) return;
    int j = 0;
    while (j + 1 < C && (int64_t)pstart[j + 1] <= i) ++j;   // C <= 64: linear search
    const int64_t off = i - pstart[j];
    qpack[i] = off < (int64_t)(bstart[j + 1] - bstart[j]) ? qperm[bstart[j] + off] : -1;
    if ((i % cmk2::QB) == 0) blk_bucket[i / cmk2::QB] = j;
}

template <typename T>
__global__ __launch_bounds__(256) void cen_side(const T *__restrict__ R, int64_t n_ref, int d, int metric,
                                                const double *__restrict__ mean, const int *__restrict__ rperm,
                                                const double *__restrict__ cen, int C, const float *__restrict__ scale3,
                                                int64_t stride, float *__restrict__ side) {
    extern __shared__ double s_cen[];   // [C][d]
    for (int t = threadIdx.x; t < C * d; t += blockDim.x) s_cen[t] = cen[t];
    __syncthreads();
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= stride) return;
    double y[64];
    const bool live = r < n_ref;
    if (live) {
        const T *x = R + (int64_t)(rperm ? rperm[r] : r) * d;
        double ns = 1.0;
        if (metric == CM_METRIC_COSINE) {
            double s = 0.0;
            for (int c = 0; c < d; ++c) { double v = (double)x[c]; s = fma(v, v, s); }
            const double nrm = sqrt(s);
            ns = nrm > 0.0 ? 1.0 / nrm : 1.0;
        }
#pragma unroll
        for (int c = 0; c < 64; ++c) y[c] = c < d ? (double)x[c] * ns - mean[c] : 0.0;
    }
    const double m2 = -2.0 * (double)scale3[0] * (double)scale3[0];
    for (int j = 0; j < C; ++j) {
        double dot = 0.0;
        if (live) {
#pragma unroll
            for (int c = 0; c < 64; ++c) if (c < d) dot = fma(s_cen[j * d + c], y[c], dot);
        }
        side[(size_t)j * stride + r] = (float)(m2 * dot);
    }
}

// flag list of a stage (queries in the order the re-rank flagged them) -> the same queries grouped by bucket, so that
// the 512-query blocks of the next stage are neighbourhoods again (bucket pruning of subset stages)
__global__ __launch_bounds__(256) void fl_labels(const int *__restrict__ flag, int64_t n, const int *__restrict__ lab,
                                                 int *__restrict__ lab_pos, int *__restrict__ counts) {
    __shared__ int hist[LS_CMAX];
    if (threadIdx.x < LS_CMAX) hist[threadIdx.x] = 0;
    __syncthreads();
    for (int rr = 0; rr < LS_ROWS / 256; ++rr) {
        const int64_t i = (int64_t)blockIdx.x * LS_ROWS + rr * 256 + threadIdx.x;
        if (i < n) { const int l = lab[flag[i]]; lab_pos[i] = l; atomicAdd(&hist[l], 1); }
    }
    __syncthreads();
    if (threadIdx.x < LS_CMAX && hist[threadIdx.x]) atomicAdd(&counts[threadIdx.x], hist[threadIdx.x]);
}
__global__ void fl_gather(const int *__restrict__ flag, const int *__restrict__ perm, int64_t n, int *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = flag[perm[i]];
}

// ------------------------------------------------------------------ bucket pruning (exact) -----------
// The locality sort leaves the reference image grouped by bucket and the queries packed by bucket, so whole
// (query block, reference bucket) pairs can be ruled out with the triangle inequality before any product is formed:
//   every query of block b lies within r_b of its centre c_b, every row of bucket j within rho_j of its centre m_j, so
//   all their squared distances are >= LB_bj^2, LB_bj = |c_b - m_j| - r_b - rho_j.
// Pass A runs the filter of every block over its nearest bucket only; the K'-th candidate of a query there is an upper
// bound D_q on its k-th neighbour distance (its filter threshold, plus |q'|^2, plus the filter's error bound).  Pass B
// then visits, for every block, only the buckets with LB_bj^2 <= max_q D_q, as contiguous tile runs of the sorted
// image, each run with its own candidate list; its thresholds start at the pass-A threshold.  A skipped row has
// d^2 >= LB^2 > D_q >= (k-th neighbour distance)^2, so it cannot be among the k nearest, whatever the filter would have
// said about it; for the rows that were visited the usual certificate (min over the lists' thresholds) applies.
// Nothing here depends on how good the buckets are: bad buckets only mean nothing is skipped.
constexpr int IVF_SB = 13;      // pass-B runs (candidate lists) per query block; slot 0 is pass A
constexpr int IVF_CSTR = 128;   // doubles per centre (round 5: the K = 128 family prunes too, d <= 128)
// the IVF_SB + 1 candidate lists per query are the one large buffer of the scheme: not worth more than 16 GiB of workspace
// Largest share (per cent) of the (block, tile) pairs a pruned stage may have to visit.  Pass A + pass B pay a second threshold
// history and several lists per query: an eighth (measured: slower than the full scan at 15 - 25 %).  The single-pass stage
// (round 5) costs ~22 ms + 311 ms x share at 1.5 M x 1.5 M, d = 50 against 269 ms for the unpruned centred scan on the same box
// (ragged 40-cluster mixture, sigma 0.5: 61 % of the pairs, 212 against 269 ms): break-even at 79 %, accepted up to 70 %; the
// gate in front of it (pairs whose balls overlap: a lower bound of the plan's share) lets 60 % through.
// CELLMAP_PRUNE_MAX overrides the 70 (the gate follows 10 below).
// ... and pass A + B of a SEEDED subset stage (its pass A starts from the queries' bounds, not from +inf: no start-up churn): 60 %
// -- 93 k flagged queries of the ragged bench leg, 57 % of the pairs: 57.1 -> 43.2 ms; 53 k at 1.5 M x 1.5 M: 34.8 -> 23.2 ms; a plan
// rejected after pass A costs ~7 ms (CELLMAP_PRUNE_MAX_SUB).
static inline int prune_max_sub_pct() {
    return cm_get_knobs().prune_max_sub;
}
static inline int prune_max_pct(bool single_pass, bool gate = false) {
    const int v = cm_get_knobs().prune_max;
    return single_pass ? (gate ? std::max(12, v - 10) : v) : 12;
}
// Fewest queries of a search whose main stage may be pruned / bucket-centred: 16 384 (65 536 until late in round 5 -- a limit of the
// pass A + B form, whose per-run lists needed many blocks to balance; the single chained pass does not: 40 k x 1.5 M, 40 clusters:
// 28.3 -> 9.3 ms, 40 k x 300 k at sigma 0.3: 9.0 -> 5.5 ms, overlapping clusters and structure-free sets unchanged).
// CELLMAP_PRUNE_MIN_NQ overrides it.
static inline int64_t prune_min_nq() {
    return (int64_t)cm_get_knobs().prune_min_nq;
}
static inline bool ivf_fits(int64_t nq_pad, int Kc_max) {
    return (size_t)(nq_pad + 256 * 512) * (IVF_SB + 1) * (size_t)Kc_max * 4 <= ((size_t)16 << 30);
}

// bounding ball of a group of prepared rows (centred at the reference mean; cosine: normalised first): group g holds
// the rows map[lo..hi), lo/hi from gstart or (gstart == nullptr) 512-row blocks of map[0..n_map); map entries < 0 skip.
// Four small kernels over a (groups, slices) grid: coordinate sums (float64 atomics) -> centre -> max squared distance
// to the centre (atomic max on the bit pattern of a non-negative double) -> radius, rounded up.
template <typename T, bool RAD>
__global__ __launch_bounds__(256) void ivf_ball_pass(const T *__restrict__ X, int d, int metric, const double *__restrict__ mean,
                                                     const int *__restrict__ map, const int *__restrict__ gstart, int64_t n_map,
                                                     int gsize, double *__restrict__ cen, unsigned long long *__restrict__ rad2,
                                                     int *__restrict__ cnt) {
    // lane e holds coordinates e and e + 64 (d <= 128); gsize: rows per group when there is no gstart (the query block of the
    // stage's filter kernel: 512, or 256 for the three-product kernel of the K = 128 family)
    __shared__ double s_acc[4][128];
    __shared__ double s_r[4];
    __shared__ int s_n[4];
    const int g = blockIdx.x, e = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int e2 = e + 64;
    const int64_t lo = gstart ? (int64_t)gstart[g] : (int64_t)g * gsize;
    int64_t hi = gstart ? (int64_t)gstart[g + 1] : lo + gsize;
    if (hi > n_map) hi = n_map;
    const double me = e < d ? mean[e] : 0.0, me2 = e2 < d ? mean[e2] : 0.0;
    const double ce = RAD ? cen[(size_t)g * IVF_CSTR + e] : 0.0, ce2 = RAD ? cen[(size_t)g * IVF_CSTR + e2] : 0.0;
    double acc = 0.0, acc2 = 0.0;   // RAD: running max of the squared distance (acc); otherwise this lane's coordinate sums
    int n = 0;
    for (int64_t r = lo + (int64_t)blockIdx.y * 4 + w; r < hi; r += 4 * (int64_t)gridDim.y) {
        const int src = map ? map[r] : (int)r;
        if (src < 0) continue;   // (wave-uniform)
        double v = e < d ? (double)X[(int64_t)src * d + e] : 0.0;
        double v2 = e2 < d ? (double)X[(int64_t)src * d + e2] : 0.0;
        if (metric == CM_METRIC_COSINE) {
            double s2 = v * v + v2 * v2;
            for (int o = 32; o > 0; o >>= 1) s2 += __shfl_xor(s2, o);
            const double nrm = sqrt(s2);
            const double ns = nrm > 0.0 ? 1.0 / nrm : 1.0;
            v *= ns; v2 *= ns;
        }
        if (RAD) {
            const double t = e < d ? v - me - ce : 0.0, t2 = e2 < d ? v2 - me2 - ce2 : 0.0;
            double s2 = t * t + t2 * t2;
            for (int o = 32; o > 0; o >>= 1) s2 += __shfl_xor(s2, o);
            acc = fmax(acc, s2);
        } else {
            acc += e < d ? v - me : 0.0;
            acc2 += e2 < d ? v2 - me2 : 0.0;
        }
        ++n;
    }
    if (RAD) {
        if (e == 0) s_r[w] = acc;
        __syncthreads();
        if (threadIdx.x == 0)
            atomicMax(&rad2[g], (unsigned long long)__double_as_longlong(fmax(fmax(s_r[0], s_r[1]), fmax(s_r[2], s_r[3]))));
    } else {
        s_acc[w][e] = acc;
        s_acc[w][e2] = acc2;
        if (e == 0) s_n[w] = n;
        __syncthreads();
        if (w < 2) {
            const int c = e + 64 * w;
            if (c < d) atomicAdd(&cen[(size_t)g * IVF_CSTR + c], s_acc[0][c] + s_acc[1][c] + s_acc[2][c] + s_acc[3][c]);
        }
        if (threadIdx.x == 0) atomicAdd(&cnt[g], s_n[0] + s_n[1] + s_n[2] + s_n[3]);
    }
}
__global__ void ivf_ball_zero(int G, double *__restrict__ cen, unsigned long long *__restrict__ rad2, int *__restrict__ cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < G * IVF_CSTR) cen[i] = 0.0;
    if (i < G) { rad2[i] = 0ull; cnt[i] = 0; }
}
__global__ void ivf_ball_centre(int G, double *__restrict__ cen, const int *__restrict__ cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < G * IVF_CSTR) { const int n = cnt[i / IVF_CSTR]; cen[i] = n ? cen[i] / (double)n : 0.0; }
}
__global__ void ivf_ball_radius(int G, double *__restrict__ rad /* in: bits of the max squared distance */) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < G) rad[i] = sqrt(rad[i]) * (1.0 + 1e-9);
}
template <typename T>
static void ivf_ball(hipStream_t st, const T *X, int d, int metric, const double *mean, const int *map, const int *gstart,
                     int64_t n_map, int G, int slices, double *cen, double *rad, int *cnt, int gsize = cmk2::QB) {
    const unsigned nb = (unsigned)((G * IVF_CSTR + 255) / 256);
    auto *rad2 = reinterpret_cast<unsigned long long *>(rad);
    hipLaunchKernelGGL(ivf_ball_zero, dim3(nb), dim3(256), 0, st, G, cen, rad2, cnt);
    hipLaunchKernelGGL((ivf_ball_pass<T, false>), dim3(G, slices), dim3(256), 0, st, X, d, metric, mean, map, gstart, n_map, gsize, cen, rad2, cnt);
    hipLaunchKernelGGL(ivf_ball_centre, dim3(nb), dim3(256), 0, st, G, cen, (const int *)cnt);
    hipLaunchKernelGGL((ivf_ball_pass<T, true>), dim3(G, slices), dim3(256), 0, st, X, d, metric, mean, map, gstart, n_map, gsize, cen, rad2, cnt);
    hipLaunchKernelGGL(ivf_ball_radius, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, st, G, rad);
}

// cheap gate before pass A: tiles of the (block, bucket) pairs whose balls overlap (LB = 0) can never be skipped; if they
// are most of the scan there is nothing to gain and pass A -- whose lists start from +inf thresholds, i.e. pay the
// start-up churn of a scan -- is not worth running
__global__ __launch_bounds__(256) void ivf_overlap(const double *__restrict__ bcen, const double *__restrict__ brad,
                                                   const double *__restrict__ rcen, const double *__restrict__ rrad,
                                                   const int *__restrict__ r_bstart, int C, int d, int BRt,
                                                   unsigned long long *__restrict__ total) {
    __shared__ int s_t[256];
    const int b = blockIdx.x, j = threadIdx.x;
    int tiles = 0;
    if (j < C && r_bstart[j + 1] > r_bstart[j]) {
        double s2 = 0.0;
        for (int c = 0; c < d; ++c) { const double t = bcen[(size_t)b * IVF_CSTR + c] - rcen[(size_t)j * IVF_CSTR + c]; s2 = fma(t, t, s2); }
        if (!(sqrt(s2) > brad[b] + rrad[j])) tiles = (r_bstart[j + 1] - r_bstart[j] + BRt - 1) / BRt;
    }
    s_t[j] = tiles;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (j < o) s_t[j] += s_t[j + o];
        __syncthreads();
    }
    if (j == 0) atomicAdd(total, (unsigned long long)s_t[0]);
}

// pass A: one work item per query block over the tiles of the nearest bucket that holds at least min_rows rows
__global__ __launch_bounds__(256) void ivf_plan_a(const double *__restrict__ bcen, const double *__restrict__ rcen,
                                                  const int *__restrict__ r_bstart, int C, int d, int min_rows, int n_tiles,
                                                  int BRt /* reference rows per tile */, int4 *__restrict__ wl,
                                                  int *__restrict__ jstar) {
    __shared__ double s_d[256];
    __shared__ int s_j[256];
    const int b = blockIdx.x, j = threadIdx.x;
    double dist = INFINITY;
    if (j < C && r_bstart[j + 1] - r_bstart[j] >= min_rows) {
        double s2 = 0.0;
        for (int c = 0; c < d; ++c) { const double t = bcen[(size_t)b * IVF_CSTR + c] - rcen[(size_t)j * IVF_CSTR + c]; s2 = fma(t, t, s2); }
        dist = s2;
    }
    s_d[j] = dist;
    s_j[j] = j;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (j < o && s_d[j + o] < s_d[j]) { s_d[j] = s_d[j + o]; s_j[j] = s_j[j + o]; }
        __syncthreads();
    }
    if (j == 0) {
        int js = s_d[0] < INFINITY ? s_j[0] : -1;
        int t0 = 0, t1 = n_tiles;
        if (js >= 0) { t0 = r_bstart[js] / BRt; t1 = (r_bstart[js + 1] + BRt - 1) / BRt; }
        wl[b] = make_int4(b, t0, t1, 0);
        jstar[b] = js;
    }
}

// pass B: the tile runs every query block still has to visit (see the section comment).  counters[0] = work items,
// counters[2..3] = 64-bit count of the tiles visited by both passes (the caller compares it with a full scan).
__global__ __launch_bounds__(256) void ivf_plan_b(const double *__restrict__ bcen, const double *__restrict__ brad,
                                                  const double *__restrict__ rcen, const double *__restrict__ rrad,
                                                  const int *__restrict__ r_bstart, int C, int d,
                                                  const int *__restrict__ qmap, int64_t nsub, const float *cthr /* (may alias thr_out) */,
                                                  int S, const float *__restrict__ qnorm, const float *__restrict__ qlo,
                                                  const float *__restrict__ qnc, const float *__restrict__ qcen,
                                                  const unsigned *__restrict__ gmax, const float *__restrict__ scale3,
                                                  float c_eps, int level, const int4 *__restrict__ wl_a,
                                                  const int *__restrict__ jstar, int4 *__restrict__ wl_b,
                                                  int *__restrict__ counters, int BRt, int is_f64, int d_rows, int split_len, int sb, int qblk,
                                                  int2 *__restrict__ runs /* chained mode: one item per block, its runs here */,
                                                  int seeds /* 1: single-pass stage -- cthr holds sample-scan seeds, no pass A */,
                                                  const float *__restrict__ qn1, const float *__restrict__ qlo1, float c_eps1,
                                                  float *thr_out /* seeds of a more precise level: widened here, in place (== cthr) */,
                                                  float widen_mul) {
    // sb <= IVF_SB: runs (candidate lists) per block this stage may use -- the re-rank holds (sb + 1) * K' candidates
    // level 4 (direct-difference fp32 filter): thresholds are full squared distances of the rows as given, with the
    // relative error -c_eps and the float64 -> float32 input rounding dlt per coordinate (see knn_rerank)
    __shared__ double s_red[256];
    __shared__ unsigned char s_need[256];
    __shared__ int s_u0[132], s_u1[132];
    const int b = blockIdx.x, tid = threadIdx.x;
    // D = max over the block's live queries of (pass-A threshold in input units + |q'|^2 + filter error bound)
    const double ymax = (double)__uint_as_float(gmax[1]);
    const double rho_y = (double)__uint_as_float(gmax[5]);
    const double dlt = is_f64 ? 2.0 * 5.9604645e-8 * (double)__uint_as_float(gmax[4]) : 0.0;
    double dm = 0.0;
    for (int64_t i = (int64_t)b * qblk + tid; i < (int64_t)(b + 1) * qblk && i < nsub; i += 256) {   // (qblk: queries per block of the stage's filter kernel)
        if (qmap && qmap[i] < 0) continue;
        float tau = cthr[i * S];
        if (thr_out && tau < INFINITY) {
            // The seed is a value of the PLAIN hi*hi filter; this stage's filter (bucket-centred / three products) is more
            // precise, and on tight clusters the plain error is of the order of the gaps between neighbours: the threshold the
            // stage starts from is the seed plus twice the plain bound, so that the rows at or below the seed on the plain
            // filter stay at or below it here.  (Tightness only: D below is computed from the widened value, whatever it is.)
            const double q1 = (double)qn1[i];
            const double e1 = (double)c_eps1 * (ymax * ymax + 2.0 * q1 * ymax + 1e-30) + 2.0 * 1.001 * ymax * ((double)qlo1[i] + q1 * rho_y);
            const double w = (double)tau + (double)widen_mul * e1 / (double)scale3[2];
            float wf = (float)w;
            if ((double)wf < w) wf = nextafterf(wf, INFINITY);
            tau = wf;
            thr_out[i] = wf;
        }
        double D = INFINITY;
        if (tau < INFINITY && level == 4) {
            const double t = (double)tau;
            D = t * (1.0 + 2.0 * (double)(-c_eps)) + 2.0 * (2.0 * dlt * sqrt((double)d_rows) * sqrt(t) + (double)d_rows * dlt * dlt);
        } else if (tau < INFINITY) {
            const double qn = (double)qnorm[i];
            double eps;   // the re-rank's bound (knn_rerank) with the global |y'| maximum
            if (qnc) {
                const double qc = (double)qnc[i], cn = (double)qcen[i];
                eps = (double)c_eps * (ymax * ymax + 2.0 * (cn + qc) * ymax + 1e-30) + 2.4e-7 * cn * ymax +
                      2.0 * 1.001 * ymax * ((level == 1 ? (double)qlo[i] : 0.0) + (level <= 2 ? qc * rho_y : 0.0));
            } else {
                eps = (double)c_eps * (ymax * ymax + 2.0 * qn * ymax + 1e-30) +
                      2.0 * 1.001 * ymax * ((level == 1 ? (double)qlo[i] : 0.0) + (level <= 2 ? qn * rho_y : 0.0));
            }
            D = (double)tau * (double)scale3[2] + qn * qn + eps;
        }
        dm = fmax(dm, D);
    }
    s_red[tid] = dm;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) s_red[tid] = fmax(s_red[tid], s_red[tid + o]);
        __syncthreads();
    }
    const double Dmax = s_red[0];
    const int js = seeds ? 0 : jstar[b];
    bool need = false;
    if (tid < C && r_bstart[tid + 1] > r_bstart[tid]) {
        double s2 = 0.0;
        for (int c = 0; c < d; ++c) { const double t = bcen[(size_t)b * IVF_CSTR + c] - rcen[(size_t)tid * IVF_CSTR + c]; s2 = fma(t, t, s2); }
        const double lb = sqrt(s2) * (1.0 - 1e-9) - (brad[b] + rrad[tid]) * (1.0 + 1e-9);
        need = !(lb > 0.0 && lb * lb > Dmax * (1.0 + 1e-6) + 1e-30);
    }
    s_need[tid] = need ? 1 : 0;
    __syncthreads();
    if (tid != 0) return;
    const int a0 = seeds ? 0 : wl_a[b].y, a1 = seeds ? 0 : wl_a[b].z;   // (single-pass stage: nothing visited yet)
    if (js < 0) {   // pass A already visited every tile
        atomicAdd(reinterpret_cast<unsigned long long *>(counters + 2), (unsigned long long)(a1 - a0));
        return;
    }
    // maximal runs of needed buckets, in tiles (neighbouring buckets share their boundary tile)
    int nr = 0;
    for (int j = 0; j < C; ++j) {
        if (!s_need[j]) continue;
        const int t0 = r_bstart[j] / BRt, t1 = (r_bstart[j + 1] + BRt - 1) / BRt;
        if (nr > 0 && t0 <= s_u1[nr - 1]) { if (t1 > s_u1[nr - 1]) s_u1[nr - 1] = t1; }
        else { s_u0[nr] = t0; s_u1[nr] = t1; ++nr; }
    }
    // minus the tiles of pass A
    int p0[134], p1[134];
    int np = 0;
    for (int r = 0; r < nr; ++r) {
        const int u0 = s_u0[r], u1 = s_u1[r];
        if (u1 <= a0 || u0 >= a1) { p0[np] = u0; p1[np] = u1; ++np; continue; }
        if (u0 < a0) { p0[np] = u0; p1[np] = a0; ++np; }
        if (u1 > a1) { p0[np] = a1; p1[np] = u1; ++np; }
    }
    if (runs) {   // chained mode (round 5): ONE work item and ONE candidate list for all the runs of the block
        long long tiles_c = a1 - a0;
        if (np > 0) {
            const int head = atomicAdd(&counters[0], 1);
            const int rbase = atomicAdd(&counters[1], np);
            for (int i = 0; i < np; ++i) { runs[rbase + i] = make_int2(p0[i], p1[i]); tiles_c += p1[i] - p0[i]; }
            wl_b[head] = make_int4(b, rbase, rbase + np, seeds ? 0 : 1);
        }
        atomicAdd(reinterpret_cast<unsigned long long *>(counters + 2), (unsigned long long)tiles_c);
        return;
    }
    // at most IVF_SB runs: close the smallest gaps (never the one that holds the pass-A tiles)
    while (np > sb) {
        int bi = -1, bg = 0x7fffffff;
        for (int i = 0; i + 1 < np; ++i) {
            const int gap = p0[i + 1] - p1[i];
            if (p1[i] <= a0 && p0[i + 1] >= a1) continue;
            if (gap < bg) { bg = gap; bi = i; }
        }
        if (bi < 0) break;   // (cannot happen: at most one gap is excluded and IVF_SB >= 2)
        p1[bi] = p1[bi + 1];
        for (int i = bi + 1; i + 1 < np; ++i) { p0[i] = p0[i + 1]; p1[i] = p1[i + 1]; }
        --np;
    }
    // long runs are halved while list slots are left: a block that cannot rule out much would otherwise occupy one
    // compute unit for a whole scan while the others idle
    while (np > 0 && np < sb) {
        int bi = 0;
        for (int i = 1; i < np; ++i) if (p1[i] - p0[i] > p1[bi] - p0[bi]) bi = i;
        const int len = p1[bi] - p0[bi];
        if (len <= split_len) break;
        p0[np] = p0[bi] + len / 2; p1[np] = p1[bi];
        p1[bi] = p0[np];
        ++np;
    }
    long long tiles = a1 - a0;
    if (np > 0) {
        const int base = atomicAdd(&counters[0], np);
        for (int i = 0; i < np; ++i) { wl_b[base + i] = make_int4(b, p0[i], p1[i], 1 + i); tiles += p1[i] - p0[i]; }
    }
    atomicAdd(reinterpret_cast<unsigned long long *>(counters + 2), (unsigned long long)tiles);
}

// every R-th row of the (sorted) reference image: the sample whose scan seeds the thresholds of the main scan.
// Round 5: the sample is INTERLEAVED over the 32-row sub-tiles of its image -- sub-tile s holds the sample rows s, s + n_st,
// s + 2 n_st, ... (n_st sub-tiles, rotated: below) -- because the seeding scan keeps sub-tile MINIMA: in sorted order the ~50 sample rows of a
// small cluster sat in two sub-tiles, the lists saw two of them and the seed was the distance to the next cluster (useless as a
// pruning radius: 90 % of the pairs left at 260 k rows in 40 clusters); spread out, every sub-tile holds its share of every
// cluster and the 4th smallest minimum of a lane is its 4th or 5th nearest sample row.  Entries beyond the sample are -1
// (padding rows: never candidates).  Any order is valid -- the seed only has to have sample rows at or below it.
__global__ void sample_map(const int *__restrict__ rperm, int64_t n_s, int R, int64_t n_img, int *__restrict__ sperm) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // position in the sample image
    if (p >= n_img) return;
    // (row r of sub-tile st holds sample row ((r - st) mod 32) n_st + st: consecutive sample rows go to consecutive sub-tiles AND
    //  rotate through the 32 rows of a sub-tile, whose two 16-row halves belong to the two lanes of a query -- the seed is the
    //  larger of the two lanes' 4th entries, so a cluster all of whose rows sat in one half would not seed at all)
    const int64_t n_st = n_img / 32, st = p / 32, r = p % 32;
    const int64_t i = ((r - st) & 31) * n_st + st;
    sperm[p] = i < n_s ? (rperm ? rperm[i * R] : (int)(i * R)) : -1;
}

// side values of the sample image's rows for the bucket-centred sample scan: row p of the image is sorted row i R (sample_map's
// interleave), bucket j's value for it sits at side[j stride + i R]
__global__ void side_sample_gather(const float *__restrict__ side, int64_t stride, int64_t n_s, int R, int64_t n_img,
                                   float *__restrict__ side_s) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_img) return;
    const int64_t n_st = n_img / 32, st = p / 32, r = p % 32;
    const int64_t i = ((r - st) & 31) * n_st + st;
    side_s[(size_t)blockIdx.y * n_img + p] = i < n_s ? side[(size_t)blockIdx.y * stride + i * R] : 0.f;
}

// Dense seeding scan of a single-pass pruned stage: every R2-th tile of each run of a chained item.  The seeding kernel walks
// "strided tile numbers" u (tile = R2 u: P1Args::perm_mul), so run (t0, t1) becomes (ceil(t0 / R2), ceil(t1 / R2)): the
// multiples of R2 inside [t0, t1) -- distinct tiles for distinct runs (the runs of a block are disjoint).
__global__ void ivf_seed_runs(const int2 *__restrict__ runs, int n_runs, int R2, int2 *__restrict__ out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_runs) return;
    const int2 t = runs[r];
    out[r] = make_int2((t.x + R2 - 1) / R2, (t.y + R2 - 1) / R2);
}

// thr[i] = min(thr[i], dense seed [+ widen x the plain hi*hi bound: see ivf_plan_b]) -- the starting thresholds of the one pass
__global__ void seed_combine(float *__restrict__ thr, const float *__restrict__ dense, int64_t n, const float *__restrict__ qn1,
                             const float *__restrict__ qlo1, float c_eps1, float widen_mul, const unsigned *__restrict__ gmax,
                             const float *__restrict__ scale3) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = dense[i];
    if (!(s < INFINITY)) return;
    if (widen_mul > 0.f) {
        const double ymax = (double)__uint_as_float(gmax[1]), rho_y = (double)__uint_as_float(gmax[5]);
        const double q1 = (double)qn1[i];
        const double e1 = (double)c_eps1 * (ymax * ymax + 2.0 * q1 * ymax + 1e-30) + 2.0 * 1.001 * ymax * ((double)qlo1[i] + q1 * rho_y);
        const double w = (double)s + (double)widen_mul * e1 / (double)scale3[2];
        s = (float)w;
        if ((double)s < w) s = nextafterf(s, INFINITY);
    }
    thr[i] = fminf(thr[i], s);
}

__global__ void fill_f32(float *__restrict__ p, int64_t n, float v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// cen_side for the (bucket, reference row) pairs a work list visits only (bucket pruning): item = (query block, first
// tile, end tile, .); same arithmetic as cen_side.  Round 5: a bucket holds ~11 query blocks whose runs overlap almost
// entirely, and pass B revisits what pass A's neighbours computed -- 88 M (bucket, row) values per step at the 40-cluster
// config-2 shape, 10 ms, most of them several times over.  A (bucket, tile) pair is now CLAIMED (one byte per pair, zeroed once
// per call) by the first workgroup that reaches it; everybody else skips the tile.  The values of a pair are the same whoever
// computes them.
// (round 5, second form: MARK then COMPUTE.  The first form had every (item, slice) workgroup walk its tiles and claim each with
//  an atomic behind two barriers -- 98 k workgroups x 9 tiles, 4 ms per call at the config-2 shape for 0.1 ms of arithmetic.)
// cen_side_mark: one bit per (bucket, tile) pair the work list visits, in `want` (bits of a bucket start at a word boundary:
// wpb words per bucket); pairs whose values exist (`done`) are left out.
__global__ __launch_bounds__(128) void cen_side_mark(const int *__restrict__ blk_bucket, const int4 *__restrict__ wl, int n_items,
                                                     const unsigned *__restrict__ done, unsigned *__restrict__ want, int wpb,
                                                     const int2 *__restrict__ runs /* chained items: (block, first run, end run, .) */) {
    if ((int)blockIdx.x >= n_items) return;
    const int4 w = wl[blockIdx.x];
    const size_t base = (size_t)blk_bucket[w.x] * (size_t)wpb;
    const int n_runs = runs ? w.z - w.y : 1;
    for (int rr = 0; rr < n_runs; ++rr) {
        const int t_b = runs ? runs[w.y + rr].x : w.y, t_e = runs ? runs[w.y + rr].y : w.z;
        // one thread per 32-tile word of the run
        for (int wd = (t_b >> 5) + (int)(blockIdx.y * blockDim.x + threadIdx.x); wd <= ((t_e - 1) >> 5); wd += (int)(gridDim.y * blockDim.x)) {
            const int lo = wd * 32 < t_b ? t_b - wd * 32 : 0, hi = (wd + 1) * 32 > t_e ? t_e - wd * 32 : 32;
            unsigned m = (hi >= 32 ? 0xffffffffu : ((1u << hi) - 1u)) & ~((1u << lo) - 1u);
            m &= ~done[base + wd];
            if (m & ~want[base + wd]) atomicOr(&want[base + wd], m);
        }
    }
}

// cen_side_compute: workgroup (word, bucket) computes the side values of the marked pairs of its word (same arithmetic as
// cen_side), then moves the bits from `want` to `done` (it owns the word).
template <typename T>
__global__ __launch_bounds__(128) void cen_side_compute(const T *__restrict__ R, int64_t n_ref, int d, int metric,
                                                        const double *__restrict__ mean, const int *__restrict__ rperm,
                                                        const double *__restrict__ cen, const float *__restrict__ scale3, int64_t stride,
                                                        float *__restrict__ side, unsigned *__restrict__ done,
                                                        unsigned *__restrict__ want, int wpb) {
    __shared__ double s_c[64];
    const int j = blockIdx.y, wd = blockIdx.x;
    const unsigned m = want[(size_t)j * wpb + wd];   // (uniform)
    if (!m) return;
    if (threadIdx.x < 64) s_c[threadIdx.x] = (int)threadIdx.x < d ? cen[(size_t)j * d + threadIdx.x] : 0.0;
    __syncthreads();
    const double m2 = -2.0 * (double)scale3[0] * (double)scale3[0];
    for (unsigned rest = m; rest; rest &= rest - 1u) {
        const int t = wd * 32 + __builtin_ctz(rest);
        const int64_t r = (int64_t)t * cmk2::BR + threadIdx.x;   // (BR = 128 rows per tile = one per thread)
        if (r >= stride) continue;
        double dot = 0.0;
        if (r < n_ref) {
            const T *x = R + (int64_t)(rperm ? rperm[r] : r) * d;
            double ns = 1.0;
            if (metric == CM_METRIC_COSINE) {
                double s2 = 0.0;
                for (int c = 0; c < d; ++c) { double v = (double)x[c]; s2 = fma(v, v, s2); }
                const double nrm = sqrt(s2);
                ns = nrm > 0.0 ? 1.0 / nrm : 1.0;
            }
            for (int c = 0; c < d; ++c) dot = fma(s_c[c], (double)x[c] * ns - mean[c], dot);
        }
        side[(size_t)j * stride + r] = (float)(m2 * dot);
    }
    __syncthreads();
    if (threadIdx.x == 0) { done[(size_t)j * wpb + wd] |= m; want[(size_t)j * wpb + wd] = 0u; }
}

__global__ void strided_rows(int *__restrict__ out, int n, int64_t stride) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int)(i * stride);
}

// ---------------------------------------------------------------------------------------------------------------
// Staged exact search for d <= 61 (split-fp16 MFMA kernel family).  The filter only has to be good enough to be
// certified, so the search climbs a ladder of filters of increasing precision and cost, and only the queries that fail
// the certificate of one stage go to the next:
//   level 1: hi*hi split product (1/3 of the MFMA work, error ~2^-10 |q'||y'|)
//   level 3: three split products (error ~5e-6 |q'||y'|)
//   level 4: direct-difference fp32 filter on the VALU (relative error (d+4)*2^-23 -- immune to the cancellation in
//            |y|^2 - 2q.y that tight, off-centre clusters cause; ~15x the cost of level 1)
//   then the exact float64 scan for whatever is left.
// A pilot on a strided sample of the queries estimates the failure rate of the cheap levels and picks the level the
// whole set starts at.  Every stage writes exact results for the queries it certifies; exactness never depends on the
// path taken.  Host synchronisations: one per stage (flag counter).
template <typename T>
cm_status knn_run_staged(cm_handle_t h, const T *d_ref, int64_t n_ref, const T *d_qry, int64_t nq, int d, int k, int metric,
                         int form, int64_t idx_offset, int flags, void *d_out_dist, int64_t *d_out_idx) {
    hipStream_t st = h->stream;
    const int bs = 256;
    const bool lp_metric = metric == CM_METRIC_MANHATTAN || metric == CM_METRIC_CHEBYSHEV || metric == CM_METRIC_CANBERRA || metric == CM_METRIC_BRAYCURTIS;
    const int path = lp_metric ? 2 : phase1_path(d, k);   // (Manhattan / Chebyshev: the VALU direct kernel is the only filter)
    // dynamic LDS of prep_rows_f16 when a block's source rows fit (64 KB without opting in): coalesced staging
    const size_t prep_lds = (size_t)bs * (d + 1) * sizeof(T) <= 64 * 1024 ? (size_t)bs * (d + 1) * sizeof(T) : 0;
    const bool wide = path == 3;     // 62 <= d <= 128: K = 128 kernels, no direct level
    const bool lowp = path == 2;     // d <= 15 (or CELLMAP_PHASE1=direct): no MFMA levels, the direct kernel is the filter
    const int dl = lowd_dpad(d);     // row stride of the direct kernel
    const bool have_direct = dl <= 128;   // (rows of up to 64 floats: two queries per lane; up to 128: one)
    // level 0: uniform-grid search (grid_knn.hip) for spatial coordinates; worth its set-up from a few thousand rows on
    const bool have_grid = lowp && d <= 3 && metric == CM_METRIC_EUCLIDEAN && n_ref >= 4096 &&
                           !cm_get_knobs().lowd_brute;
    const int dbg = cm_get_knobs().dbg;
    int forced = cm_get_knobs().filter_products;
    if (forced < 1 || forced > 5 || (forced == 4 && !have_direct) || lowp) forced = 0;   // (5 = bucket-centred hi*hi level)
    if (wide && forced == 2) forced = 3;   // the K = 128 family has no two-product kernel

    KnnPlan2 pf;   // plan of the full query set at level 1 (widest candidate lists): sizes the buffers
    CM_TRY(make_plan2(h, n_ref, nq, d, k, 1, 1, &pf));
    // Long lists at the bucket-centred level (round 5).  A single list of K' = k + max(k, 30) candidates certifies a query only if
    // its k-th and K'-th neighbours lie farther apart than twice the filter error; on very tight clusters (sigma 0.1 at a
    // spread of 5: 93 % of 1.5 M queries) they do not, the level was written off (hint -5) and the direct-difference level
    // took the whole set: 297 ms.  With 80 more candidates the centred level certifies 97 % of that set and the call takes
    // 86 ms -- at the price of a re-rank that gathers 140 candidates instead of 60 (19 ms instead of 6), which is why the
    // long lists are not the default: the level is first tried with the short lists; if it fails for most queries the handle
    // remembers "centred, long lists" (hint 6) for the next call of the same shape, and only if that fails too, -5.
    const bool cen_long = !cm_get_knobs().no_hint && h->staged_hint_d == d && h->staged_hint_k == k && h->staged_hint_metric == metric &&
                          h->staged_hint_level == 6 && h->staged_long_nref == n_ref && h->staged_long_nq == nq &&   // (another data set of the same width starts with the short lists again)
                          k + (k > 30 ? k : 30) + 80 + 2 * cmk2::CL <= 512;
    const int kc5_extra = cen_long ? 80 : 0;
    const int Kc_max = pf.Kc + 16 + kc5_extra;   // (+16: the bucket-centred level uses k + max(30, k) candidates)
    const int64_t nq_pad = pf.nq_pad;
    // subsets may use more reference splits than the full set: rows of the candidate buffers
    // (x up to 4 reference splits of the four-accumulator stage, see run_stage; bounded for very large query sets)
    const int64_t cand_rows = std::max<int64_t>(std::max<int64_t>(nq_pad * pf.S, nq_pad + (int64_t)cmk2::QB * 512),
                                                std::min<int64_t>(4 * (nq_pad + 1024), nq_pad + 1024 + 2000000));
    const int64_t pack_slack = (int64_t)LS_CMAX * cmk2::QB;   // bucket-aligned packing pads every bucket to whole blocks
    float *qnorm = (float *)cm_ws_alloc(h, sizeof(float) * (nq_pad + pack_slack));
    float *qlo = (float *)cm_ws_alloc(h, sizeof(float) * (nq_pad + pack_slack));   // |lo part| of every prepared query row
    int *cand = (int *)cm_ws_alloc(h, sizeof(int) * cand_rows * Kc_max);
    float *cthr = (float *)cm_ws_alloc(h, sizeof(float) * cand_rows);
    int *flag_a = (int *)cm_ws_alloc(h, sizeof(int) * (nq + 1));
    int *flag_b = (int *)cm_ws_alloc(h, sizeof(int) * (nq + 1));
    const int n_pilot = nq >= 32768 ? 2048 : 0;
    int *sample = (int *)cm_ws_alloc(h, sizeof(int) * 4096);
    double *bound = (double *)cm_ws_alloc(h, sizeof(double) * (nq + 1));
    float *seedb = (float *)cm_ws_alloc(h, sizeof(float) * (nq + 1));                     // next-stage threshold seed per query
    float *seed_sub = (float *)cm_ws_alloc(h, sizeof(float) * (nq_pad + pack_slack));      // ... gathered in a subset's packed order
    unsigned *words = (unsigned *)cm_ws_alloc(h, 128);  // [0..1] gmax, [2] flag count, [3] overflow, [4] maxabs, [8..10] scale
    int *fb_count = (int *)cm_ws_alloc(h, sizeof(int) * FBS_NQ);
    double *fb_v = (double *)cm_ws_alloc(h, sizeof(double) * (size_t)FBS_NQ * FB_CAP);
    int *fb_i = (int *)cm_ws_alloc(h, sizeof(int) * (size_t)FBS_NQ * FB_CAP);
    int *ovf_list = (int *)cm_ws_alloc(h, sizeof(int) * OVF_CAP);   // queries with more than FB_CAP rows inside their bound
    // any tile image: <= 256 B per row (K <= 64) / 512 B per row + side block (K = 128)
    const bool big_rows = wide || dl > 64;   // 512 B per prepared row: K = 128 images, direct rows of more than 64 floats
    char *Rp = (char *)cm_ws_alloc(h, (size_t)pf.n_tiles * (big_rows ? (size_t)cmk2::wide_tile_bytes() : (size_t)cmk2::tile_bytes(4)));
    char *Qp = (char *)cm_ws_alloc(h, (size_t)(nq_pad + pack_slack) * (big_rows ? 512 : 256));
    // (x 128 queries per wave: the four-accumulator kernel; the two-accumulator kernels use half of it)
    uint2 *kept = (uint2 *)cm_ws_alloc(h, sizeof(uint2) * (size_t)320 * cmk2::NW * 128 * Kc_max);
    uint2 *lbuf = (uint2 *)cm_ws_alloc(h, sizeof(uint2) * (size_t)320 * cmk2::NW * 128 * 2 * cmk2::CL);
    double *cpart = (double *)cm_ws_alloc(h, sizeof(double) * CMEAN_BLOCKS * CMEAN_COLS);
    double *d_mean = (double *)cm_ws_alloc(h, sizeof(double) * CMEAN_COLS);
    if (!qnorm || !qlo || !cand || !cthr || !flag_a || !flag_b || !sample || !bound || !words || !fb_count || !fb_v || !fb_i || !ovf_list || !Rp ||
        !Qp || !kept || !lbuf || !cpart || !d_mean || !seedb || !seed_sub)
        CM_FAIL(h, CM_E_NOMEM, "k-NN: device workspace exhausted");
    if (pf.grid > 320) CM_FAIL(h, CM_E_INTERNAL, "k-NN: unexpected resident workgroup count %d", pf.grid);
    float *scale3 = reinterpret_cast<float *>(words + 8);
    cm_mark(h, 0);
    hipLaunchKernelGGL(zero_words, dim3(1), dim3(64), 0, st, words, 32);
    hipLaunchKernelGGL(init_bounds, dim3((unsigned)((nq + 1 + 255) / 256)), dim3(256), 0, st, bound, seedb, nq + 1);
    // centre at the reference mean (translation leaves every distance unchanged), common power-of-two scale
    hipLaunchKernelGGL(colmean_partial<T>, dim3(CMEAN_BLOCKS), dim3(256), 0, st, d_ref, n_ref, d, metric, cpart);
    hipLaunchKernelGGL(colmean_final, dim3(1), dim3(CMEAN_COLS), 0, st, cpart, n_ref, d, d_mean);
    // (the direct / grid filters work on the rows as given: their rounding bound needs max |coordinate| of the raw values)
    const double *mean_ma = lowp ? (const double *)nullptr : (const double *)d_mean;
    if (metric == CM_METRIC_EUCLIDEAN) {
        hipLaunchKernelGGL(maxabs_flat<T>, dim3(2048), dim3(256), 0, st, d_ref, n_ref * (int64_t)d, d, mean_ma, words);
        hipLaunchKernelGGL(maxabs_flat<T>, dim3(2048), dim3(256), 0, st, d_qry, nq * (int64_t)d, d, mean_ma, words);
    } else {
        hipLaunchKernelGGL(maxabs_rows<T>, dim3((unsigned)((n_ref + bs - 1) / bs)), dim3(bs), 0, st, d_ref, n_ref, d, metric, mean_ma, words);
        hipLaunchKernelGGL(maxabs_rows<T>, dim3((unsigned)((nq + bs - 1) / bs)), dim3(bs), 0, st, d_qry, nq, d, metric, mean_ma, words);
    }
    hipLaunchKernelGGL(f16_scale_kernel, dim3(1), dim3(64), 0, st, words, scale3, 12);
    CM_HIP(h, hipGetLastError());
    // locality sort (see ls_assign): permutation of the reference rows for the tile images, of the queries for the order
    // in which full-set stages pack them into waves
    int *rperm = nullptr, *qperm = nullptr, *q_bstart = nullptr, *q_bcount = nullptr, *r_bstart = nullptr;
    int *q_lab = nullptr, *flag_s = nullptr, *fl_tmp = nullptr, *fl_cnt = nullptr;   // bucket of every query; sorted flag list
    // The sort only pays on data with cluster structure (bursts of candidates, bucket pruning): 3.5 ms of 243 at config 2 for
    // bucketing, the two permutations and the re-rank's row map, plus the pruning gate (1 ms) and the pilot (3 ms).
    // Round 4: whether the data has any is decided here, inside the call, by the structure pre-test on a sample (sg_*; one
    // 16-byte host read) -- no handle memory involved, so a first call takes the same route as the twentieth.
    bool no_structure = false;
    const bool sort_wanted = !lowp && n_ref >= 65536 && nq >= 4096 && !cm_get_knobs().no_sort;
    if (sort_wanted && d <= SG_STR && !cm_get_knobs().no_pretest && !forced && !(flags & CM_KNN_FORCE_FALLBACK)) {
        const int C = LS_CMAX;
        const int ns_r = (int)std::min<int64_t>(SG_N, n_ref), ns_q = (int)std::min<int64_t>(SG_N, nq);
        int *sg_lab = (int *)cm_ws_alloc(h, sizeof(int) * 2 * SG_N);
        int *sg_cnt = (int *)cm_ws_alloc(h, sizeof(int) * 2 * LS_CMAX + 16);
        double *sg_cen = (double *)cm_ws_alloc(h, sizeof(double) * 2 * (size_t)LS_CMAX * SG_STR);
        double *sg_rad = (double *)cm_ws_alloc(h, sizeof(double) * 2 * LS_CMAX);
        if (!sg_lab || !sg_cnt || !sg_cen || !sg_rad) CM_FAIL(h, CM_E_NOMEM, "k-NN: device workspace exhausted");
        unsigned long long *sg_out = reinterpret_cast<unsigned long long *>(sg_cnt + 2 * LS_CMAX);
        CM_HIP(h, hipMemsetAsync(sg_cnt, 0, sizeof(int) * 2 * LS_CMAX + 16, st));
        for (int side = 0; side < 2; ++side) {
            const T *X = side ? d_qry : d_ref;
            const int ns = side ? ns_q : ns_r;
            const int64_t stride = (side ? nq : n_ref) / ns;
            hipLaunchKernelGGL(sg_assign<T>, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, st, X, stride, ns, d, d_ref, n_ref / C, C,
                               sg_lab + side * SG_N, sg_cnt + side * LS_CMAX);
            hipLaunchKernelGGL(sg_ball<T>, dim3(C), dim3(256), 0, st, X, stride, ns, d, metric, (const int *)(sg_lab + side * SG_N),
                               sg_cen + (size_t)side * LS_CMAX * SG_STR, sg_rad + side * LS_CMAX);
        }
        hipLaunchKernelGGL(sg_overlap, dim3(C), dim3(256), 0, st, (const double *)(sg_cen + (size_t)LS_CMAX * SG_STR), (const double *)(sg_rad + LS_CMAX),
                           (const int *)(sg_cnt + LS_CMAX), (const double *)sg_cen, (const double *)sg_rad, (const int *)sg_cnt, C, d, sg_out);
        CM_HIP(h, hipGetLastError());
        int *ho = h->d_nflag_pinned + 8;   // (two 64-bit words: pinned ints 8..11)
        CM_HIP(h, hipMemcpyAsync(ho, sg_out, 16, hipMemcpyDeviceToHost, st));
        CM_HIP(h, hipStreamSynchronize(st));
        unsigned long long ov = 0, tot = 0;
        memcpy(&ov, ho, 8); memcpy(&tot, ho + 2, 8);
        no_structure = tot > 0 && ov * 16ull >= tot * 15ull;
        if (dbg & 16) fprintf(stderr, "[cellmap dbg] structure pre-test: bucket balls overlap for %.2f %% of the (query, reference) sample pairs -> %s\n",
                              tot ? 100.0 * (double)ov / (double)tot : 0.0, no_structure ? "plain route" : "sort + pruning gate + pilot");
    }
    const bool sort_skip = no_structure;
    if (sort_wanted && !sort_skip) {
        const int C = LS_CMAX;
        int *lab = (int *)cm_ws_alloc(h, sizeof(int) * (size_t)std::max<int64_t>(n_ref, nq));
        int *cnt_c = (int *)cm_ws_alloc(h, sizeof(int) * 2 * LS_CMAX);
        rperm = (int *)cm_ws_alloc(h, sizeof(int) * (size_t)n_ref);
        qperm = (int *)cm_ws_alloc(h, sizeof(int) * (size_t)nq);
        r_bstart = (int *)cm_ws_alloc(h, sizeof(int) * (LS_CMAX + 1));
        if (nq >= prune_min_nq()) {
            flag_s = (int *)cm_ws_alloc(h, sizeof(int) * (size_t)(nq + 1));
            fl_tmp = (int *)cm_ws_alloc(h, sizeof(int) * 2 * (size_t)(nq + 1));
            fl_cnt = (int *)cm_ws_alloc(h, sizeof(int) * 2 * LS_CMAX);
            if (!flag_s || !fl_tmp || !fl_cnt) CM_FAIL(h, CM_E_NOMEM, "k-NN: device workspace exhausted");
        }
        q_lab = lab;   // (after the loop below: the bucket of every query)
        if (!lab || !cnt_c || !rperm || !qperm || !r_bstart) CM_FAIL(h, CM_E_NOMEM, "k-NN: device workspace exhausted");
        auto assign = ls_assign<T>;
        int *b_order = (int *)cm_ws_alloc(h, sizeof(int) * LS_CMAX);
        if (!b_order) CM_FAIL(h, CM_E_NOMEM, "k-NN: device workspace exhausted");
        hipLaunchKernelGGL(ls_chain<T>, dim3(1), dim3(64), 0, st, d_ref, n_ref / C, d, C, b_order);
        if (cm_get_knobs().no_chain) hipLaunchKernelGGL(strided_rows, dim3(1), dim3(256), 0, st, b_order, C, (int64_t)1);   // (experiments)
        for (int side = 0; side < 2; ++side) {
            const T *X = side ? d_qry : d_ref;
            const int64_t nn = side ? nq : n_ref;
            int *perm = side ? qperm : rperm;
            CM_HIP(h, hipMemsetAsync(cnt_c, 0, sizeof(int) * 2 * LS_CMAX, st));
            hipLaunchKernelGGL(assign, dim3((unsigned)((nn + LS_ROWS - 1) / LS_ROWS)), dim3(256), 0, st, X, nn, d, d_ref, n_ref / C, C, lab, cnt_c, (const int *)b_order);
            if (side == 0) hipLaunchKernelGGL(ls_starts, dim3(1), dim3(64), 0, st, cnt_c, C, r_bstart);   // (bucket pruning)
            hipLaunchKernelGGL(ls_offsets, dim3(1), dim3(64), 0, st, cnt_c, C, cnt_c + LS_CMAX);
            hipLaunchKernelGGL(ls_scatter, dim3((unsigned)((nn + LS_ROWS - 1) / LS_ROWS)), dim3(256), 0, st, lab, nn, cnt_c + LS_CMAX, perm);
        }
        // query buckets: [start of bucket j in qperm] for the centred mode (cnt_c still holds the query counts)
        q_bstart = (int *)cm_ws_alloc(h, sizeof(int) * (LS_CMAX + 1));
        q_bcount = cnt_c;
        if (!q_bstart) CM_FAIL(h, CM_E_NOMEM, "k-NN: device workspace exhausted");
        hipLaunchKernelGGL(ls_starts, dim3(1), dim3(64), 0, st, cnt_c, C, q_bstart);
        CM_HIP(h, hipGetLastError());
    }
    // bucket-centred hi*hi level (level 5): state built on first use
    // (worth it from ~131k queries on: every bucket is padded to whole 512-query blocks; smaller sets only when forced)
    // The bucket-centred level exists from 65 536 queries on (round 5, late; 131 072 before): that is where bucket pruning starts, and
    // a centred stage whose blocks are bucket-aligned prunes an order of magnitude better than the plain levels' locality-sorted
    // blocks (100 k x 1.5 M, 40 clusters: 45.1 -> 10.4 ms; 70 k queries at sigma 0.2: 50.9 -> 15.6 ms).  Below 131 072 queries the
    // bucket-aligned packing pads the set by up to a factor of two, so there the level is only kept if the pruning gate says it
    // will pay (cen_abandon below): overlapping clusters go back to the three-product level at the price of the gate (~2 ms).
    const int64_t cen_min_nq = prune_min_nq();
    const bool have_cen = rperm && !wide && !lowp && !cm_get_knobs().no_centre && (nq >= cen_min_nq || forced == 5);
    bool cen_abandon = false;
    int *qpack = nullptr, *blk_bucket = nullptr;
    double *cen = nullptr;
    float *side = nullptr, *qnc = nullptr, *qcen = nullptr;
    int64_t n_packed = 0, side_stride = 0;
    bool side_full = false;   // side values computed for every (bucket, reference row) pair
    unsigned *side_claimed = nullptr;   // (bucket, tile) pairs whose side values exist already + pairs wanted by the current work list (cen_side_mark)
    // bucket pruning (ivf_*): full-set stages of the K <= 64 MFMA levels; state built on first use
    const bool hint_same = h->staged_hint_d == d && h->staged_hint_k == k && h->staged_hint_metric == metric;
    // (the "did not pay" memory belongs to one problem: same (d, k, metric) AND the same set sizes)
    if (!hint_same || h->staged_ivf_nref != n_ref || h->staged_ivf_nq != nq) h->staged_ivf_skip = 0;
    // (round 5: the K = 128 family prunes too -- its hi*hi level on 512-query blocks like the K <= 64 kernels, its three-product
    //  level on the 256-query blocks of knn_cand_f16w)
    bool ivf_try = rperm && !lowp && nq >= prune_min_nq() && ivf_fits(nq_pad, Kc_max) && !forced && !cm_get_knobs().no_ivf;
    if (ivf_try && h->staged_ivf_skip > 0 && !cm_get_knobs().no_hint) { --h->staged_ivf_skip; ivf_try = false; }   // (it did not pay last time)
    const int ivf_S = IVF_SB + 1;
    double *ivf_rcen = nullptr, *ivf_rrad = nullptr, *ivf_bcen = nullptr, *ivf_brad = nullptr;
    int *ivf_rcnt = nullptr, *ivf_bcnt = nullptr, *ivf_jstar = nullptr, *ivf_ctr = nullptr, *ivf_cand = nullptr;
    int4 *ivf_wla = nullptr, *ivf_wlb = nullptr;
    int2 *ivf_runs = nullptr;
    float *ivf_cthr = nullptr;
    size_t ivf_list_cap = 0;   // candidate lists (entries of Kc_max ints) ivf_cand / ivf_cthr hold
    h->knn_sub_splits = 0;
    h->knn_wide_retries = 0;
    int ivf_pct = 0;   // tiles visited by the pruned main stage, per cent of a full scan (0: pruning not used)
    int ivf_passes = 0;   // ... and its form: 1 = single pass from sample-scan seeds, 2 = pass A + pass B
    // sample-seeded thresholds of the plain full scan (see run_stage)
    char *seed_img = nullptr;
    int *seed_map = nullptr;
    float *seed_thr = nullptr;
    float *side_s = nullptr;     // side values of the sample image's rows (bucket-centred sample scan)
    float *seed2_thr = nullptr;  // dense seeds of a single-pass pruned stage (every R2-th tile of the block's runs)
    int2 *seed2_runs = nullptr;
    char *Qs = nullptr;          // plain query image / norms for the sample scan of a bucket-centred single-pass stage
    float *qn_s = nullptr, *qlo_s = nullptr;
    bool last_seeded = false;   // the last stage ran from seeded thresholds: a failure may hold fewer than k candidates
    // level 0 state: the grid over the reference rows is built on first use
    void *grid_info = nullptr;
    int64_t *grid_start = nullptr;
    float *grid_xyz = nullptr;
    int *grid_idx = nullptr;

    g_dbg_ctr = nullptr;
    unsigned long long *d_ctr = nullptr;
    if (dbg & 4) {
        static unsigned long long *s_ctr = nullptr;
        if (!s_ctr) (void)hipMalloc((void **)&s_ctr, 256);
        (void)hipMemsetAsync(s_ctr, 0, 256, st);
        d_ctr = s_ctr;
        g_dbg_ctr = s_ctr;
    }

    auto side_full_compute = [&]() -> cm_status {
        if (side_full) return CM_OK;
        const int C = LS_CMAX;
        const size_t lds_c = sizeof(double) * (size_t)C * d;
        auto sidek = cen_side<T>;
        CM_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void *>(sidek), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c));
        hipLaunchKernelGGL(sidek, dim3((unsigned)((side_stride + 255) / 256)), dim3(256), lds_c, st, d_ref, n_ref, d, metric,
                           (const double *)d_mean, (const int *)rperm, (const double *)cen, C, (const float *)scale3, side_stride, side);
        CM_HIP(h, hipGetLastError());
        side_full = true;
        return CM_OK;
    };
    int img = -1;          // reference image currently in Rp: 0 full [hi|lo] tiles, 1 hi-only tiles, 2 direct (fp32 rows)
    int last_Kc = Kc_max, last_S = 1, last_variant = 0;   // (variant: which filter kernel the stage launched, see cm_knn_stats)
    // one stage: filter `level` over the query subset qmap[0..nsub) (nullptr = all), re-rank + certify, failures -> flag_out
    int seeds_from_level = 0;   // level of the stage whose re-rank wrote the current per-query seeds (0: none)
    int64_t n_unbounded = 0;   // of the flagged queries of the last stage: those without a finite bound (read_flags)
    int64_t n_over = 0;        // ... and those with more candidates than the re-rank's capacity
    bool last_wide = false, no_wide = false;   // the last stage split the reference beyond 16 ways / no more such stages in this search
    auto run_stage = [&](int level_in, const int *qmap, int64_t nsub, int *flag_out, bool rerank, bool marks, bool pilot) -> cm_status {
        int level = level_in;
        last_seeded = false;
        if (level == 0) {   // uniform-grid candidate search (d <= 3): same outputs as a filter kernel, one list per query
            if (!grid_info) CM_TRY(cm_grid_build<T>(h, d_ref, n_ref, d, k, &grid_info, &grid_start, &grid_xyz, &grid_idx));
            const int Kc0 = k + knn_slack(k);
            hipLaunchKernelGGL(zero_words, dim3(1), dim3(64), 0, st, words + 2, 2);  // flag count, overflow
            if (marks) cm_mark(h, 1);
            CM_TRY(cm_grid_query<T>(h, d_qry, nsub, d, qmap, grid_info, grid_start, grid_xyz, grid_idx, k, Kc0, words, cand, cthr));
            if (marks) cm_mark(h, 2);
            last_Kc = Kc0; last_S = 1;
            hipLaunchKernelGGL(knn_rerank<T>, dim3((unsigned)((nsub + 3) / 4)), dim3(256), rerank_lds_bytes(1, Kc0, d), st, d_ref, d_qry, nsub, d, cand, cthr,
                               1, Kc0, k, metric, form, qnorm, words, -(float)(d + 4) * 2.0f * 5.9604645e-8f, 0.f,
                               (const float *)nullptr, (const double *)nullptr, idx_offset, flags, d_out_dist, d_out_idx, flag_out,
                               (int *)(words + 2), bound, qmap, 0, (const float *)nullptr, 0, (const int *)nullptr);
            CM_HIP(h, hipGetLastError());
            if (marks) cm_mark(h, 3);
            return CM_OK;
        }
        const bool cenl = level == 5;   // bucket-centred hi*hi level: the filter arithmetic of level 1
        if (cenl) level = 1;
        KnnPlan2 p2;
        const bool w3 = wide && level >= 3;   // K = 128, three products: knn_cand_f16w (256-query blocks, full rows)
        CM_TRY(make_plan2(h, n_ref, nsub, d, k, level == 4 ? 2 : (w3 ? 3 : 1), level == 4 ? 3 : level, &p2));
        if (wide) p2.KSTEPS = cmk2::KSTEPS_W;
        if (cenl) {   // the centred level keeps a wider candidate list: its residual error is dominated by bucket quality
            const int kc2 = k + (k > 30 ? k : 30) + kc5_extra;
            if (kc2 <= Kc_max && kc2 + 2 * cmk2::CL <= 512) p2.Kc = kc2;   // (every candidate buffer is sized for Kc_max)
        }
        // ... and so does the three-product stage that takes over its failures (round 5): those are queries of tight clusters
        // whose k-th and K'-th neighbours lie closer together than the filter errors, and their seeds -- bounds plus error
        // margins -- admit dozens of rows per list.  With k + 8 candidates (the plain three-product slack) 3 % of the 98 k
        // queries the centred level left at sigma 0.2 failed again and took a direct-difference stage (45 ms in all); with the
        // centred level's list length none did (11.5 ms).
        if (!cenl && level == 3 && !w3 && !marks && !pilot && seeds_from_level == 5) {
            // (+ the long lists' 80 when the centred stage ran with them: what even 140 candidates could not certify there -- the
            //  hardest 3 % of the sigma 0.1 set -- fails a single list of 60 here for a fifth of the queries whenever the stage's
            //  lists happen to hold a query's whole cluster in one: measured on the pruned form of the stage)
            const int kc2 = k + (k > 30 ? k : 30) + (nsub >= cm_get_knobs().ivf_submin ? kc5_extra : 0);   // (small subsets certify with the short lists: 0 of 1 607 left at sigma 0.12)
            if (kc2 <= Kc_max && kc2 + 2 * cmk2::CL <= 512 && kc2 > p2.Kc) {
                p2.Kc = kc2;
                while (p2.S > 1 && p2.S * p2.Kc > RR_NCMAX) { p2.tps = (p2.n_tiles + p2.S - 2) / (p2.S - 1); p2.S = (p2.n_tiles + p2.tps - 1) / p2.tps; }
            }
        }
        // Pruning gate of a stage (allocations and the reference buckets' balls once per search; then the balls of the stage's
        // query blocks and the tiles of the (block, bucket) pairs whose balls overlap): clears ivf_try when nothing can be gained.
        // Called by the pruned branch below -- and, for a bucket-centred main stage, BEFORE the kernel is chosen: an unpruned
        // centred stage runs on the four-accumulator kernel, a pruned one on work lists.
        bool gate_done = false;
        int gate_ov_pct = 100;   // share of the (block, tile) pairs whose balls overlap, as the gate counted it
        auto ivf_gate = [&](int qblk, int BRt) -> cm_status {
            const int C = LS_CMAX;
            const int64_t nqb_max = (nq_pad + pack_slack) / (big_rows ? 256 : cmk2::QB) + 1;
            if (!ivf_rcen) {
                ivf_rcen = (double *)cm_ws_alloc(h, sizeof(double) * (size_t)C * IVF_CSTR);
                ivf_rrad = (double *)cm_ws_alloc(h, sizeof(double) * C);
                ivf_rcnt = (int *)cm_ws_alloc(h, sizeof(int) * C);
                ivf_bcen = (double *)cm_ws_alloc(h, sizeof(double) * (size_t)nqb_max * IVF_CSTR);
                ivf_brad = (double *)cm_ws_alloc(h, sizeof(double) * (size_t)nqb_max);
                ivf_bcnt = (int *)cm_ws_alloc(h, sizeof(int) * (size_t)nqb_max);
                ivf_jstar = (int *)cm_ws_alloc(h, sizeof(int) * (size_t)nqb_max);
                ivf_wla = (int4 *)cm_ws_alloc(h, sizeof(int4) * (size_t)nqb_max);
                ivf_wlb = (int4 *)cm_ws_alloc(h, sizeof(int4) * (size_t)nqb_max * IVF_SB);
                ivf_ctr = (int *)cm_ws_alloc(h, 64);
                ivf_runs = (int2 *)cm_ws_alloc(h, sizeof(int2) * (size_t)nqb_max * 136);   // (a block has at most C / 2 + 1 maximal runs)
                if (!ivf_rcen || !ivf_rrad || !ivf_rcnt || !ivf_bcen || !ivf_brad || !ivf_bcnt || !ivf_jstar || !ivf_wla || !ivf_wlb ||
                    !ivf_ctr || !ivf_runs)
                    CM_FAIL(h, CM_E_NOMEM, "k-NN: device workspace exhausted");
                ivf_ball<T>(st, d_ref, d, metric, (const double *)d_mean, (const int *)rperm, (const int *)r_bstart, n_ref, C, 32,
                            ivf_rcen, ivf_rrad, ivf_rcnt);
            }
            ivf_ball<T>(st, d_qry, d, metric, (const double *)d_mean, qmap, (const int *)nullptr, nsub, p2.nqb, 1, ivf_bcen, ivf_brad,
                        ivf_bcnt, qblk);
            hipLaunchKernelGGL(zero_words, dim3(1), dim3(64), 0, st, (unsigned *)ivf_ctr, 16);
            hipLaunchKernelGGL(ivf_overlap, dim3(p2.nqb), dim3(256), 0, st, (const double *)ivf_bcen, (const double *)ivf_brad,
                               (const double *)ivf_rcen, (const double *)ivf_rrad, (const int *)r_bstart, C, d, BRt,
                               reinterpret_cast<unsigned long long *>(ivf_ctr + 4));
            {
                int *ho = h->d_nflag_pinned + 10;
                CM_HIP(h, hipMemcpyAsync(ho, ivf_ctr + 4, 8, hipMemcpyDeviceToHost, st));
                CM_HIP(h, hipStreamSynchronize(st));
                unsigned long long ov;
                memcpy(&ov, ho, 8);
                const unsigned long long full0 = (unsigned long long)p2.nqb * (unsigned long long)p2.n_tiles;
                if (dbg & 16) fprintf(stderr, "[cellmap dbg] bucket pruning: balls overlap for %llu of %llu (block, tile) pairs\n", ov, full0);
                gate_ov_pct = (int)(ov * 100ull / std::max<unsigned long long>(full0, 1ull));
                // (a stage that can run in a single pass: the same conditions as `single` below)
                const bool single_ok_ = marks && level != 4 && !w3 && !(dbg & (128 | 512)) && (!(dbg & 16384) || p2.nqb >= 4 * (h->n_cu > 0 ? h->n_cu : 256) || (dbg & 256)) &&
                                        !forced && !(flags & CM_KNN_FORCE_FALLBACK) && !cm_get_knobs().no_seed;
                unsigned long long pmx_ = (unsigned long long)prune_max_pct(single_ok_, true);
                if (!marks && seeds_from_level >= 1) pmx_ = std::max<unsigned long long>(pmx_, (unsigned long long)prune_max_sub_pct());   // (seeded subset stage)
                if (ov * 100ull > full0 * pmx_) {   // (see the threshold after the plan)
                    ivf_try = false;
                    if (marks) { h->staged_ivf_skip = 15; h->staged_ivf_nref = n_ref; h->staged_ivf_nq = nq; }
                }
            }
            gate_done = true;
            return CM_OK;
        };
        if (cenl && marks && ivf_try && !pilot && rerank && nsub >= prune_min_nq() && std::min(IVF_SB, RR_NCMAX / p2.Kc - 1) >= 2 && !(dbg & 8192))
            CM_TRY(ivf_gate((int)(p2.nq_pad / std::max(p2.nqb, 1)), cmk2::BR));
        // (the padded set visits its share of the pairs twice over: kept below 30 % overlap -- ragged mixtures at 50 % measured
        //  15 - 24 % slower centred and pruned than unpruned on the three-product level, equal blobs at 2 - 20 % four times faster)
        if (cenl && marks && !pilot && rerank && (!ivf_try || gate_ov_pct > 30) && nq < 131072 && !forced && !(flags & CM_KNN_FORCE_FALLBACK) && !(dbg & 8192)) {
            cen_abandon = true;   // (nothing has been launched for the stage yet: the caller starts over on the three-product level)
            return CM_OK;
        }
        // Four accumulators per wave (knn_phase1_q4.h) for large full-set scans of the hi*hi level at K = 64: 1 024-query
        // blocks, so the padded query count is rounded up accordingly (every buffer carries that slack) and the 512-query
        // kernels of the same stage (seed scan) see an even number of blocks.
        // CELLMAP_Q4: 0 = never, 2 = whenever the kernel exists for the stage (tests), default = large sets only (a block is
        // 1 024 queries: below two blocks per CU the 512-query kernel fills the chip better).
        const int q4_env = cm_get_knobs().q4;
        // (round 5, late: the bucket-centred level too, once it is known that the stage will not be pruned -- pruned stages run on
        //  work lists, which this kernel does not take; the first call of a shape still decides that inside the stage)
        const bool cen_q4 = cenl && marks && !ivf_try && !(dbg & 7) && !(dbg & 8192);   // (CELLMAP_DBG=8192: off -- A/B)
        bool q4 = level == 1 && (!cenl || cen_q4) && !w3 && !wide && (p2.KSTEPS == 4 || p2.KSTEPS == 2) && !pilot && q4_env != 0 &&
                  (q4_env == 2 || nsub >= 32768);
        if (q4) {
            // Work partitioning.  A stage is ceil(blocks x S / resident workgroups) rounds of work items, an item = one query
            // block against 1/S of the reference.  An item's time does not fall like 1/S: about a quarter of a full scan is
            // per-query candidate handling, and a split still appends ~K'(1 + ln(R0 / (S K'))) rows per query (R0 ~ 12 K':
            // rank of the seed) -- measured: two half scans cost 1.15 x one full scan, four quarter scans 1.35 x.  Pick,
            // between the 512-query kernel with the plan's split count and the 1 024-query kernel (13 % faster per pair,
            // measured) with 1..8 splits, what this model says finishes first.  Forced mode (tests) keeps the plan's splits.
            const int res = h->n_cu > 0 ? h->n_cu : 256;
            const int64_t nb4 = (nsub + cmk2::QBQ - 1) / cmk2::QBQ;
            auto item = [](int S) { return 0.75 / S + 0.25 * (1.0 + log(std::max(1.0, 12.0 / S))) / (1.0 + log(12.0)); };
            auto rounds = [&](int64_t blocks, int S) { return (double)((blocks * S + res - 1) / res); };
            int S4 = p2.S;
            if (q4_env != 2) {
                double best = 1e30;
                for (int S = 1; S <= 8; ++S) {
                    if (S > 1 && (S * p2.Kc > RR_NCMAX || p2.n_tiles / S < 256 || (nb4 * cmk2::QBQ) * S > cand_rows)) continue;
                    const double r = rounds(nb4, S) * item(S);
                    if (r < best - 1e-9) { best = r; S4 = S; }
                }
                if (best >= rounds(p2.nqb, p2.S) * 0.565 * item(p2.S)) q4 = false;
            }
            if (q4) {
                p2.nq_pad = nb4 * cmk2::QBQ;
                p2.nqb = (int)(p2.nq_pad / cmk2::QB);
                p2.tps = (p2.n_tiles + S4 - 1) / S4;
                p2.S = (p2.n_tiles + p2.tps - 1) / p2.tps;
            }
        }
        // Round 5: the same rounds x item-cost choice for the hi*hi level of the K = 128 family (knn_cand_f16p<8>, one 512-query
        // workgroup per CU): its plan used to split the reference only when there were fewer query blocks than CUs, so a set of
        // 1.2 .. 3 rounds left up to 40 % of the last round idle.  Splitting is cheap at this width -- the per-query candidate
        // handling is a twentieth of an item, not a quarter (one rank's share of config 4 at 1.5 M / 3 M / 6 M reference rows
        // per query: 2.19 / 2.09 / 2.06 e-10 s per pair, profiles/r05_rank_shares.md) -- so the model takes 1 .. 4 splits when
        // that finishes sooner.  Large sets (config 4's 2 930 blocks: 11.45 rounds) keep S = 1.
        if (wide && level == 1 && !pilot && !cenl && !forced && p2.S == 1 && cm_get_knobs().q4 != 0) {
            const int res = h->n_cu > 0 ? h->n_cu : 256;
            auto item = [](int S) { return 0.95 / S + 0.05 * (1.0 + log(std::max(1.0, 12.0 / S))) / (1.0 + log(12.0)); };
            auto rounds = [&](int64_t blocks, int S) { return (double)((blocks * S + res - 1) / res); };
            double best = rounds(p2.nqb, 1) * item(1);
            int Sw = 1;
            for (int S = 2; S <= 4; ++S) {
                if (S * p2.Kc > RR_NCMAX || p2.n_tiles / S < 256 || (int64_t)p2.nq_pad * S > cand_rows) continue;
                const double r = rounds(p2.nqb, S) * item(S);
                if (r < best * 0.94) { best = r; Sw = S; }   // (6 % margin: the split's lists cost the re-rank ~2 ms per 200 k queries too)
            }
            if (Sw > 1) {
                p2.tps = (p2.n_tiles + Sw - 1) / Sw;
                p2.S = (p2.n_tiles + p2.tps - 1) / p2.tps;
                const int64_t nw = (int64_t)p2.nqb * p2.S;
                p2.grid = (int)(nw < res ? nw : res);
            }
        }
        // Round 4: under CELLMAP_R16=1 the 128-queries-per-wave stage at K = 64 runs on the 16x16x32 MFMA shape
        // (knn_phase1_r16.h: built, exact, measured level with the 32x32x16 kernel -- profiles/r04_mfma_shape_probe.md -- and
        // therefore not the default).  Its chains are 2 instructions of 32 products each: see c_eps below.
#ifdef CM_ABLATE
        const bool r16 = q4 && p2.KSTEPS == 4 && cm_get_knobs().r16 != 0;
#else
        const bool r16 = false;   // (the 16x16x32 re-tile is built by `make ABLATE=1` only)
#endif
        const int want = level == 4 ? 2 : (level < 3 ? 1 : 0);
        const int64_t nref_pad = (int64_t)p2.n_tiles * (level == 4 ? cmk2::lowd_rows(dl) : cmk2::BR);
        if (img != want) {
            if (want == 2)
                hipLaunchKernelGGL(prep_rows_lowd<T>, dim3((unsigned)((nref_pad + bs - 1) / bs)), dim3(bs), 0, st, d_ref, n_ref, d,
                                   nref_pad, 1, metric, reinterpret_cast<float *>(Rp), dl, (const int *)rperm);
            else
                hipLaunchKernelGGL(prep_rows_f16<T>, dim3((unsigned)((nref_pad + bs - 1) / bs)), dim3(bs), prep_lds, st, d_ref, n_ref, d,
                                   p2.KSTEPS, nref_pad, 1, metric, scale3, (const double *)d_mean, Rp, (float *)nullptr, words,
                                   wide ? 1 : 0, want, (const int *)rperm, (float *)nullptr, (const double *)nullptr,
                                   (const int *)nullptr, (float *)nullptr, (float *)nullptr, (int)prep_lds);
            img = want;
        }
        if (level == 4) {
            hipLaunchKernelGGL(prep_rows_lowd<T>, dim3((unsigned)((p2.nq_pad + bs - 1) / bs)), dim3(bs), 0, st, d_qry, nsub, d,
                               p2.nq_pad, 0, metric, reinterpret_cast<float *>(Qp), dl, qmap);
        } else {
            hipLaunchKernelGGL(prep_rows_f16<T>, dim3((unsigned)((p2.nq_pad + bs - 1) / bs)), dim3(bs), prep_lds, st, d_qry, nsub, d,
                               p2.KSTEPS, p2.nq_pad, 0, metric, scale3, (const double *)d_mean, Qp, qnorm, (unsigned *)nullptr,
                               wide ? 1 : 0, 0, qmap, qlo, cenl ? (const double *)cen : (const double *)nullptr,
                               cenl ? (const int *)blk_bucket : (const int *)nullptr, cenl ? qnc : (float *)nullptr,
                               cenl ? qcen : (float *)nullptr, (int)prep_lds);
            if (!w3) {   // (knn_cand_f16w keeps the plan of make_plan2: 2-deep ring of full tiles)
                p2.tileb = (size_t)cmk2::img_tile_bytes(p2.KSTEPS, want == 1);
                p2.lds = cmk2::ring_depth(want == 1, p2.KSTEPS) * (p2.tileb + (cenl ? (q4 ? 2048 : 1024) : 0)) + 64;   // (centred: + the side block(s) of a tile)
            }
        }
        // Subset stages start from what their queries already know (see the unpruned launch below).  Seeds of the plain levels
        // 1..3 serve a level at least as precise as the one that left them; seeds of the bucket-centred level (round 5) serve
        // the plain MFMA levels with the target level's own error added (seed_gather).
        const bool sub_seed_ok = !marks && !pilot && rerank && qmap && !cenl && level != 4 && !w3 && !forced &&
                                 ((seeds_from_level >= 1 && seeds_from_level <= 3 && seeds_from_level <= level) || seeds_from_level == 5) &&
                                 !(flags & CM_KNN_FORCE_FALLBACK) && !cm_get_knobs().no_subseed && !cm_get_knobs().no_seed;
        // Small seeded subsets (round 5): a stage of one or a few query blocks used to split the reference 13 - 16 ways (what
        // S x K' <= RR_NCMAX allows) -- 16 work items on 256 compute units, 8 ms for the 379 queries the centred level left
        // at the 40-cluster config-2 shape (sigma 0.3).  Every query of such a subset starts from a FINITE seed (counted:
        // n_unbounded == 0), so its lists hold a few rows beyond k in all: the reference is split as many ways as fills the
        // chip and the re-rank compacts the S mostly empty lists into RR_NCMAX slots (as after a pruned stage; a query
        // with more is left to the next level).
        // (the direct-difference level too: its seed is the query's bound plus that filter's relative error, float32 inputs --
        //  see seed_a below; 852 queries of the 40-cluster d = 128 shape took 38 ms on 16 work items)
        const bool sub_seed4_ok = !marks && !pilot && rerank && qmap && level == 4 && !forced && sizeof(T) == 4 && seeds_from_level >= 1 &&
                                  seeds_from_level != 4 && !(flags & CM_KNN_FORCE_FALLBACK) && !cm_get_knobs().no_subseed && !cm_get_knobs().no_seed;
        bool wide_split = false;
        if ((sub_seed_ok || sub_seed4_ok) && n_unbounded == 0 && !q4 && !no_wide && !(dbg & 2048)) {   // (CELLMAP_DBG=2048: off -- A/B)
            const int res = h->n_cu > 0 ? h->n_cu : 256;
            const int resident = res * (8 / cmk2::NW);
            int Sw = resident / std::max(p2.nqb, 1);
            Sw = std::min(Sw, std::min(256, p2.n_tiles / 16));
            while (Sw > p2.S && (int64_t)p2.nq_pad * Sw > cand_rows) --Sw;
            if (Sw >= 2 * p2.S) {
                p2.tps = (p2.n_tiles + Sw - 1) / Sw;
                p2.S = (p2.n_tiles + p2.tps - 1) / p2.tps;
                const int64_t nw = (int64_t)p2.nqb * p2.S;
                p2.grid = (int)(nw < resident ? nw : resident);
                wide_split = true;
            }
        }
        last_wide = wide_split;
        if ((int64_t)p2.nq_pad * p2.S > cand_rows) CM_FAIL(h, CM_E_INTERNAL, "k-NN: candidate buffer too small for a stage");
        if (!marks && !pilot && p2.S > h->knn_sub_splits) h->knn_sub_splits = p2.S;
        cmk2::P1Args a;
        a.Q = reinterpret_cast<const _Float16 *>(Qp);
        a.R = Rp;
        a.n_tiles = p2.n_tiles; a.tiles_per_split = p2.tps; a.S = p2.S; a.nqb = p2.nqb; a.Kc = p2.Kc;
        a.cand = cand; a.cthr = cthr; a.kept = kept; a.lbuf = lbuf;
        a.nprod = level == 4 ? 3 : level;
        a.dbg = dbg & ~(16 | 64 | 128 | 256 | 512 | 1024 | 2048 | 4096 | 8192 | 16384);
        a.ctr = d_ctr;   // (bits 16, 64, ... are host-side switches: the production kernel bodies stay selected)
        if (cenl) { a.side = side; a.blk_bucket = blk_bucket; a.side_stride = side_stride; }
        a.perm_mul = 1;
        if (level != 4 && !w3 && p2.n_tiles > 8 && !cm_get_knobs().no_perm) {   // knn_cand_f16: scattered tile order
            unsigned long long pm = (unsigned long long)((double)p2.n_tiles * 0.6180339887498949) | 1ull;
            auto gcd = [](unsigned long long x, unsigned long long y) { while (y) { unsigned long long t = x % y; x = y; y = t; } return x; };
            while (gcd(pm, (unsigned long long)p2.n_tiles) != 1ull) pm += 2;
            a.perm_mul = (unsigned)pm;
        }
        last_Kc = p2.Kc; last_S = p2.S;
        // error-bound constants of this level (knn_rerank; the pruning plan bounds the pass-A thresholds with them too)
        float c_eps, c_split = 0.f;
        if (level == 4) {
            c_eps = -(float)(d + 4) * 2.0f * 5.9604645e-8f;   // relative bound of the fp32 direct sum (knn_rerank)
        } else {
            // dropped split products: |ql.y'| <= 2^-11 |q||y'| each (fp16 round-to-nearest), y' = -2y
            c_split = (level == 1 ? 2.0f : (level == 2 ? 1.0f : 0.f)) * 4.8828125e-4f * 1.01f;
            // Accumulation + representation (8.4e-7).  The plain levels 1 and 2 use the ANALYTIC worst case of the chain -- n = 16 KSTEPS
            // products per split product summed into the accumulator (+ the |y|^2 operand of K = 128), every addition rounding by
            // at most one unit in the last place of a partial sum that never exceeds the sum of the magnitudes (truncation
            // assumed, whatever the order inside an instruction): (n + 1) 2^-23 sum|terms|.  It holds for either MFMA shape (16
            // or 32 products per instruction: the same number of additions).  The dropped split products (c_split / the lo
            // norms) are 60 - 270 x larger at these levels, so the proven constant costs nothing: config 2 and the sigma = 2
            // mixture certify every query at level 1 in the same time under either constant (profiles/r05_accumulation_bound.md),
            // and the exactness argument of the level every benchmark configuration's headline certifies at contains no
            // measured constant.
            // Level 3 and the bucket-centred level keep the device-validated model of the f16 MFMA chain, doubled
            // (tests/test_gpu_parity.py::test_mfma_f16_accumulation_bound checks it on the device the tests run on): there the
            // accumulation term IS the error bound -- three exact products at level 3; at the centred level the dropped products
            // scale with the bucket radius, not with |q'| -- and the analytic constant (2.1 x / 4 x the model) sends the queries of
            // tight clusters on to the next level: measured on the sigma = 0.1 mixture, 750 k x 1.5 M, the centred level
            // certified 69 % under the model and the pilot skipped it altogether under the analytic constant (97 -> 185 ms,
            // same file).  DESIGN.md section 4.
            if (level <= 2 && !cenl && !(dbg & 64))   // (CELLMAP_DBG=64: the measured model at every level -- the A/B of profiles/r05_accumulation_bound.md)
                c_eps = (float)(16 * p2.KSTEPS * level + (wide ? 1 : 0) + 1) * 1.1920929e-7f * 1.001f + 8.4e-7f;   // (+1: the |y|^2 operand of K = 128)
            else {
                c_eps = 2.0f * (float)(16 + 2 * (p2.KSTEPS * level + (wide ? 1 : 0))) * 5.9604645e-8f + 8.4e-7f;   // K = 128: + the |y|^2 add
                if (r16) c_eps = 2.0f * (float)(32 + 2 * (p2.KSTEPS / 2)) * 5.9604645e-8f + 8.4e-7f;   // (not reachable: r16 is a level-1 kernel)
            }
        }
        hipLaunchKernelGGL(zero_words, dim3(1), dim3(64), 0, st, words + 2, 2);  // flag count, overflow
        hipLaunchKernelGGL(zero_words, dim3(1), dim3(64), 0, st, words + 22, 3);  // flagged queries without a finite bound; over capacity; fewer than k candidates
        if (marks) cm_mark(h, 1);   // everything before the main stage's filter launch (pilot included) counts as "prep"
        const int *rr_cand = cand;
        const float *rr_cthr = cthr;
        int rr_S = p2.S, rr_ncap = wide_split ? RR_NCMAX : 0;
        bool pruned = false, sample_done = false;
        const bool full_set = qmap && (qmap == (const int *)qperm || qmap == (const int *)qpack || qmap == (const int *)flag_s);
        auto launch = [&](const cmk2::P1Args &x) -> cm_status {
            const int op = lp_metric ? (metric == CM_METRIC_MANHATTAN ? 1 : (metric == CM_METRIC_CHEBYSHEV ? 2 : (metric == CM_METRIC_CANBERRA ? 3 : 4))) : 0;
            if (!x.seed_len) last_variant = level == 4 ? 5 : (w3 ? 4 : ((q4 && !x.wlist) ? (r16 ? 3 : 2) : 1));
            if (level == 4 && dl > 64) return cm_p1direct_wide_launch(h, dl, op, p2.grid, p2.lds, x);
            if (level == 4 && lp_metric) return cm_p1lowd_metric_launch(h, dl, op, p2.grid, p2.lds, x);
            if (level == 4) return cm_p1lowd_launch(h, dl, p2.grid, p2.lds, x);
            if (w3) return cm_p1f16w_launch(h, p2.grid, p2.lds, x);
            if (q4 && !x.wlist && !x.seed_len) {
                cmk2::P1Args y = x;
                y.nqb = (int)(p2.nq_pad / cmk2::QBQ);
                const int res = h->n_cu > 0 ? h->n_cu : 256;
                const int64_t nw = (int64_t)y.nqb * y.S;
#ifdef CM_ABLATE
                if (r16) return cm_p1f16r_launch(h, p2.KSTEPS, (int)(nw < res ? nw : res), p2.lds, y);
#endif
                return cm_p1f16q_launch(h, p2.KSTEPS, (int)(nw < res ? nw : res), p2.lds, y);
            }
            return cm_p1f16_launch(h, p2.KSTEPS, p2.grid, p2.lds, x);
        };
        // Sample scan (threshold seeding): a value with at least m = 8 rows of a 1/R row sample at or below it, per query
        // (knn_cand_f16 in seeding mode: two lanes per query, each keeping its 4 smallest sub-tile minima in registers -- no
        // lists, no merges).  The number of sample rows among the query's k nearest is ~Binomial(k, 1/R) ~ Poisson(k/R); the
        // seed is too tight (fewer than k rows below it) only if that number reaches m.  R = the smallest power of two from 16
        // on with P(Poisson(k/R) >= m) < 1e-7 (k = 30: R = 64, seed at rank ~ 8*64 of the full set): a retry stage is then a
        // rarity, and the result never depends on it (certificate + escalation).  Qimg: PLAIN (not bucket-centred) query rows.
        auto sample_scan = [&](const cmk2::P1Args &base, const _Float16 *Qimg, int r_mul = 1, bool cen_mode = false) -> cm_status {
            const int m = 8, L = 4;
            int R = 16;
            for (; R < 4096; R *= 2) {
                const double x = (double)k / R;
                double pterm = exp(-x), cum = pterm;
                for (int j = 1; j < m; ++j) { pterm *= x / j; cum += pterm; }
                if (1.0 - cum < 1e-7) break;
            }
            // (r_mul: a single-pass pruned stage with dense seeds needs this scan for the plan's radii only -- a 4 x coarser sample,
            //  rank ~2 048 instead of ~512, widens the radii by 3 % at d = 50 and costs a quarter)
            while (r_mul > 1 && R < 4096 && (n_ref + 2 * R - 1) / (2 * R) >= 64 * cmk2::BR) { R *= 2; r_mul /= 2; }
            const int64_t n_s = (n_ref + R - 1) / R;
            const int n_s_tiles = (int)((n_s + cmk2::BR - 1) / cmk2::BR);
            const size_t tb1 = (size_t)cmk2::img_tile_bytes(p2.KSTEPS, true);
            if (!seed_img) {
                seed_img = (char *)cm_ws_alloc(h, (size_t)n_s_tiles * tb1);
                seed_map = (int *)cm_ws_alloc(h, sizeof(int) * (size_t)n_s_tiles * cmk2::BR);
                seed_thr = (float *)cm_ws_alloc(h, sizeof(float) * (size_t)(nq_pad + pack_slack + 1024));
                if (!seed_img || !seed_map || !seed_thr) CM_FAIL(h, CM_E_NOMEM, "k-NN: device workspace exhausted");
            }
            hipLaunchKernelGGL(sample_map, dim3((unsigned)(((int64_t)n_s_tiles * cmk2::BR + 255) / 256)), dim3(256), 0, st, (const int *)rperm, n_s, R,
                               (int64_t)n_s_tiles * cmk2::BR, seed_map);
            hipLaunchKernelGGL(prep_rows_f16<T>, dim3((unsigned)(((int64_t)n_s_tiles * cmk2::BR + bs - 1) / bs)), dim3(bs), prep_lds, st, d_ref, (int64_t)n_s_tiles * cmk2::BR, d,
                               p2.KSTEPS, (int64_t)n_s_tiles * cmk2::BR, 1, metric, scale3, (const double *)d_mean, seed_img, (float *)nullptr,
                               (unsigned *)nullptr, wide ? 1 : 0, 1, (const int *)seed_map, (float *)nullptr, (const double *)nullptr,
                               (const int *)nullptr, (float *)nullptr, (float *)nullptr, (int)prep_lds);
            cmk2::P1Args sa = base;
            sa.Q = Qimg; sa.R = seed_img; sa.n_tiles = n_s_tiles; sa.tiles_per_split = n_s_tiles; sa.S = 1;
            sa.cand = nullptr; sa.cthr = seed_thr; sa.nprod = 1; sa.perm_mul = 1;
            sa.side = nullptr; sa.blk_bucket = nullptr;   // (a bucket-centred stage seeds from the plain hi*hi values)
            sa.wlist = nullptr; sa.n_wlist = 0; sa.wruns = nullptr; sa.wctr = nullptr; sa.thr0 = nullptr; sa.thr0_slot0 = 0; sa.no_sched = 0;
            sa.seed_len = L;
            if (d_ctr) sa.ctr = d_ctr + 16;   // (diagnostics: the sample scan counts separately)
            const size_t lds_s = (size_t)cmk2::ring_depth(true, p2.KSTEPS) * tb1 + 64;
            if (cen_mode) {
                // bucket-centred sample scan (Qimg = the CENTRED query image): the sample rows' side values are gathered from the
                // full side array, the seeds are values of the centred filter itself -- nothing to widen
                const int64_t n_img = (int64_t)n_s_tiles * cmk2::BR;
                if (!side_s) {
                    side_s = (float *)cm_ws_alloc(h, sizeof(float) * (size_t)LS_CMAX * (size_t)n_img);
                    if (!side_s) CM_FAIL(h, CM_E_NOMEM, "k-NN: device workspace exhausted");
                }
                hipLaunchKernelGGL(side_sample_gather, dim3((unsigned)((n_img + 255) / 256), LS_CMAX), dim3(256), 0, st, (const float *)side,
                                   side_stride, n_s, R, n_img, side_s);
                sa.side = side_s; sa.side_stride = n_img; sa.blk_bucket = blk_bucket;
                CM_TRY(cm_p1f16_seed10_launch(h, p2.KSTEPS, p2.grid, (size_t)cmk2::ring_depth(true, p2.KSTEPS) * (tb1 + 1024) + 64, sa));
                return CM_OK;
            }
            CM_TRY(cm_p1f16_launch(h, p2.KSTEPS, p2.grid, lds_s, sa));
            return CM_OK;
        };
        // candidate lists per block: the re-rank ranks at most RR_NCMAX candidates per query, and on the last level a query
        // must never be dropped for holding too many (its only way out would be the exact scan with a loose bound)
        const int ivf_sb = std::min(IVF_SB, RR_NCMAX / p2.Kc - 1);
        if (ivf_try && full_set && !pilot && rerank && ivf_sb >= 2 && nsub >= (marks ? prune_min_nq() : (int64_t)cm_get_knobs().ivf_submin)) {
            const int qblk = (int)(p2.nq_pad / std::max(p2.nqb, 1));   // queries per work item of this stage's filter kernel (512; 256: knn_cand_f16w, direct rows beyond 64 floats)
            const int BRt = level == 4 ? cmk2::lowd_rows(dl) : cmk2::BR;   // reference rows per tile of this level's image

            // ---- bucket pruning: pass A over every block's nearest bucket, plan, pass B over what cannot be ruled out
            const int C = LS_CMAX;
            const int64_t nqb_max = (nq_pad + pack_slack) / (big_rows ? 256 : cmk2::QB) + 1;
            // Chained pass B (round 5): with four or more blocks per compute unit the blocks balance the chip by themselves, so a
            // block's runs are scanned by ONE work item into ONE candidate list (knn_cand_f16, P1Args::wruns): 2 lists per query
            // instead of 14 -- a seventh of the list memory (5.7 -> 0.8 GB at the config-2 shape) and of its 0xff fill, and the
            // re-rank gathers 2 K' candidates instead of up to 6 K' (a cluster split over several buckets used to fill a list per
            // bucket: 17 ms of re-rank at 1.5 M clustered queries).  Few blocks, the direct level and the K = 128 three-product
            // kernel keep one list per run (runs halved for load balance).
            const int res_ = h->n_cu > 0 ? h->n_cu : 256;
            // (round 5, late: the MAIN stage chains -- and takes the single pass below -- whatever its block count: measured against pass A + B
            //  at 375 k / 187.5 k / 100 k / 66 k queries x 1.5 M rows, 40 clusters: 26.0 / 18.3 / 52.6 / 51.0 -> 16.5 / 11.5 / 45.4 / 45.0 ms;
            //  CELLMAP_DBG=16384: the four-blocks-per-CU rule for the main stage too -- A/B)
            const bool single_able = marks && !forced && !(flags & CM_KNN_FORCE_FALLBACK) && !cm_get_knobs().no_seed && !(dbg & (512 | 16384));
            const bool chain = level != 4 && !w3 && (p2.nqb >= 4 * res_ || single_able || (dbg & 256)) && !(dbg & 128);   // (CELLMAP_DBG=128: one list per run everywhere -- A/B; 256: chained whatever the block count -- tests)
            const int use_S = chain ? 2 : ivf_S;
            {
                const size_t need = (size_t)(p2.nq_pad + 1024) * (size_t)use_S;
                if (need > ivf_list_cap) {
                    ivf_cand = (int *)cm_ws_alloc(h, sizeof(int) * need * Kc_max);
                    ivf_cthr = (float *)cm_ws_alloc(h, sizeof(float) * need);
                    if (!ivf_cand || !ivf_cthr) CM_FAIL(h, CM_E_NOMEM, "k-NN: device workspace exhausted");
                    ivf_list_cap = need;
                }
            }
            if (!gate_done) CM_TRY(ivf_gate(qblk, BRt));   // (a centred stage asked before it chose its kernel)
            auto side_items = [&](const int4 *wl, int n_items, const int2 *runs = nullptr) {   // centred level: side values of the pairs a list visits
                if (!cenl || side_full || n_items < 1) return;
                // (side_claimed: two bitmaps, one bit per (bucket, tile) pair each, allocated and zeroed with the side values in setup_centred)
                const int wpb = (pf.n_tiles + 31) / 32 + 1;
                hipLaunchKernelGGL(cen_side_mark, dim3((unsigned)n_items, 1), dim3(128), 0, st, (const int *)blk_bucket, wl, n_items,
                                   (const unsigned *)side_claimed, side_claimed + (size_t)LS_CMAX * wpb, wpb, runs);
                hipLaunchKernelGGL(cen_side_compute<T>, dim3((unsigned)wpb, LS_CMAX), dim3(128), 0, st, d_ref, n_ref, d, metric,
                                   (const double *)d_mean, (const int *)rperm, (const double *)cen, (const float *)scale3, side_stride, side,
                                   side_claimed, side_claimed + (size_t)LS_CMAX * wpb, wpb);
            };
            // Single-pass pruned stage (round 5): the radii of the plan come from a SAMPLE SCAN instead of pass A.  Pass A ran
            // every block over its nearest bucket from +inf thresholds -- the start-up churn of a scan plus a list finalisation
            // per block, as long as pass B itself (2 x 22.5 ms at the 40-cluster config-2 shape) -- only to learn a radius.
            // The sample scan (every R-th row of the sorted reference, 2 ms) gives every query a value with >= 8 sample rows at
            // or below it: in input units plus the hi*hi error bound that is a radius D_q with >= 8 reference rows inside, and
            // the ONE chained pass that follows starts every query's threshold at its seed.
            // Exactness: D_q = (the threshold the query STARTS from, in input units) + |q'|^2 + this level's filter error bound;
            // a bucket is skipped for a block only if every row in it is farther than max D_q (+ margins) from every query of
            // the block, i.e. its value on this level's filter exceeds the query's starting threshold -- such a row could not
            // have entered a list that starts there, so the lists are those of an unpruned seeded scan, and a seed that was
            // too tight ends, as there, with fewer than k candidates and the query moves on.  Where the seed came from does
            // not enter the argument.
            const bool single = chain && marks && ivf_try && !forced && !(flags & CM_KNN_FORCE_FALLBACK) && !cm_get_knobs().no_seed &&
                                !(dbg & 512);   // (CELLMAP_DBG=512: the two-pass form -- A/B, tests)
            if (single) {
                const _Float16 *Qplain = a.Q;
                const float *qn_p = qnorm, *qlo_p = qlo;
                if (cenl) {   // the centred stage's query image holds q' - c: the sample scan needs the plain rows
                    if (!Qs) {
                        Qs = (char *)cm_ws_alloc(h, (size_t)(nq_pad + pack_slack) * 256);
                        qn_s = (float *)cm_ws_alloc(h, sizeof(float) * (nq_pad + pack_slack));
                        qlo_s = (float *)cm_ws_alloc(h, sizeof(float) * (nq_pad + pack_slack));
                        if (!Qs || !qn_s || !qlo_s) CM_FAIL(h, CM_E_NOMEM, "k-NN: device workspace exhausted");
                    }
                    hipLaunchKernelGGL(prep_rows_f16<T>, dim3((unsigned)((p2.nq_pad + bs - 1) / bs)), dim3(bs), prep_lds, st, d_qry, nsub, d,
                                       p2.KSTEPS, p2.nq_pad, 0, metric, scale3, (const double *)d_mean, Qs, qn_s, (unsigned *)nullptr,
                                       wide ? 1 : 0, 0, qmap, qlo_s, (const double *)nullptr, (const int *)nullptr, (float *)nullptr,
                                       (float *)nullptr, (int)prep_lds);
                    Qplain = reinterpret_cast<const _Float16 *>(Qs); qn_p = qn_s; qlo_p = qlo_s;
                }
                const bool dense_ok = (level == 1 || cenl) && !wide && !(dbg & 1024);   // (the dense seeding scan below exists for this stage)
                CM_TRY(sample_scan(a, Qplain, dense_ok ? 4 : 1));
                sample_done = !cenl && !dense_ok;   // (the coarse sample's seeds are good enough for radii, not for an unpruned scan)
                hipLaunchKernelGGL(zero_words, dim3(1), dim3(64), 0, st, (unsigned *)ivf_ctr, 16);
                const float c_eps1 = (float)(16 * p2.KSTEPS + (wide ? 1 : 0) + 1) * 1.1920929e-7f * 1.001f + 8.4e-7f;   // analytic hi*hi constant (below)
                const bool widen = cenl || level != 1;   // a more precise filter than the seeds': see ivf_plan_b
                hipLaunchKernelGGL(ivf_plan_b, dim3(p2.nqb), dim3(256), 0, st, (const double *)ivf_bcen, (const double *)ivf_brad,
                                   (const double *)ivf_rcen, (const double *)ivf_rrad, (const int *)r_bstart, C, d, qmap, nsub,
                                   (const float *)seed_thr, 1, (const float *)qnorm, (const float *)qlo,
                                   cenl ? (const float *)qnc : (const float *)nullptr, cenl ? (const float *)qcen : (const float *)nullptr,
                                   (const unsigned *)words, (const float *)scale3, c_eps, level, (const int4 *)ivf_wla,
                                   (const int *)ivf_jstar, ivf_wlb, ivf_ctr, BRt, sizeof(T) == 8 ? 1 : 0, d, 256, ivf_sb, qblk, ivf_runs, 1,
                                   qn_p, qlo_p, c_eps1, widen ? seed_thr : (float *)nullptr, cm_get_knobs().seed_widen);
                CM_HIP(h, hipGetLastError());
                int *hc = h->d_nflag_pinned + 4;
                CM_HIP(h, hipMemcpyAsync(hc, ivf_ctr, 4 * sizeof(int), hipMemcpyDeviceToHost, st));
                CM_HIP(h, hipStreamSynchronize(st));
                unsigned long long tiles;
                memcpy(&tiles, hc + 2, 8);
                const unsigned long long full = (unsigned long long)p2.nqb * (unsigned long long)p2.n_tiles;
                if (dbg & 16)
                    fprintf(stderr, "[cellmap dbg] bucket pruning (single pass): %llu of %llu (block, tile) pairs to visit, %d runs for %d of %d blocks\n",
                            tiles, full, hc[1], hc[0], p2.nqb);
                if (tiles * 100ull <= full * (unsigned long long)prune_max_pct(true)) {
                    // (a block without a run -- seeds too tight for any bucket -- writes nothing: empty lists for everybody first)
                    CM_HIP(h, hipMemsetAsync(cand, 0xff, sizeof(int) * (size_t)p2.nq_pad * p2.Kc, st));
                    hipLaunchKernelGGL(fill_f32, dim3((unsigned)((p2.nq_pad + 255) / 256)), dim3(256), 0, st, cthr, (int64_t)p2.nq_pad, INFINITY);
                    if (hc[0] > 0) {
                        // Dense seeds.  The sample-scan seed sits at rank ~8 R (512 at k = 30) of the full set; the pass over a
                        // block's own cluster then appends that many rows per query before its lists tighten -- at the
                        // 40-cluster config-2 shape 328 appends and 7 merges per query, 70 % of the pass (in-kernel counters),
                        // for 2.4 % of the pairs.  So the items' runs are first walked in seeding mode over every R2-th TILE
                        // (10-entry lists: >= 20 rows of a 1/R2 sample at or below the seed; R2 = the smallest power of two
                        // with P(Poisson(k / R2) >= 20) < 1e-7 -- 8 at k = 30: rank ~160), no lists, no merges, 1/R2 of the
                        // MFMA work of the pass; the pass starts from the smaller of the two seeds.  (The plan above used the
                        // looser one, so nothing it ruled out can enter a list that starts lower still.)
                        const bool dense = dense_ok;   // (hi-only tile image; CELLMAP_DBG=1024: off -- A/B)
                        if (dense) {
                            int R2 = 2;
                            for (; R2 < 64; R2 *= 2) {
                                const double x = (double)k / R2;
                                double pterm = exp(-x), cum = pterm;
                                for (int j = 1; j < 20; ++j) { pterm *= x / j; cum += pterm; }
                                if (1.0 - cum < 1e-7) break;
                            }
                            if (!seed2_thr) {
                                seed2_thr = (float *)cm_ws_alloc(h, sizeof(float) * (size_t)(nq_pad + pack_slack + 1024));
                                seed2_runs = (int2 *)cm_ws_alloc(h, sizeof(int2) * (size_t)nqb_max * 136);
                                if (!seed2_thr || !seed2_runs) CM_FAIL(h, CM_E_NOMEM, "k-NN: device workspace exhausted");
                            }
                            hipLaunchKernelGGL(fill_f32, dim3((unsigned)((p2.nq_pad + 255) / 256)), dim3(256), 0, st, seed2_thr, (int64_t)p2.nq_pad, INFINITY);
                            hipLaunchKernelGGL(ivf_seed_runs, dim3((unsigned)((hc[1] + 255) / 256)), dim3(256), 0, st, (const int2 *)ivf_runs, hc[1], R2, seed2_runs);
                            // (the scan runs THIS stage's filter -- for a bucket-centred stage the centred one, on the side values
                            //  of the runs' tiles -- so its seeds are values the pass itself will compute: no widening, and none of
                            //  the plain filter's noise, which on tight clusters (sigma <= 0.2 x the spread / 25) exceeds the gaps
                            //  between the k-th and the K'-th neighbour: seeds from noisy values landed between the two for most
                            //  queries, the lists ended short of K' with the seed as their threshold, and 93 % failed the certificate)
                            side_items(ivf_wlb, hc[0], (const int2 *)ivf_runs);
                            cmk2::P1Args sd = a;
                            sd.S = 1; sd.cand = nullptr; sd.cthr = seed2_thr; sd.nprod = 1;
                            sd.perm_mul = (unsigned)R2;
                            sd.wlist = ivf_wlb; sd.n_wlist = hc[0]; sd.wruns = seed2_runs; sd.wctr = ivf_ctr + 10;
                            sd.thr0 = nullptr; sd.thr0_slot0 = 0; sd.no_sched = 0; sd.seed_len = 10;
                            if (d_ctr) sd.ctr = d_ctr + 16;
                            CM_TRY(cm_p1f16_seed10_launch(h, p2.KSTEPS, p2.grid, p2.lds, sd));
                            hipLaunchKernelGGL(seed_combine, dim3((unsigned)((p2.nq_pad + 255) / 256)), dim3(256), 0, st, seed_thr, (const float *)seed2_thr,
                                               (int64_t)nsub, qn_p, qlo_p, c_eps1, 0.f, (const unsigned *)words, (const float *)scale3);
                        }
                        cmk2::P1Args b = a;
                        b.S = 1; b.perm_mul = 1;
                        b.wlist = ivf_wlb; b.n_wlist = hc[0]; b.wruns = ivf_runs; b.wctr = ivf_ctr + 9;
                        b.thr0 = seed_thr; b.thr0_slot0 = 0; b.no_sched = 1;
                        if (!dense) side_items(ivf_wlb, hc[0], (const int2 *)ivf_runs);
                        CM_TRY(launch(b));
                    }
                    pruned = true;
                    rr_S = 1;
                    last_seeded = !cenl;   // (failures of a plain level retry it unseeded, as after a seeded full scan; the centred level has no subset form)
                    if (marks) { ivf_pct = (int)std::max<unsigned long long>(1ull, tiles * 100ull / std::max<unsigned long long>(full, 1ull)); ivf_passes = 1; }
                } else if (tiles * 3ull > full) {   // hopeless (always, now that the single pass accepts up to 70 %; CELLMAP_PRUNE_MAX below 33: pass A + B gets its try)
                    ivf_try = false;
                    if (marks) { h->staged_ivf_skip = 15; h->staged_ivf_nref = n_ref; h->staged_ivf_nq = nq; }
                }
            }
            if (ivf_try && !pruned) {
            hipLaunchKernelGGL(ivf_plan_a, dim3(p2.nqb), dim3(256), 0, st, (const double *)ivf_bcen, (const double *)ivf_rcen,
                               (const int *)r_bstart, C, d, 2 * p2.Kc, p2.n_tiles, BRt, ivf_wla, ivf_jstar);
            const int64_t n_ent = (int64_t)p2.nq_pad * use_S;
            CM_HIP(h, hipMemsetAsync(ivf_cand, 0xff, sizeof(int) * (size_t)n_ent * p2.Kc, st));
            hipLaunchKernelGGL(fill_f32, dim3((unsigned)((n_ent + 255) / 256)), dim3(256), 0, st, ivf_cthr, n_ent, INFINITY);
            hipLaunchKernelGGL(zero_words, dim3(1), dim3(64), 0, st, (unsigned *)ivf_ctr, 16);
            cmk2::P1Args b = a;
            b.cand = ivf_cand; b.cthr = ivf_cthr; b.S = use_S; b.perm_mul = 1;
            b.wlist = ivf_wla; b.n_wlist = p2.nqb;
            b.wctr = ivf_ctr + 8;
            // subset stage: pass A starts from the bound its queries already have (see the unpruned case below): the
            // resulting thresholds -- the plan's radii -- can only be tighter, and stay valid (at least k rows of the FULL set
            // lie at or below the seed, whether pass A's tiles hold them or not)
            // (the direct-difference level ranks by the distance value itself: bound x (1 + 4 x its certificate's relative
            //  error), float32 inputs only -- float64 inputs add a coordinate-rounding term the seed does not model)
            const bool seed_a = !marks && !pilot && rerank && qmap && !forced && !(flags & CM_KNN_FORCE_FALLBACK) && !w3 && !cenl &&
                                seeds_from_level >= 1 && (level == 4 ? (sizeof(T) == 4 && seeds_from_level != 4)
                                                                     : ((seeds_from_level <= 3 && seeds_from_level <= level) || seeds_from_level == 5)) &&
                                !cm_get_knobs().no_subseed && !cm_get_knobs().no_seed;
            if (seed_a) {
                hipLaunchKernelGGL(seed_scatter_slot0, dim3((unsigned)((nsub + 255) / 256)), dim3(256), 0, st, (const float *)seedb,
                                   (const double *)bound, qmap, nsub, use_S, level == 4 ? 1 : 0, metric == CM_METRIC_COSINE ? 1 : 0,
                                   16.0 * (double)(d + 4) * 5.9604645e-8, ivf_cthr,
                                   (level != 4 && seeds_from_level == 5) ? (const float *)qnorm : (const float *)nullptr, (const float *)qlo,
                                   (const unsigned *)words, (const float *)scale3, c_eps, level);
                b.thr0_slot0 = 1;
            }
            side_items(ivf_wla, p2.nqb);
            CM_TRY(launch(b));
            // long runs are halved for load balance -- but every half is one more candidate list per query for the re-rank to
            // gather (a cluster split over six buckets filled all of them: 360 candidates per query, 17 ms of re-rank at the
            // 40-cluster config-2 shape).  With four or more blocks per compute unit the blocks balance the chip by themselves.
            const int split_len = level == 4 ? 32 : 256;
            hipLaunchKernelGGL(ivf_plan_b, dim3(p2.nqb), dim3(256), 0, st, (const double *)ivf_bcen, (const double *)ivf_brad,
                               (const double *)ivf_rcen, (const double *)ivf_rrad, (const int *)r_bstart, C, d, qmap, nsub,
                               (const float *)ivf_cthr, use_S, (const float *)qnorm, (const float *)qlo,
                               cenl ? (const float *)qnc : (const float *)nullptr, cenl ? (const float *)qcen : (const float *)nullptr,
                               (const unsigned *)words, (const float *)scale3, c_eps, level, (const int4 *)ivf_wla,
                               (const int *)ivf_jstar, ivf_wlb, ivf_ctr, BRt, sizeof(T) == 8 ? 1 : 0, d, split_len, ivf_sb, qblk,
                               chain ? ivf_runs : (int2 *)nullptr, 0, (const float *)nullptr, (const float *)nullptr, 0.f, (float *)nullptr, 0.f);
            CM_HIP(h, hipGetLastError());
            int *hc = h->d_nflag_pinned + 4;
            CM_HIP(h, hipMemcpyAsync(hc, ivf_ctr, 4 * sizeof(int), hipMemcpyDeviceToHost, st));
            CM_HIP(h, hipStreamSynchronize(st));
            unsigned long long tiles;
            memcpy(&tiles, hc + 2, 8);
            const unsigned long long full = (unsigned long long)p2.nqb * (unsigned long long)p2.n_tiles;
            if (dbg & 16)
                fprintf(stderr, "[cellmap dbg] bucket pruning: %llu of %llu (block, tile) pairs to visit, %d pass-B runs for %d blocks\n",
                        tiles, full, hc[0], p2.nqb);
            // Pruning pays when little is left: every list starts its own threshold history (pass A from +inf) and the
            // re-rank ranks several lists per query, so at 15-25 % of the pairs the pruned stage measured SLOWER than the
            // full scan (tests/fuzz/fuzz_paths.py); at an eighth and below it wins.
            if (tiles * 100ull <= full * (unsigned long long)(seed_a ? prune_max_sub_pct() : 12)) {
                if (hc[0] > 0) {
                    b.wlist = ivf_wlb; b.n_wlist = hc[0]; b.thr0_slot0 = 1;
                    b.wruns = chain ? ivf_runs : (const int2 *)nullptr;
                    b.wctr = ivf_ctr + 9;
                    side_items(ivf_wlb, hc[0], chain ? (const int2 *)ivf_runs : (const int2 *)nullptr);
                    CM_TRY(launch(b));
                }
                pruned = true;
                rr_cand = ivf_cand; rr_cthr = ivf_cthr; rr_S = use_S;
                // re-rank capacity: the lists are mostly empty; a query that overflows it is left to the next level -- on
                // the last level that would be the exact scan, so there the capacity covers every list
                const bool last_level = level == 4 || (level == 3 && !have_direct);
                rr_ncap = std::min(RR_NCMAX, (chain ? 2 : (last_level ? ivf_sb + 1 : 6)) * p2.Kc);
                // (seeded subset stage: the seeds are bounds plus error margins, on tight clusters dozens of rows per list lie
                //  below them -- a fifth of the 42 k queries the long-list centred level left at sigma 0.1 overflowed 512 slots)
                if (seed_a && !chain) rr_ncap = std::min(RR_NCMAX, (ivf_sb + 1) * p2.Kc);   // (every list in full: no overflow)
                if (marks) { ivf_pct = (int)std::max<unsigned long long>(1ull, tiles * 100ull / std::max<unsigned long long>(full, 1ull)); ivf_passes = 2; }
            } else {
                ivf_try = false;   // the buckets overlap too much: full scan, and do not try again for a while
                if (marks) { h->staged_ivf_skip = 15; h->staged_ivf_nref = n_ref; h->staged_ivf_nq = nq; }
            }
            }
        }
        if (!pruned) {
            if (cenl) CM_TRY(side_full_compute());
            // Sample-seeded thresholds.  A scan that starts from +inf thresholds appends ~K' ln(n/K') rows per query and
            // merges on a doubling schedule; most of that churn happens while the thresholds are still far from final.
            // So the hi*hi filter first scans every 64th row of the image with a short list: its K_s-th value is, for
            // nearly every query, a threshold with a few hundred rows below it in the full set -- the main scan starts
            // from it, appends a few times K' rows and merges only when a log fills.  Any starting threshold keeps the
            // invariant behind the certificate (non-candidates have values >= the final threshold); a query whose seed
            // was too tight simply ends with fewer than k candidates below it and moves on to the next level.
            const bool seed = marks && !pilot && rerank && !cenl && level != 4 && !w3 && p2.n_tiles >= 1024 &&
                              !forced && !(flags & CM_KNN_FORCE_FALLBACK) &&   // (pinned levels do not escalate: a seed failure would have nowhere to go)
                              !cm_get_knobs().no_seed;
            if (seed) {
                if (!sample_done) CM_TRY(sample_scan(a, a.Q));   // (a single-pass pruned attempt that did not pay left its seeds)
                a.thr0 = seed_thr;
                a.no_sched = 1;
            }
            // ... and the unpruned bucket-centred stage (round 5, late: overlapping clusters -- ragged, anisotropic mixtures -- take the
            // centred level WITHOUT pruning, 300 ms of unseeded scan at 1.5 M x 1.5 M) seeds from a centred scan of the same sample
            const bool seed_c = marks && !pilot && rerank && cenl && p2.n_tiles >= 1024 && !forced && !(flags & CM_KNN_FORCE_FALLBACK) &&
                                !cm_get_knobs().no_seed && !(dbg & 4096);   // (CELLMAP_DBG=4096: off -- A/B)
            if (seed_c) {
                CM_TRY(sample_scan(a, a.Q, 1, true));
                a.thr0 = seed_thr;
                a.no_sched = 1;
            }
            last_seeded = seed;
            // Subset stages start from what their queries already know: the previous stage left, per flagged query, the k-th
            // exact candidate value less |q'|^2 plus its filter error (knn_rerank, seedb) -- at least k rows have filter
            // values at or below it on this level too, so the scan appends a few times K' rows instead of K' ln(n / K') and
            // merges when a log fills.  Queries without a bound (fewer than k candidates so far) start from +inf.
            // (only after a level whose error bounds this level's from above: 1, 2, 3 -- not the bucket-centred level, whose
            //  error on tight clusters is far below the plain levels')
            if (sub_seed4_ok && wide_split) {   // direct level, small seeded subset: every split starts from slot 0 of the query's thresholds
                const int64_t n_ent = (int64_t)p2.nq_pad * p2.S;
                hipLaunchKernelGGL(fill_f32, dim3((unsigned)((n_ent + 255) / 256)), dim3(256), 0, st, cthr, n_ent, INFINITY);   // (padding rows: +inf)
                hipLaunchKernelGGL(seed_scatter_slot0, dim3((unsigned)((nsub + 255) / 256)), dim3(256), 0, st, (const float *)seedb,
                                   (const double *)bound, qmap, nsub, p2.S, 1, metric == CM_METRIC_COSINE ? 1 : 0,
                                   16.0 * (double)(d + 4) * 5.9604645e-8, cthr);
                a.thr0_slot0 = 1;
            }
            const bool sub_seed = !seed && sub_seed_ok;
            if (sub_seed) {
                if (seeds_from_level == 5)
                    hipLaunchKernelGGL(seed_gather, dim3((unsigned)((p2.nq_pad + 255) / 256)), dim3(256), 0, st, (const float *)seedb, qmap, nsub,
                                       (int64_t)p2.nq_pad, seed_sub, (const float *)qnorm, (const float *)qlo, (const unsigned *)words,
                                       (const float *)scale3, c_eps, level);
                else
                    hipLaunchKernelGGL(seed_gather, dim3((unsigned)((p2.nq_pad + 255) / 256)), dim3(256), 0, st, (const float *)seedb, qmap, nsub,
                                       (int64_t)p2.nq_pad, seed_sub);
                a.thr0 = seed_sub;
                a.no_sched = 1;
            }
            CM_TRY(launch(a));
        }
        if (marks) cm_mark(h, 2);
        if (!rerank) return CM_OK;
        const int pilot_kc = pilot ? p2.Kc : 0;
        const size_t rr_lds = rr_ncap ? rerank_lds_bytes(1, rr_ncap, d) : rerank_lds_bytes(p2.S, p2.Kc, d);
        hipLaunchKernelGGL(knn_rerank<T>, dim3((unsigned)((nsub + 3) / 4)), dim3(256), rr_lds, st, d_ref, d_qry, nsub, d, rr_cand, rr_cthr,
                           rr_S, p2.Kc, k, metric, form, qnorm, words, c_eps, c_split,
                           level == 4 ? (const float *)nullptr : scale3, level == 4 ? (const double *)nullptr : (const double *)d_mean,
                           idx_offset, flags, d_out_dist, d_out_idx, flag_out, (int *)(words + 2), bound, qmap, pilot_kc,
                           level == 4 ? (const float *)nullptr : (const float *)qlo, level, (const int *)rperm,
                           cenl ? (const float *)qnc : (const float *)nullptr, cenl ? (const float *)qcen : (const float *)nullptr, rr_ncap,
                           pilot ? (float *)nullptr : seedb);
        CM_HIP(h, hipGetLastError());
        if (marks) cm_mark(h, 3);
        return CM_OK;
    };
    auto read_flags = [&](int64_t *n) -> cm_status {
        CM_HIP(h, hipMemcpyAsync(h->d_nflag_pinned, words + 2, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
        CM_HIP(h, hipMemcpyAsync(h->d_nflag_pinned + 14, words + 22, 2 * sizeof(int), hipMemcpyDeviceToHost, st));   // (pinned [14] is free between searches)
        CM_HIP(h, hipStreamSynchronize(st));
        *n = h->d_nflag_pinned[0];
        n_unbounded = h->d_nflag_pinned[14];
        n_over = h->d_nflag_pinned[15];
        return CM_OK;
    };

    // bucket-centred mode: packed query order, bucket centroids, side values (one host read of the 64 bucket counts)
    auto setup_centred = [&]() -> cm_status {
        if (qpack) return CM_OK;
        const int C = LS_CMAX;
        int hcount[LS_CMAX], hps[LS_CMAX + 1];
        CM_HIP(h, hipMemcpyAsync(hcount, q_bcount, sizeof(int) * C, hipMemcpyDeviceToHost, st));
        CM_HIP(h, hipStreamSynchronize(st));
        int64_t run = 0;
        for (int j = 0; j < C; ++j) { hps[j] = (int)run; run += ((int64_t)hcount[j] + cmk2::QB - 1) / cmk2::QB * cmk2::QB; }
        hps[C] = (int)run;
        n_packed = run;
        side_stride = (int64_t)pf.n_tiles * cmk2::BR + 256;
        int *pstart = (int *)cm_ws_alloc(h, sizeof(int) * (C + 1));
        qpack = (int *)cm_ws_alloc(h, sizeof(int) * (size_t)n_packed);
        blk_bucket = (int *)cm_ws_alloc(h, sizeof(int) * (size_t)(n_packed / cmk2::QB + 1));
        cen = (double *)cm_ws_alloc(h, sizeof(double) * (size_t)C * d);
        side = (float *)cm_ws_alloc(h, sizeof(float) * (size_t)C * side_stride);
        qnc = (float *)cm_ws_alloc(h, sizeof(float) * (size_t)n_packed);
        qcen = (float *)cm_ws_alloc(h, sizeof(float) * (size_t)n_packed);
        const size_t claim_words = 2 * (size_t)C * (size_t)((pf.n_tiles + 31) / 32 + 1);   // one bit per (bucket, tile) pair, twice (done / wanted): cen_side_mark
        side_claimed = (unsigned *)cm_ws_alloc(h, sizeof(unsigned) * claim_words);
        if (!pstart || !qpack || !blk_bucket || !cen || !side || !qnc || !qcen || !side_claimed) CM_FAIL(h, CM_E_NOMEM, "k-NN: device workspace exhausted");
        CM_HIP(h, hipMemsetAsync(blk_bucket, 0, sizeof(int) * (size_t)(n_packed / cmk2::QB + 1), st));   // (the four-accumulator kernel pads to 1 024 queries: one more block, bucket 0)
        CM_HIP(h, hipMemsetAsync(side_claimed, 0, sizeof(unsigned) * claim_words, st));
        CM_HIP(h, hipMemcpyAsync(pstart, hps, sizeof(int) * (C + 1), hipMemcpyHostToDevice, st));
        CM_HIP(h, hipStreamSynchronize(st));   // (hps lives on this stack frame)
        double *cen_part = (double *)cm_ws_alloc(h, sizeof(double) * (size_t)C * CEN_SLICES * 64);
        if (!cen_part) CM_FAIL(h, CM_E_NOMEM, "k-NN: device workspace exhausted");
        hipLaunchKernelGGL(cen_centroid<T>, dim3(C, CEN_SLICES), dim3(256), 0, st, d_qry, d, metric, (const double *)d_mean, (const int *)qperm,
                           (const int *)q_bstart, cen_part);
        hipLaunchKernelGGL(cen_centroid_final, dim3(C), dim3(64), 0, st, (const double *)cen_part, (const int *)q_bstart, d, cen);
        hipLaunchKernelGGL(cen_pack, dim3((unsigned)((n_packed + 255) / 256)), dim3(256), 0, st, (const int *)qperm, (const int *)q_bstart,
                           (const int *)pstart, C, n_packed, qpack, blk_bucket);
        CM_HIP(h, hipGetLastError());
        return CM_OK;   // (the side values are computed by the stage that needs them: side_full_compute / cen_side_mark + cen_side_compute)
    };

    // ---- choose the level the whole set starts at
    int start = lowp ? (have_grid ? 0 : 4) : 1;
    const bool hint_ok = h->staged_hint_d == d && h->staged_hint_k == k && h->staged_hint_metric == metric &&
                         !cm_get_knobs().no_hint;
    // (round 5: a set that STARTED on the three-product or the direct level and certified most of its queries there starts there
    //  again -- its pilot, two sample stages over the whole reference, was 30 of the 230 ms of the 40-cluster d = 128 shape)
    // (hints 3, 4 and 6 -- start levels that give up on cheaper ones -- hold for the problem size they were learnt on: another data set of
    //  the same width gets its pilot again)
    const bool same_size = h->staged_long_nref == n_ref && h->staged_long_nq == nq;
    const bool hinted = hint_ok && (h->staged_hint_level == 1 || (h->staged_hint_level == 3 && same_size) || (h->staged_hint_level == 4 && have_direct && same_size) ||
                                    ((h->staged_hint_level == 5 || (h->staged_hint_level == 6 && cen_long)) && have_cen));
    const bool cen_bad = hint_ok && h->staged_hint_level == -5 && same_size;   // the centred level failed for most queries last time (on this problem size)
    if (lowp) {}   // grid or direct kernel: nothing to choose
    else if (forced) start = forced;
    else if (flags & CM_KNN_FORCE_FALLBACK) start = 3;
    else if (hinted) start = h->staged_hint_level == 6 ? 5 : h->staged_hint_level;   // that level certified most queries last time: no pilot
    else if (no_structure) {}   // structure-free embeddings: nearest-neighbour distances are of the order of the data's spread, the
                                // hi*hi level's error bound is far below the gaps it has to resolve -- and if not, the ladder escalates
    else if (n_pilot > 0 && !(dbg & 3)) {
        hipLaunchKernelGGL(strided_rows, dim3((n_pilot + 255) / 256), dim3(256), 0, st, sample, n_pilot, nq / n_pilot);
        int64_t nf = 0;
        CM_TRY(run_stage(1, sample, n_pilot, flag_b, true, false, true));
        CM_TRY(read_flags(&nf));
        if (nf * 2 > n_pilot && have_cen && !cen_bad) {
            start = 5;   // the cancellation that defeats level 1 shrinks with the bucket radius: try it centred
        } else if (nf * 2 > n_pilot) {   // level 1 would fail for more than half: not worth its time
            CM_TRY(run_stage(3, sample, n_pilot, flag_b, true, false, true));
            CM_TRY(read_flags(&nf));
            start = (have_direct && nf * 5 > (int64_t)n_pilot * 4) ? 4 : 3;
        }
    }

    // ---- the ladder
    if (start == 5 && !have_cen) start = 1;
    if (start == 5) CM_TRY(setup_centred());
    const int *qmap = start == 5 ? (const int *)qpack : (const int *)qperm;   // full set: bucket-packed / locality-sorted order
    int64_t nsub = start == 5 ? n_packed : nq;
    int *flag_out = flag_a;
    int level = start, main_Kc = Kc_max, main_S = 1, main_variant = 0;
    for (int stage = 0;; ++stage) {
        CM_TRY(run_stage(level, qmap, nsub, flag_out, !(dbg & 3), stage == 0, false));
        if (cen_abandon) {   // a small set on the centred level that cannot be pruned: the padding is not worth it
            cen_abandon = false;
            if (dbg & 16) fprintf(stderr, "[cellmap dbg] centred level abandoned (%lld queries, nothing to prune): three-product level\n", (long long)nq);
            h->staged_hint_level = -5; h->staged_hint_d = d; h->staged_hint_k = k; h->staged_hint_metric = metric;
            h->staged_long_nref = n_ref; h->staged_long_nq = nq;
            level = start = 3; qmap = (const int *)qperm; nsub = nq;
            --stage;
            continue;
        }
        if (stage == 0) { main_Kc = last_Kc; main_S = last_S; main_variant = last_variant; }
        if (dbg & 3) break;   // timing experiments: phase 1 only
        // A wide split counts on lists that hold a few rows beyond k.  On very tight clusters the seeds' error margins admit
        // hundreds of rows (sigma 0.1 at a spread of 5: more than the 1 024 the re-rank compacts, for 1 172 of 1 178 queries):
        // those queries are not sent on to a coarser-grained level -- the stage is repeated for its flagged queries with the
        // ordinary split count, whose lists the re-rank takes in full.
        if (last_wide) {
            int64_t nf0 = 0;
            CM_TRY(read_flags(&nf0));
            if (n_over > 0) {
                if (dbg & 16) fprintf(stderr, "[cellmap dbg] stage %d level %d: %lld of %lld flagged queries over the re-rank's capacity after a wide split: the level again, ordinary splits\n",
                                      stage, level, (long long)n_over, (long long)nf0);
                qmap = flag_out; nsub = nf0;
                flag_out = flag_out == flag_a ? flag_b : flag_a;
                seeds_from_level = level;
                no_wide = true;
                ++h->knn_wide_retries;
                continue;
            }
        }
        // (after a seeded stage the failures first get an unseeded pass of the same level: the image is in place, and
        // most of them only had a seed that was too tight)
        const int next = (forced || (flags & CM_KNN_FORCE_FALLBACK)) ? 0
                         : (last_seeded ? level
                            : (level == 0 ? 4 : (level == 4 ? 0 : ((level < 3 || level == 5) ? 3 : (have_direct ? 4 : 0)))));
        if (!next) break;
        int64_t nf = 0;
        CM_TRY(read_flags(&nf));
        // The bucket-centred level is the most precise of the matrix-core filters on clustered data; when it leaves more than
        // a fifth of its queries uncertified the clusters are tight against their offsets and the three-product level fails
        // for most of them as well (40 clusters, sigma 0.1: 69 % again): go straight to the direct-difference filter.
        // A small remainder is cheaper on the three-product level (sigma 0.2: 6 ms against 36).
        int next_level = next;
        if (level == 5 && next == 3 && have_direct && nf * 5 > nsub && !cm_get_knobs().no_skip3) next_level = 4;
        if (stage == 0 && !forced && !lowp) {   // remember whether the cheapest level is worth starting with next time
            const bool good = (level == 1 || level == 3 || level == 4 || level == 5) && nf * 2 <= nq;
            // (the centred level with short lists failed for most: next time with long lists; those failed too: not again)
            const bool long_ok = k + (k > 30 ? k : 30) + 80 + 2 * cmk2::CL <= 512;
            if (level == 5 && !cen_long)
                // (short lists: beyond a twelfth uncertified the long lists are cheaper than the stage that takes the failures --
                //  measured at 1.5 M x 1.5 M: 6.5 % -> 12 ms for that stage against ~15 ms for the longer re-rank; 23 % -> 98 ms)
                h->staged_hint_level = (nf * 12 > nq && long_ok) ? 6 : (good ? 5 : -5);
            else
                h->staged_hint_level = good ? (level == 5 ? 6 : level) : (level == 5 ? -5 : 0);
            h->staged_hint_d = d; h->staged_hint_k = k; h->staged_hint_metric = metric;
            if (h->staged_hint_level == 6 || h->staged_hint_level == 3 || h->staged_hint_level == 4 || h->staged_hint_level == -5) { h->staged_long_nref = n_ref; h->staged_long_nq = nq; }
        }
        if (dbg & 16) {
            unsigned gc[4] = {0, 0, 0, 0}, why[2] = {0, 0};
            (void)hipMemcpy(gc, words + 12, 16, hipMemcpyDeviceToHost);
            (void)hipMemcpy(why, words + 23, 8, hipMemcpyDeviceToHost);
            fprintf(stderr, "[cellmap dbg] stage %d level %d: %lld of %lld queries uncertified, %lld without a bound, %u over the re-rank's capacity, %u with fewer than k candidates (grid give-ups: <k %u, >cap %u, >K' %u)\n",
                    stage, level, (long long)nf, (long long)nsub, (long long)n_unbounded, why[0], why[1], gc[0], gc[1], gc[2]);
        }
        if (nf == 0) break;
        // the sliced exact scan is cheaper than another stage for a handful of queries -- except after the grid stage,
        // whose give-ups carry no usable bound (no candidates were evaluated for them)
        // ... nor after a seeded stage: a query whose seed was too tight may hold fewer than k candidates (bound = inf)
        // (round 5: counted, not assumed -- the re-rank counts the flagged queries whose bound is still +inf)
        if (nf <= cm_get_knobs().fb_max && level != 0 && n_unbounded == 0) break;
        qmap = flag_out;
        nsub = nf;
        if (ivf_try && flag_s && nf >= cm_get_knobs().ivf_submin) {   // group the flagged queries by bucket: the next stage can prune as well
            hipLaunchKernelGGL(zero_words, dim3(2), dim3(256), 0, st, (unsigned *)fl_cnt, 2 * LS_CMAX);
            hipLaunchKernelGGL(fl_labels, dim3((unsigned)((nf + LS_ROWS - 1) / LS_ROWS)), dim3(256), 0, st, (const int *)flag_out, nf,
                               (const int *)q_lab, fl_tmp, fl_cnt);
            hipLaunchKernelGGL(ls_offsets, dim3(1), dim3(64), 0, st, fl_cnt, LS_CMAX, fl_cnt + LS_CMAX);
            hipLaunchKernelGGL(ls_scatter, dim3((unsigned)((nf + LS_ROWS - 1) / LS_ROWS)), dim3(256), 0, st, (const int *)fl_tmp, nf,
                               fl_cnt + LS_CMAX, fl_tmp + (nq + 1));
            hipLaunchKernelGGL(fl_gather, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, st, (const int *)flag_out,
                               (const int *)(fl_tmp + (nq + 1)), nf, flag_s);
            CM_HIP(h, hipGetLastError());
            qmap = flag_s;
        }
        flag_out = flag_out == flag_a ? flag_b : flag_a;
        seeds_from_level = level;   // (the stage that just ran flagged these queries and left their seeds)
        level = next_level;
    }
    if (g_dbg_ctr) {
        unsigned long long hc2[32] = {0};
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(hc2, g_dbg_ctr, 256, hipMemcpyDeviceToHost);
        for (int part = 0; part < 2; ++part) {
            const unsigned long long *hc = hc2 + 16 * part;
            if (part && !hc[5]) continue;
            fprintf(stderr, "[cellmap dbg] %s branches=%llu appended=%llu prunes=%llu prune_cycles=%llu wave_cycles=%llu waves=%llu event_cycles=%llu barrier_cycles=%llu dma_wait_cycles=%llu mfma_cluster_cycles=%llu prologue_cycles=%llu loop_cycles=%llu final_cycles=%llu\n",
                    part ? "sample-scan" : "filter", hc[0], hc[1], hc[2], hc[3], hc[4], hc[5], hc[6], hc[7], hc[8], hc[9], hc[10], hc[11], hc[12]);
        }
    }
    if (dbg & 3) return CM_OK;

    // ---- exact float64 scan for the queries no stage could certify
    hipLaunchKernelGGL(zero_words, dim3((FBS_NQ + 255) / 256), dim3(256), 0, st, (unsigned *)fb_count, FBS_NQ);
    {
        int rc = (int)((52 * 1024 - (d >= RR_DMAX ? sizeof(double) * FBS_QG * q_stride(d) / 2 : 0)) / ((size_t)d * sizeof(T)));   // rows per staged chunk
        rc = rc >= 256 ? 256 : (rc >= 32 ? rc / 32 * 32 : (rc < 1 ? 1 : rc));
        const size_t fb_lds = (size_t)rc * d * sizeof(T) + sizeof(double) * FBS_QG * q_stride(d);
        auto scan = knn_fallback_scan<T>;
        CM_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void *>(scan), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fb_lds));
        hipLaunchKernelGGL(scan, dim3(FBS_SLICES, FBS_FY), dim3(256), fb_lds, st, d_ref, n_ref, d_qry, d, metric, form, flag_out,
                           (const int *)(words + 2), bound, fb_count, fb_v, fb_i, rc);
    }
    hipLaunchKernelGGL(knn_fallback_rank<T>, dim3(FBS_NQ), dim3(256), 0, st, k, metric, idx_offset, flags, flag_out,
                       (const int *)(words + 2), (const int *)fb_count, (const double *)fb_v, (const int *)fb_i, d_out_dist,
                       d_out_idx, (int *)(words + 3), ovf_list);
    hipLaunchKernelGGL(knn_exact_fallback<T>, dim3(FB_BLOCKS), dim3(256), sizeof(double) * q_stride(d), st, d_ref, n_ref, d_qry, d, k, metric, form,
                       idx_offset, flags, flag_out, (const int *)(words + 2), bound, d_out_dist, d_out_idx,
                       (int *)(words + 3), ovf_list, FBS_NQ);
    // queries with more than FB_CAP rows inside their bound (words[20] counts them; empty launch otherwise)
    hipLaunchKernelGGL(knn_fallback_select<T>, dim3(256), dim3(256), sizeof(double) * q_stride(d), st, d_ref, n_ref, d_qry, d, k, metric, form,
                       idx_offset, flags, (const int *)(words + 3), (const int *)ovf_list, d_out_dist, d_out_idx);
    CM_HIP(h, hipGetLastError());
    cm_mark(h, 4);
    CM_HIP(h, hipMemcpyAsync(h->d_nflag_pinned, words + 2, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
    h->knn_stats[1] = main_Kc;
    h->knn_stats[2] = main_S + 100 * start + 10000 * main_variant;
    h->knn_stats[3] = (wide ? 4 : (lowp ? 3 : 2)) + 100 * ivf_pct + 100000 * ivf_passes;
    return CM_OK;
}

// ---- shapes outside every filter family: the exact scan (knn_exact_scan)
bool termwise_metric(int metric) {
    return metric == CM_METRIC_MANHATTAN || metric == CM_METRIC_CHEBYSHEV || metric == CM_METRIC_CANBERRA || metric == CM_METRIC_BRAYCURTIS;
}
bool needs_exact_scan(int64_t n_ref, int64_t nq, int d, int k, int metric) {
    if (metric >= CM_METRIC_HAMMING) return true;   // the counting metrics: no filter at any width
    if (termwise_metric(metric)) return lowd_dpad(d) > 128 || k + knn_slack(k) + 2 * cmk2::CL > 512;   // (the VALU direct filter's limits)
    if (phase1_path(d, k) != 0) return false;
    KnnPlan p;
    return make_plan_quiet(n_ref, nq, d, k, &p) != 0;   // k beyond the f32 filter's candidate buffers
}
// list capacity: k, room to grow between cuts (k / 2, at least 256) and the 256 slots one row block can fill
int exact_scan_cap(int k) { const int c = k + (k / 2 > 256 ? k / 2 : 256) + XS_T; return (c + 255) / 256 * 256; }
constexpr size_t XS_LDS_MAX = 160 * 1024;

template <typename T>
cm_status knn_run_scan(cm_handle_t h, const T *d_ref, int64_t n_ref, const T *d_qry, int64_t nq, int d, int k, int metric,
                       int form, int64_t idx_offset, int flags, void *d_out_dist, int64_t *d_out_idx) {
    hipStream_t st = h->stream;
    // k beyond the on-chip lists of one query per workgroup (~4 200 entries): passes of k_pass entries (knn_exact_scan, lb_*)
    int k_pass = k;
    while (exact_scan_lds_bytes<T>(1, exact_scan_cap(k_pass)) > XS_LDS_MAX && k_pass > 64) k_pass = k_pass > 4096 ? 4096 : k_pass / 2;
    if (exact_scan_lds_bytes<T>(1, exact_scan_cap(k_pass)) > XS_LDS_MAX) CM_FAIL(h, CM_E_INTERNAL, "k-NN: exact scan list sizing");
    const int CAP = exact_scan_cap(k_pass);
    int QG = 0;
    for (int g : {4, 2, 1})
        if (exact_scan_lds_bytes<T>(g, CAP) <= (g == 1 ? XS_LDS_MAX : (size_t)80 * 1024)) { QG = g; break; }   // (two workgroups per CU while the lists allow)
    if (!QG) CM_FAIL(h, CM_E_INTERNAL, "k-NN: exact scan list sizing");
    unsigned *words = (unsigned *)cm_ws_alloc(h, 128);
    double *lb_v = nullptr;
    int *lb_i = nullptr;
    if (k_pass < k) {
        lb_v = (double *)cm_ws_alloc(h, sizeof(double) * (size_t)std::max<int64_t>(nq, 1));
        lb_i = (int *)cm_ws_alloc(h, sizeof(int) * (size_t)std::max<int64_t>(nq, 1));
        if (!lb_v || !lb_i) CM_FAIL(h, CM_E_NOMEM, "k-NN: device workspace exhausted");
    }
    if (!words) CM_FAIL(h, CM_E_NOMEM, "k-NN: device workspace exhausted");
    cm_mark(h, 0);
    hipLaunchKernelGGL(zero_words, dim3(1), dim3(64), 0, st, words, 32);
    cm_mark(h, 1);
    cm_mark(h, 2);
    cm_mark(h, 3);
    if (nq > 0) {
        const size_t lds = exact_scan_lds_bytes<T>(QG, CAP);
        const bool vec = d % (16 / (int)sizeof(T)) == 0 && (reinterpret_cast<uintptr_t>(d_ref) & 15) == 0;   // 16-byte row pieces
        auto kern = vec ? (QG == 4 ? knn_exact_scan<T, 4, true> : (QG == 2 ? knn_exact_scan<T, 2, true> : knn_exact_scan<T, 1, true>))
                        : (QG == 4 ? knn_exact_scan<T, 4, false> : (QG == 2 ? knn_exact_scan<T, 2, false> : knn_exact_scan<T, 1, false>));
        CM_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        for (int off = 0; off < k; off += k_pass) {
            const int kp = std::min(k_pass, k - off);
            hipLaunchKernelGGL(kern, dim3((unsigned)((nq + QG - 1) / QG)), dim3(XS_T), lds, st, d_ref, n_ref, d_qry, nq, d, kp, metric, form,
                               idx_offset, flags, CAP, d_out_dist, d_out_idx, k, off, lb_v, lb_i, off > 0 ? 1 : 0);
        }
        CM_HIP(h, hipGetLastError());
    }
    cm_mark(h, 4);
    CM_HIP(h, hipMemcpyAsync(h->d_nflag_pinned, words + 2, 2 * sizeof(int), hipMemcpyDeviceToHost, st));   // (no filter: nothing flagged)
    h->knn_stats[1] = CAP;
    h->knn_stats[2] = QG;
    h->knn_stats[3] = 5;
    h->knn_sub_splits = 0;
    h->knn_wide_retries = 0;
    return CM_OK;
}

// Entry point of the search.  d <= 128 (and k small enough for the in-register merges) runs the staged search above;
// wider rows or very large k use the f32-input MFMA filter (knn_phase1.h: LDS candidate buffers, one pass, certificate
// from the fmaf-chain bound), then the same re-rank and exact fallback.
template <typename T>
cm_status knn_run_t(cm_handle_t h, const T *d_ref, int64_t n_ref, const T *d_qry, int64_t nq, int d, int k, int metric,
                    int form, int64_t idx_offset, int flags, void *d_out_dist, int64_t *d_out_idx) {
    hipStream_t st = h->stream;
    if (needs_exact_scan(n_ref, nq, d, k, metric))
        return knn_run_scan<T>(h, d_ref, n_ref, d_qry, nq, d, k, metric, form, idx_offset, flags, d_out_dist, d_out_idx);
    if (phase1_path(d, k) != 0 || metric == CM_METRIC_MANHATTAN || metric == CM_METRIC_CHEBYSHEV || metric == CM_METRIC_CANBERRA || metric == CM_METRIC_BRAYCURTIS)
        return knn_run_staged<T>(h, d_ref, n_ref, d_qry, nq, d, k, metric, form, idx_offset, flags, d_out_dist, d_out_idx);
    KnnPlan p;
    CM_TRY(make_plan(h, n_ref, nq, d, k, &p));
    const int Kc = p.Kc, S = p.S;
    const int64_t nq_pad = p.nq_pad;
    float *qnorm = (float *)cm_ws_alloc(h, sizeof(float) * nq_pad);
    int *cand = (int *)cm_ws_alloc(h, sizeof(int) * nq_pad * S * Kc);
    float *cthr = (float *)cm_ws_alloc(h, sizeof(float) * nq_pad * S);
    int *flag_list = (int *)cm_ws_alloc(h, sizeof(int) * (nq + 1));
    double *bound = (double *)cm_ws_alloc(h, sizeof(double) * (nq + 1));
    unsigned *words = (unsigned *)cm_ws_alloc(h, 128);  // [0..1] gmax, [2] flag count, [3] overflow
    int *fb_count = (int *)cm_ws_alloc(h, sizeof(int) * FBS_NQ);                       // sliced exact fallback: candidate lists
    double *fb_v = (double *)cm_ws_alloc(h, sizeof(double) * (size_t)FBS_NQ * FB_CAP);
    int *fb_i = (int *)cm_ws_alloc(h, sizeof(int) * (size_t)FBS_NQ * FB_CAP);
    int *ovf_list = (int *)cm_ws_alloc(h, sizeof(int) * OVF_CAP);
    float *Rp = (float *)cm_ws_alloc(h, sizeof(float) * p.nref_pad * p.KP);
    float *Qp = (float *)cm_ws_alloc(h, sizeof(float) * p.nq_pad * p.KP);
    if (!qnorm || !cand || !cthr || !flag_list || !bound || !words || !fb_count || !fb_v || !fb_i || !ovf_list || !Rp || !Qp)
        CM_FAIL(h, CM_E_NOMEM, "k-NN: device workspace exhausted");
    cm_mark(h, 0);
    hipLaunchKernelGGL(zero_words, dim3(1), dim3(64), 0, st, words, 32);
    const int bs = 256;
    hipLaunchKernelGGL(prep_rows<T>, dim3((unsigned)((p.nref_pad + bs - 1) / bs)), dim3(bs), 0, st, d_ref, n_ref, d,
                       p.KP, p.nref_pad, 1, metric, Rp, (float *)nullptr, words);
    hipLaunchKernelGGL(prep_rows<T>, dim3((unsigned)((p.nq_pad + bs - 1) / bs)), dim3(bs), 0, st, d_qry, nq, d, p.KP,
                       p.nq_pad, 0, metric, Qp, qnorm, (unsigned *)nullptr);
    CM_HIP(h, hipGetLastError());
    cm_mark(h, 1);
    dim3 grid((unsigned)(p.nq_pad / (p.NW * 32)), (unsigned)p.S);
    cm_status s = CM_E_INTERNAL;
    bool ok = (d > 128 && ((s = cm_p1_wide_launch(h, p.KP, p.NW, grid, p.lds, Qp, Rp, p.tps, p.n_tiles, p.Kc, p.C, cand, cthr)), true)) ||
              cm_p1_launch_g0(h, p.KP, p.NW, grid, p.lds, Qp, Rp, p.tps, p.n_tiles, p.Kc, p.C, cand, cthr, &s) ||
              cm_p1_launch_g1(h, p.KP, p.NW, grid, p.lds, Qp, Rp, p.tps, p.n_tiles, p.Kc, p.C, cand, cthr, &s) ||
              cm_p1_launch_g2(h, p.KP, p.NW, grid, p.lds, Qp, Rp, p.tps, p.n_tiles, p.Kc, p.C, cand, cthr, &s) ||
              cm_p1_launch_g3(h, p.KP, p.NW, grid, p.lds, Qp, Rp, p.tps, p.n_tiles, p.Kc, p.C, cand, cthr, &s);
    if (!ok) CM_FAIL(h, CM_E_INTERNAL, "k-NN: no phase-1 instantiation for KP=%d", p.KP);
    CM_TRY(s);
    cm_mark(h, 2);
    // fp32 error bound of the MFMA fmaf chain (gamma_n, n = KP + slack for the float conversions): see DESIGN.md
    const float c_eps = (float)(p.KP + 16) * 5.9604645e-8f * 1.01f;
    // rows wider than KNN_DMAX: the kernels that stage query rows in LDS read them from global memory instead (GQ), and every
    // flagged query goes to the one-workgroup-per-query scan (the sliced scan stages reference rows in LDS as well)
    const bool gq = d > KNN_DMAX;
    auto rr = gq ? knn_rerank<T, true> : knn_rerank<T, false>;
    hipLaunchKernelGGL(rr, dim3((unsigned)((nq + 3) / 4)), dim3(256), rerank_lds_bytes(S, Kc, d), st, d_ref, d_qry, nq, d, (const int *)cand,
                       (const float *)cthr, S, Kc, k, metric, form, (const float *)qnorm, (const unsigned *)words, c_eps, 0.f,
                       (const float *)nullptr, (const double *)nullptr, idx_offset, flags, d_out_dist, d_out_idx, flag_list,
                       (int *)(words + 2), bound, (const int *)nullptr, 0, (const float *)nullptr, 0, (const int *)nullptr,
                       (const float *)nullptr, (const float *)nullptr, 0, (float *)nullptr);
    cm_mark(h, 3);
    hipLaunchKernelGGL(zero_words, dim3((FBS_NQ + 255) / 256), dim3(256), 0, st, (unsigned *)fb_count, FBS_NQ);
    if (!gq) {
        int rc = (int)((52 * 1024 - (d >= RR_DMAX ? sizeof(double) * FBS_QG * q_stride(d) / 2 : 0)) / ((size_t)d * sizeof(T)));   // rows per staged chunk
        rc = rc >= 256 ? 256 : (rc >= 32 ? rc / 32 * 32 : (rc < 1 ? 1 : rc));
        const size_t fb_lds = (size_t)rc * d * sizeof(T) + sizeof(double) * FBS_QG * q_stride(d);
        auto scan = knn_fallback_scan<T>;
        CM_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void *>(scan), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fb_lds));
        hipLaunchKernelGGL(scan, dim3(FBS_SLICES, FBS_FY), dim3(256), fb_lds, st, d_ref, n_ref, d_qry, d, metric, form, flag_list,
                           (const int *)(words + 2), bound, fb_count, fb_v, fb_i, rc);
        hipLaunchKernelGGL(knn_fallback_rank<T>, dim3(FBS_NQ), dim3(256), 0, st, k, metric, idx_offset, flags, flag_list,
                           (const int *)(words + 2), (const int *)fb_count, (const double *)fb_v, (const int *)fb_i, d_out_dist,
                           d_out_idx, (int *)(words + 3), ovf_list);
    }
    auto xf = gq ? knn_exact_fallback<T, true> : knn_exact_fallback<T, false>;
    hipLaunchKernelGGL(xf, dim3(FB_BLOCKS), dim3(256), gq ? 0 : sizeof(double) * q_stride(d), st, d_ref, n_ref, d_qry, d, k, metric, form,
                       idx_offset, flags, (const int *)flag_list, (const int *)(words + 2), (const double *)bound, d_out_dist, d_out_idx,
                       (int *)(words + 3), ovf_list, gq ? 0 : FBS_NQ);
    // queries with more than FB_CAP rows inside their bound (words[20] counts them; empty launch otherwise)
    auto fs = gq ? knn_fallback_select<T, true> : knn_fallback_select<T, false>;
    hipLaunchKernelGGL(fs, dim3(256), dim3(256), gq ? 0 : sizeof(double) * q_stride(d), st, d_ref, n_ref, d_qry, d, k, metric, form,
                       idx_offset, flags, (const int *)(words + 3), (const int *)ovf_list, d_out_dist, d_out_idx);
    CM_HIP(h, hipGetLastError());
    cm_mark(h, 4);
    // stats mirror (read back lazily by cm_knn_stats after a synchronize)
    CM_HIP(h, hipMemcpyAsync(h->d_nflag_pinned, words + 2, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
    h->knn_stats[1] = Kc;
    h->knn_stats[2] = S;
    h->knn_stats[3] = 1;
    h->knn_sub_splits = 0;
    h->knn_wide_retries = 0;
    return CM_OK;
}

}  // namespace

// inputs of the ratio metrics: is there a non-zero magnitude below 2^-100 (see cm_knn_run)?
template <typename T>
static __global__ void tiny_nonzero(const T *__restrict__ x, int64_t n, unsigned *__restrict__ flag) {
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double a = fabs((double)x[i]);
        bad |= a > 0.0 && a < 7.888609052210118e-31;
    }
    if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

// Workspace a search reserves on entry: what the COMMON route of a problem of this shape carves (every unconditional array
// of knn_run_staged / knn_run_t, the structure pre-test, the threshold seeds, the locality sort).  Routes only clustered or
// difficult data take -- bucket pruning (a candidate list per query and sub-bucket: 5.7 GB at config 2), the bucket-centred
// level (256 side values per reference row: 1.5 GB) -- add a workspace segment when they run (cm_ws_alloc grows on demand);
// round 4 reserved the sum of every route up front: 16 GB at config 2, of which the benchmark's route touches 3.
size_t cm_knn_ws_bytes(int64_t n_ref, int64_t nq, int32_t d, int32_t k, int32_t dtype, int32_t metric) {
    if (metric == CM_METRIC_CORRELATION)   // float64 centred copies of both sides, then the cosine search over them
        return cm_knn_ws_bytes(n_ref, nq, d, k, CM_F64, CM_METRIC_COSINE) + cm_align(8 * (size_t)n_ref * d) + cm_align(8 * (size_t)nq * d) + 4096;
    (void)dtype;
    if (needs_exact_scan(n_ref, nq, d, k, metric)) return (size_t)1 << 20;   // (the exact scan keeps its lists on chip)
    int slack = knn_slack_level(k, 1);
    int Kc = k + slack;
    int64_t nq_pad = (nq + 511) / 512 * 512;
    // rows of the per-split candidate lists: reference splits exist only while there are fewer query blocks than resident
    // workgroups (make_plan / make_plan2), so blocks x splits stays below twice the resident count
    const int64_t split_rows = std::min<int64_t>(nq_pad * 16, nq_pad + (int64_t)2 * 640 * 512);
    size_t b = 512 * 1024 + (1 << 20) /* structure pre-test (sg_*) */ + 2 * cm_align(4 * nq_pad) + cm_align(sizeof(int) * split_rows * Kc) + cm_align(4 * split_rows) +
               cm_align(4 * (nq + 1)) + cm_align(8 * (nq + 1)) + 4096 + cm_align(4 * FBS_NQ) + cm_align((size_t)12 * FBS_NQ * FB_CAP) + cm_align(4 * OVF_CAP) + 1024;
    // (d <= 128 always takes the staged sizing: the manhattan / chebyshev metrics run the staged search whatever k is, and
    //  this function does not see the metric)
    if (use_f16_path(d, k) || d <= 128) {
        int KSTEPS = d + cmk2::NORM_SLOTS <= 32 ? 2 : (d + cmk2::NORM_SLOTS <= 64 ? 4 : 8);
        int64_t n_tiles = (n_ref + cmk2::BR - 1) / cmk2::BR;
        size_t tileb = (size_t)cmk2::tile_bytes(KSTEPS > 4 ? KSTEPS : 4) + 1024;   // any tile image of the staged search: <= 256 B per row
        const int64_t pack_slack = (int64_t)256 * cmk2::QB;
        b += cm_align((size_t)n_tiles * tileb + 65536) + cm_align((size_t)(nq_pad + pack_slack) * (KSTEPS > 4 || lowd_dpad(d) > 64 ? 512 : 256));
        b += cm_align(sizeof(uint2) * (size_t)320 * 8 * 128 * (Kc + 16)) + cm_align(sizeof(uint2) * (size_t)320 * 8 * 128 * 2 * cmk2::CL);
        // staged search (knn_run_staged): the candidate buffer serves subset stages with more reference splits than the full set
        const int64_t cand_rows = std::max<int64_t>(std::max<int64_t>(split_rows, nq_pad + (int64_t)cmk2::QB * 512),
                                                    std::min<int64_t>(4 * (nq_pad + 1024), nq_pad + 1024 + 2000000));
        b += cm_align(sizeof(int) * (size_t)cand_rows * (Kc + 16)) + cm_align(4 * (size_t)cand_rows) + 3 * cm_align(4 * (size_t)(nq_pad + pack_slack)) +
             3 * cm_align(4 * (nq + 1)) + cm_align(8 * (nq + 1)) + 65536;
        b += cm_align((size_t)(n_ref / 16 + 512) * 136) + cm_align(4 * (size_t)(n_ref / 16 + 256)) +
             cm_align(4 * (size_t)(nq_pad + 1024)) + 4096;   // threshold seeds: hi-only image of a 1/R sample (R >= 16), row map, seeds
        if (d <= 3) b += cm_grid_ws_bytes(n_ref, cm_grid_ncell_max(n_ref, k));
        b += cm_align(4 * (size_t)std::max<int64_t>(n_ref, nq)) + cm_align(4 * (size_t)n_ref) + cm_align(4 * (size_t)nq) + 3 * cm_align(4 * (size_t)(nq + 1)) + 65536;   // locality sort
    }
    if (!use_f16_path(d, k)) {   // f32-input kernel: prepared rows of KP floats
        int KP = pick_kp(d);
        if (KP < 0 || d > 128) KP = (d + 1 + cmk::KCW - 1) / cmk::KCW * cmk::KCW;   // chunked kernel: padded to the chunk width
        int64_t nref_pad = (n_ref + cmk::BR - 1) / cmk::BR * cmk::BR;
        b += cm_align(sizeof(float) * nref_pad * KP) + cm_align(sizeof(float) * nq_pad * KP);
    }
    return b + (b >> 6) + (8u << 20);   // (+1.5 % and 8 MB: alignment of the many small allocations)
}

// The common route plus what the routes of CLUSTERED data add on demand (bucket pruning's candidate lists and plans, the
// bucket-centred level's packed maps and side values): for callers that know their embeddings are clustered and want even the
// first call to allocate nothing (cm_workspace_bytes with all_routes; bench.py's clustered leg).
size_t cm_knn_ws_bytes_all_routes(int64_t n_ref, int64_t nq, int32_t d, int32_t k, int32_t dtype, int32_t metric) {
    size_t b = cm_knn_ws_bytes(n_ref, nq, d, k, dtype, metric);
    if (needs_exact_scan(n_ref, nq, d, k, metric) || metric == CM_METRIC_CORRELATION) return b;
    if (!(use_f16_path(d, k) || d <= 128)) return b;
    const int Kc = k + knn_slack_level(k, 1);
    const int64_t nq_pad = (nq + 511) / 512 * 512;
    if (d + cmk2::NORM_SLOTS <= 64)   // bucket-centred level: packed maps, side values (LS_CMAX buckets)
        b += cm_align(4 * (size_t)(nq + 256 * 512)) * 3 + cm_align(4 * (size_t)(nq / 512 + 300)) + cm_align(8 * (size_t)256 * d) +
             cm_align(4 * (size_t)256 * (size_t)(n_ref + 512)) + cm_align(4 * (size_t)256 * 512) * 2 + cm_align((size_t)256 * 512 * 256) + 8192 +
             cm_align(8 * (size_t)256 * (size_t)(n_ref / (32 * cmk2::BR) + 2));   // (+ the two (bucket, tile) bitmaps of cen_side_mark)
    if (nq >= prune_min_nq() && n_ref >= 65536 && ivf_fits(nq_pad, Kc + 16)) {   // bucket pruning: IVF_SB + 1 candidate lists per query, plans
        const size_t nqp = (size_t)(nq_pad + 256 * 512), nqb = nqp / 256 + 1;
        const size_t lists = nq_pad / 512 >= 1024 ? 2 : IVF_SB + 1;   // (chained pass B from four blocks per compute unit on)
        b += cm_align(4 * nqp * lists * (size_t)(Kc + 16)) + cm_align(4 * nqp * lists) + cm_align(8 * nqb * 136) +
             cm_align(8 * nqb * IVF_CSTR) + cm_align(8 * nqb) + 3 * cm_align(4 * nqb) + cm_align(16 * nqb) * (IVF_SB + 1) +
             cm_align(8 * (size_t)256 * IVF_CSTR) + 8192;
        b += 3 * cm_align(4 * (size_t)(nq + 1)) + 8192;
        if (d + cmk2::NORM_SLOTS <= 64)   // single-pass pruned stage at the bucket-centred level: plain query image + norms for the sample scan
            b += cm_align(nqp * 256) + 2 * cm_align(4 * nqp) + cm_align(4 * (nqp + 1024));
        b += cm_align(4 * (nqp + 1024)) + cm_align(8 * nqb * 136);   // dense seeds of the single-pass stage + their strided runs
    }
    return b + (b >> 6);
}

// "correlation" (scipy.spatial.distance.cdist through sklearn's brute force): x - mean(x) per row, in float64.  One wave per row.
template <typename T>
__global__ __launch_bounds__(256) void center_rows(const T *__restrict__ x, int64_t n, int d, double *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    double s = 0.0;
    for (int c = lane; c < d; c += 64) s += (double)x[i * d + c];
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) s += __shfl_xor(s, sft);
    const double m = s / (double)d;
    for (int c = lane; c < d; c += 64) out[i * d + c] = (double)x[i * d + c] - m;
}

cm_status cm_knn_run(cm_handle_t h, const void *d_ref, int64_t n_ref, const void *d_qry, int64_t nq, int32_t d,
                     int32_t dtype, int32_t k, int32_t metric, int32_t form, int64_t idx_offset, int32_t flags,
                     void *d_out_dist, int64_t *d_out_idx) {
    if (n_ref < 1 || nq < 0 || d < 1) CM_FAIL(h, CM_E_INVALID, "k-NN: empty reference or bad shape (n_ref=%lld, d=%d)", (long long)n_ref, d);
    if (k < 1 || k > n_ref)
        CM_FAIL(h, CM_E_INVALID, "Expected n_neighbors <= n_samples_fit, but n_neighbors = %d, n_samples_fit = %lld", k, (long long)n_ref);
    if (n_ref > 0x7fffff00LL) CM_FAIL(h, CM_E_UNSUPPORTED, "k-NN: more than 2^31 reference rows per shard");
    if (metric == CM_METRIC_CORRELATION) {   // cosine over float64 centred copies (cellmap.h); float64 distances out
        if (dtype != CM_F32 && dtype != CM_F64) CM_FAIL(h, CM_E_INVALID, "k-NN: unknown dtype id %d", dtype);
        double *rc = (double *)cm_ws_alloc(h, 8 * (size_t)n_ref * d), *qc = (double *)cm_ws_alloc(h, 8 * (size_t)(nq > 0 ? nq : 1) * d);
        if (!rc || !qc) CM_FAIL(h, CM_E_NOMEM, "k-NN (correlation): device workspace exhausted");
        if (dtype == CM_F32) {
            hipLaunchKernelGGL(center_rows<float>, dim3((unsigned)((n_ref + 3) / 4)), dim3(256), 0, h->stream, (const float *)d_ref, n_ref, d, rc);
            if (nq > 0) hipLaunchKernelGGL(center_rows<float>, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, h->stream, (const float *)d_qry, nq, d, qc);
        } else {
            hipLaunchKernelGGL(center_rows<double>, dim3((unsigned)((n_ref + 3) / 4)), dim3(256), 0, h->stream, (const double *)d_ref, n_ref, d, rc);
            if (nq > 0) hipLaunchKernelGGL(center_rows<double>, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, h->stream, (const double *)d_qry, nq, d, qc);
        }
        CM_HIP(h, hipGetLastError());
        return knn_run_t<double>(h, rc, n_ref, qc, nq, d, k, CM_METRIC_COSINE, form, idx_offset, flags | CM_KNN_F64_OUT, d_out_dist, d_out_idx);
    }
    if (metric == CM_METRIC_SQEUCLIDEAN) {   // sklearn: brute force only, squared distances out (knn.py:429 hands the name through)
        metric = CM_METRIC_EUCLIDEAN;
        form = CM_FORM_GEMM;
        flags |= CM_KNN_SQUARED;
    }
    const bool ratio_metric = metric == CM_METRIC_CANBERRA || metric == CM_METRIC_BRAYCURTIS;
    const bool count_metric = metric >= CM_METRIC_HAMMING && metric <= CM_METRIC_NAN_EUCLIDEAN;   // (every metric of the exact scan: the counts, haversine, nan_euclidean)
    if (metric == CM_METRIC_HAVERSINE && d != 2) CM_FAIL(h, CM_E_INVALID, "Haversine distance only valid in 2 dimensions");
    if (metric != CM_METRIC_EUCLIDEAN && metric != CM_METRIC_COSINE && metric != CM_METRIC_MANHATTAN && metric != CM_METRIC_CHEBYSHEV && !ratio_metric && !count_metric)
        CM_FAIL(h, CM_E_INVALID, "k-NN: unknown metric id %d", metric);
    if (ratio_metric && (dtype == CM_F32 || dtype == CM_F64) && nq > 0 && !needs_exact_scan(n_ref, nq, d, k, metric)) {
        // The fp32 filter of the ratio metrics divides by |x| + |y| with the hardware reciprocal, which has no denormal support, and
        // its image of float64 rows flushes magnitudes below float32's normal range: a term |x - y| / (|x| + |y|) of such
        // coordinates would come out inf / NaN and drop a true neighbour without moving the threshold (round-4 advisor finding
        // for float32 input; float64 input used to be refused).  One pass over both inputs (read by every later kernel anyway):
        // data holding a non-zero magnitude below 2^-100 takes the exact scan -- float64 pair values by the re-rank's own
        // arithmetic, no filter -- instead.
        unsigned *tiny = (unsigned *)cm_ws_alloc(h, 256);
        if (!tiny) CM_FAIL(h, CM_E_NOMEM, "k-NN: device workspace exhausted");
        CM_HIP(h, hipMemsetAsync(tiny, 0, 4, h->stream));
        if (dtype == CM_F64) {
            hipLaunchKernelGGL(tiny_nonzero<double>, dim3(1024), dim3(256), 0, h->stream, (const double *)d_ref, n_ref * (int64_t)d, tiny);
            hipLaunchKernelGGL(tiny_nonzero<double>, dim3(1024), dim3(256), 0, h->stream, (const double *)d_qry, nq * (int64_t)d, tiny);
        } else {
            hipLaunchKernelGGL(tiny_nonzero<float>, dim3(1024), dim3(256), 0, h->stream, (const float *)d_ref, n_ref * (int64_t)d, tiny);
            hipLaunchKernelGGL(tiny_nonzero<float>, dim3(1024), dim3(256), 0, h->stream, (const float *)d_qry, nq * (int64_t)d, tiny);
        }
        CM_HIP(h, hipMemcpyAsync(h->d_nflag_pinned + 14, tiny, 4, hipMemcpyDeviceToHost, h->stream));
        CM_HIP(h, hipStreamSynchronize(h->stream));
        if (h->d_nflag_pinned[14]) {
            const int fm = (d > 15 || k >= n_ref / 2) ? CM_FORM_GEMM : CM_FORM_DIRECT;
            const int f2 = form == CM_FORM_AUTO ? fm : form;
            if (dtype == CM_F32)
                return knn_run_scan<float>(h, (const float *)d_ref, n_ref, (const float *)d_qry, nq, d, k, metric, f2, idx_offset, flags, d_out_dist, d_out_idx);
            return knn_run_scan<double>(h, (const double *)d_ref, n_ref, (const double *)d_qry, nq, d, k, metric, f2, idx_offset, flags, d_out_dist, d_out_idx);
        }
    }
    if (form == CM_FORM_AUTO) form = (d > 15 || k >= n_ref / 2) ? CM_FORM_GEMM : CM_FORM_DIRECT;
    if (nq == 0) return CM_OK;
    if (dtype == CM_F32)
        return knn_run_t<float>(h, (const float *)d_ref, n_ref, (const float *)d_qry, nq, d, k, metric, form, idx_offset, flags, d_out_dist, d_out_idx);
    if (dtype == CM_F64)
        return knn_run_t<double>(h, (const double *)d_ref, n_ref, (const double *)d_qry, nq, d, k, metric, form, idx_offset, flags, d_out_dist, d_out_idx);
    CM_FAIL(h, CM_E_INVALID, "k-NN: unknown dtype id %d", dtype);
}

cm_status cm_knn_merge_run(cm_handle_t h, const double *d_dist, const int64_t *d_idx, int32_t nparts, int64_t part_stride,
                           int64_t nq, int32_t k, int32_t sqrt_out, double *d_out_dist, int64_t *d_out_idx) {
    if (nparts < 1 || k < 1 || nparts * k > MG_NCMAX) CM_FAIL(h, CM_E_INVALID, "merge: nparts*k = %d exceeds %d", nparts * k, MG_NCMAX);
    if (part_stride < nq * k) CM_FAIL(h, CM_E_INVALID, "merge: part stride %lld is smaller than nq*k = %lld", (long long)part_stride, (long long)(nq * k));
    if (nq == 0) return CM_OK;
    hipLaunchKernelGGL(knn_merge, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, h->stream, d_dist, d_idx, nparts, part_stride, nq, k,
                       sqrt_out, d_out_dist, d_out_idx);
    CM_HIP(h, hipGetLastError());
    return CM_OK;
}

// code-object preload (cm_create): one empty launch makes the runtime load this translation unit's kernels onto the
// device ahead of the first real call (the load used to land inside the first call's stage times)
static __global__ void cm_touch_knn() {}
void cm_preload_knn(hipStream_t st) { hipLaunchKernelGGL(cm_touch_knn, dim3(1), dim3(64), 0, st); }
