// K1-K3 (graph kernel -> canonical, row-normalised mapping matrix) and A0/A0n/A1/A2/A3 (transfers) on gfx950.
// These stages are HBM-bound byte/gather work: coalesced row reads, wave-per-row cooperative sorting in LDS,
// 16-byte vector loads for the wide SpMM.  No MFMA here on purpose.
//
// Reference interfaces replaced (under /root/reference/src/cellmapper/model/):
//   knn.py:68-77,166-226  validity mask + kernel values        knn.py:79-111  COO -> canonical CSR
//   cellmapper.py:99-137  row normalisation + float32 cast     cellmapper.py:589-623  one-hot vote
//   cellmapper.py:338,373,628  mapping_matrix @ dense          cellmapper.py:373  mapping_matrix @ sparse
#include "cm_internal.h"

#include <type_traits>
#include <algorithm>
#include <cstdlib>

#include <math.h>

#ifndef CM_A3_EXP
#define CM_A3_EXP 0    // timing experiments only (scripts/build_a3_variant.sh; wrong results): 2 no
                       // accumulator updates (loads kept), 8 no output stores, 16 no wait for the rows in front
#endif
#ifndef CM_A3_X1
#define CM_A3_X1 1   // slices of at most 64 entries: the branch-free gather with one entry per lane (0: the entry-per-lane gather)
#endif
#ifndef CM_A3_EPL1
#define CM_A3_EPL1 1
#endif
#ifndef CM_A3_CSTRIDE
#define CM_A3_CSTRIDE 1   // words between the chained scan's per-row entries (experiment: 8 = one cache line each)
#endif
#ifndef CM_A3_NB8
#define CM_A3_NB8 1   // neighbours per register set of the branch-free gather: 8 up to two entries per lane (-6 % on narrow payloads against 4;
                       // sixteen: +5 % again; at three entries per lane -1 % at 1 M rows of config 5 but nothing at 4 M), 4 / 2 beyond
#endif
#ifndef CM_A3_OCC4
#define CM_A3_OCC4 2   // narrow payloads (at most CM_A3_OCCN chunks per wave slice) with at most this many entries per lane: the kernel is
                       // held to 128 registers so that TWO workgroups share a CU and overlap their phases (-24 ... -33 %); 0 = off
#endif
#ifndef CM_A3_OCCN
#define CM_A3_OCCN 32   // ... up to this many 64-column chunks per wave slice (32: 16 384 output columns, 66 KB of LDS per workgroup)
#endif
#ifndef CM_A3_LBW
#define CM_A3_LBW 4    // chained scan of the wave-sliced product: 64 x this many rows in front are read per look-back step
#endif
#ifndef CM_A3_NSET
#define CM_A3_NSET 2   // register sets of the 16-byte gather (3 = two batches in flight behind the one being applied: measured no faster, 32 registers more)
#endif

namespace {

constexpr int W_KMAX = 1024;  // max neighbours per row handled by the wave-per-row kernels
constexpr int RED_BLOCKS = 512;

// ------------------------------------------------------------------ moments of the valid distances ---
__global__ __launch_bounds__(256) void moments_partial(const double *__restrict__ dist, const int64_t *__restrict__ idx,
                                                        int64_t n, const double *__restrict__ center,
                                                        double *__restrict__ partial /*[gridDim.x][3]*/) {
    const double c = center ? *center : 0.0;
    double s1 = 0.0, s2 = 0.0, cnt = 0.0;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        double v = dist[t];
        if (idx[t] != -1 && isfinite(v)) { double u = v - c; s1 += u; s2 += u * u; cnt += 1.0; }
    }
    __shared__ double sh[3][256];
    sh[0][threadIdx.x] = s1; sh[1][threadIdx.x] = s2; sh[2][threadIdx.x] = cnt;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
            sh[2][threadIdx.x] += sh[2][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partial[blockIdx.x * 3 + 0] = sh[0][0];
        partial[blockIdx.x * 3 + 1] = sh[1][0];
        partial[blockIdx.x * 3 + 2] = sh[2][0];
    }
}
__global__ void moments_final(const double *__restrict__ partial, int nb, double *__restrict__ out3) {
    if (threadIdx.x < 3) {
        double s = 0.0;
        for (int b = 0; b < nb; ++b) s += partial[b * 3 + threadIdx.x];
        out3[threadIdx.x] += s;
    }
}
__global__ void param_kernel(const double *__restrict__ m, int kernel, double *__restrict__ param) {
    if (threadIdx.x == 0) {
        double p = 0.0;
        if (kernel == CM_KERNEL_GAUSSIAN) p = m[0] / m[2];  // global mean distance (knn.py:196)
        else if (kernel == CM_KERNEL_SCARCHES) { double sd = sqrt(m[1] / m[2]); double r = 2.0 / sd; p = r * r; }  // knn.py:206-207
        if (!(m[2] > 0.0)) p = NAN;
        *param = p;
    }
}

// ------------------------------------------------------------------ K1-K3 fused, one lane group per row ----
// G = 32 (k <= 32: two rows per wave) or 64 lanes per row.  Same arithmetic as a COO -> CSR conversion followed by
// scipy's row normalisation: duplicates summed into their first occurrence in stored order, columns ascending, float64
// row sum accumulated in column order.
template <int G>
__global__ __launch_bounds__(256) void weights_rows(const double *__restrict__ dist, const int64_t *__restrict__ idx,
                                                     int64_t nq, int k, int kernel, const double *__restrict__ param_p,
                                                     double eps, int32_t *__restrict__ cols, float *__restrict__ vals,
                                                     int32_t *__restrict__ cntp, double *__restrict__ conn) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int RPB = 256 / G;   // rows per workgroup
    const int gl = threadIdx.x % G, g = threadIdx.x / G;
    // per row: wv[k] double, sw[k] double, col[k] int, scol[k] int, head[k] int
    unsigned char *base = smem + (size_t)g * ((size_t)k * (8 + 8 + 4 + 4 + 4) + 32);
    base = (unsigned char *)(((uintptr_t)base + 15) & ~(uintptr_t)15);
    double *wv = (double *)base;
    double *sw = wv + k;
    int *col = (int *)(sw + k);
    int *scol = col + k;
    int *head = scol + k;
    const int64_t i = (int64_t)blockIdx.x * RPB + g;
    const bool live = i < nq;   // (all lanes of a wave stay in the kernel: the wave barriers below are wave-wide)
    const double param = (kernel == CM_KERNEL_GAUSSIAN || kernel == CM_KERNEL_SCARCHES) ? *param_p : 0.0;
    const double two_s2 = 2.0 * (param * param);
    if (live)
        for (int t = gl; t < k; t += G) {
            double dd = dist[i * k + t];
            int64_t c = idx[i * k + t];
            bool valid = (c != -1) && isfinite(dd);
            double wt = 0.0;
            if (valid) {
                switch (kernel) {
                case CM_KERNEL_GAUSSIAN: wt = exp(-(dd * dd) / two_s2); break;
                case CM_KERNEL_SCARCHES: wt = exp(-dd / param); break;
                case CM_KERNEL_INVERSE_DISTANCE: wt = 1.0 / (dd + eps); break;
                case CM_KERNEL_IDENTITY: wt = dd; break;
                default: wt = 1.0;
                }
            }
            wv[t] = wt;
            col[t] = valid ? (int)c : -1;
        }
    __builtin_amdgcn_wave_barrier();
    // head = first occurrence of its column (duplicates are summed into it, like COO -> CSR does)
    if (live)
        for (int t = gl; t < k; t += G) {
            int c = col[t];
            int hd = c >= 0;
            for (int j = 0; j < t && hd; ++j) hd = (col[j] != c);
            head[t] = hd;
        }
    __builtin_amdgcn_wave_barrier();
    int myheads = 0;
    if (live)
        for (int t = gl; t < k; t += G) {
            if (!head[t]) continue;
            int c = col[t];
            int rk = 0;
            double sum = 0.0;
#pragma unroll 4
            for (int j = 0; j < k; ++j) {
                int cj = col[j];
                rk += (head[j] && cj < c) ? 1 : 0;
                if (cj == c) sum += wv[j];
            }
            scol[rk] = c;
            sw[rk] = sum;
            ++myheads;
        }
    int m = myheads;
    for (int o = G / 2; o > 0; o >>= 1) m += __shfl_xor(m, o);
    __builtin_amdgcn_wave_barrier();
    if (!live) return;
    // float64 row sum in CSR order (scipy csr_matvec against ones), zero -> 1, multiply by the reciprocal
    double s = 0.0;
#pragma unroll 8
    for (int r = 0; r < m; ++r) s = s + sw[r] * 1.0;
    if (s == 0.0) s = 1.0;
    const double rinv = 1.0 / s;
    for (int r = gl; r < k; r += G) {
        bool on = r < m;
        cols[i * k + r] = on ? scol[r] : 0;
        vals[i * k + r] = on ? (float)(sw[r] * rinv) : 0.f;
        if (conn) conn[i * k + r] = on ? sw[r] : 0.0;
    }
    if (gl == 0) cntp[i] = m;
}

// The same, register-resident (k <= 128 neighbours per row, column indices below 2^25): one row per group of G lanes with S
// entries per lane.  The kernel above ranks every entry by counting over the row in LDS -- O(k^2) LDS reads and a serial float64
// row sum that re-reads LDS, 8 % of the HBM roofline at k = 30 and 16 x off it at k = 90.  Here the entries are SORTED by
// (column, stored position) with a bitonic network across the lanes (keys in registers, partners by lane shuffles), so the
// output position of an entry is its element index, its weight follows by one shuffle from the lane that loaded it, and
// the float64 row sum runs over the sorted weights by lane broadcasts in CSR order -- the arithmetic of the kernel above
// (COO -> CSR conversion + scipy's row normalisation), no LDS traffic.  A row with a duplicated column (legal, rare:
// precomputed graphs) finishes on one lane through a small LDS scratch: duplicates summed in stored order into the first.
template <int G, int S>
__global__ __launch_bounds__(256) void weights_rows_sorted(const double *__restrict__ dist, const int64_t *__restrict__ idx,
                                                            int64_t nq, int k, int kernel, const double *__restrict__ param_p,
                                                            double eps, int32_t *__restrict__ cols, float *__restrict__ vals,
                                                            int32_t *__restrict__ cntp, double *__restrict__ conn) {
    constexpr int N = G * S, RPB = 256 / G;
    constexpr unsigned KINV = 0xffffffffu;
    __shared__ int s_col[RPB][N];
    __shared__ double s_w[RPB][N];
    const int lane = threadIdx.x & 63, gl = threadIdx.x % G, g = threadIdx.x / G, gbase = lane & ~(G - 1);
    const int64_t i = (int64_t)blockIdx.x * RPB + g;
    const bool live = i < nq;
    const double param = (kernel == CM_KERNEL_GAUSSIAN || kernel == CM_KERNEL_SCARCHES) ? *param_p : 0.0;
    const double two_s2 = 2.0 * (param * param);
    unsigned key[S];
    double w[S];
#pragma unroll
    for (int sl = 0; sl < S; ++sl) {
        const int e = sl * G + gl;
        key[sl] = KINV;
        w[sl] = 0.0;
        if (live && e < k) {
            const double dd = dist[i * k + e];
            const int64_t c = idx[i * k + e];
            if (c != -1 && isfinite(dd)) {
                double wt;
                switch (kernel) {
                case CM_KERNEL_GAUSSIAN: wt = exp(-(dd * dd) / two_s2); break;
                case CM_KERNEL_SCARCHES: wt = exp(-dd / param); break;
                case CM_KERNEL_INVERSE_DISTANCE: wt = 1.0 / (dd + eps); break;
                case CM_KERNEL_IDENTITY: wt = dd; break;
                default: wt = 1.0;
                }
                w[sl] = wt;
                key[sl] = ((unsigned)c << 7) | (unsigned)e;
            }
        }
    }
    // ---- bitonic sort of the N keys of a row (element index = slot * G + lane of the group), ascending
#pragma unroll
    for (int kk = 2; kk <= N; kk <<= 1) {
#pragma unroll
        for (int j = kk >> 1; j > 0; j >>= 1) {
            if (j >= G) {   // (S == 2, last merge: the partner is the lane's other slot, ascending)
                const unsigned lo = key[0] < key[S - 1] ? key[0] : key[S - 1], hi = key[0] < key[S - 1] ? key[S - 1] : key[0];
                key[0] = lo; key[S - 1] = hi;
            } else {
#pragma unroll
                for (int sl = 0; sl < S; ++sl) {
                    const unsigned other = (unsigned)__shfl_xor((int)key[sl], j, 64);
                    const int e = sl * G + gl;
                    const bool asc = (e & kk) == 0, lower = (gl & j) == 0;
                    const unsigned mn = key[sl] < other ? key[sl] : other, mx = key[sl] < other ? other : key[sl];
                    key[sl] = (lower == asc) ? mn : mx;
                }
            }
        }
    }
    // ---- the sorted entries: column, weight (from the lane and slot that loaded it), duplicates
    int col[S];
    double ws[S];
    bool valid[S];
    int m = 0;
    bool dup = false;
#pragma unroll
    for (int sl = 0; sl < S; ++sl) {
        valid[sl] = key[sl] != KINV;
        col[sl] = (int)(key[sl] >> 7);
        const int pos = (int)(key[sl] & 127u), srcl = gbase | (pos & (G - 1));
        double v = __shfl(w[0], srcl, 64);
        if (S > 1) { const double v1 = __shfl(w[S - 1], srcl, 64); v = pos >= G ? v1 : v; }
        ws[sl] = valid[sl] ? __dadd_rn(0.0, v) : 0.0;
        unsigned long long bal = __ballot(valid[sl]);
        if (G == 32) bal = (bal >> gbase) & 0xffffffffull;
        m += __popcll(bal);
    }
#pragma unroll
    for (int sl = 0; sl < S; ++sl) {
        int prev = __shfl_up(col[sl], 1, G);                     // element e - 1 (same slot)
        if (sl > 0) { const int last = __shfl(col[sl - 1], gbase | (G - 1), 64); prev = gl == 0 ? last : prev; }
        const int e = sl * G + gl;
        dup = dup || (valid[sl] && e > 0 && prev == col[sl]);
    }
    unsigned long long dbal = __ballot(dup);
    if (G == 32) dbal = (dbal >> gbase) & 0xffffffffull;
    if (dbal == 0ull) {
        // float64 row sum in CSR order (scipy csr_matvec against ones), zero -> 1, multiply by the reciprocal
        // (the sorted weights reach every lane through scalar registers: v_readlane with a uniform index, no LDS crossbar)
        auto bcast = [](double x, int l) {
            const long long b = __double_as_longlong(x);
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(b & 0xffffffffll), l);
            const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), l);
            return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
        };
        double sum = 0.0;
        if (G == 64) {
            const int m0 = m < 64 ? m : 64;
            for (int r = 0; r < m0; ++r) sum = sum + bcast(ws[0], r) * 1.0;
            if (S > 1) for (int r = 64; r < m; ++r) sum = sum + bcast(ws[S - 1], r - 64) * 1.0;
        } else {   // two rows per wave: the row of the lower half in lanes 0..31, of the upper half in lanes 32..63
            const int ma = __builtin_amdgcn_readlane(m, 0), mb = __builtin_amdgcn_readlane(m, 32);
            const int mm = ma > mb ? ma : mb;
            for (int r = 0; r < mm; ++r) {
                const double va = bcast(ws[0], r), vb = bcast(ws[0], 32 + r);
                const double v = lane < 32 ? va : vb;
                sum = r < m ? sum + v * 1.0 : sum;
            }
        }
        if (sum == 0.0) sum = 1.0;
        const double rinv = 1.0 / sum;
        if (live) {
#pragma unroll
            for (int sl = 0; sl < S; ++sl) {
                const int e = sl * G + gl;
                if (e < k) {
                    const bool on = e < m;
                    cols[i * k + e] = on ? col[sl] : 0;
                    vals[i * k + e] = on ? (float)(ws[sl] * rinv) : 0.f;
                    if (conn) conn[i * k + e] = on ? ws[sl] : 0.0;
                }
            }
            if (gl == 0) cntp[i] = m;
        }
    } else {
        // a duplicated column in this row: one lane merges the sorted run (summing in stored order into the first occurrence)
#pragma unroll
        for (int sl = 0; sl < S; ++sl) { s_col[g][sl * G + gl] = col[sl]; s_w[g][sl * G + gl] = ws[sl]; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (live && gl == 0) {
            int p = 0;
            for (int r = 0; r < m; ++r) {
                if (r > 0 && s_col[g][r] == s_col[g][p - 1]) s_w[g][p - 1] = s_w[g][p - 1] + s_w[g][r];
                else { s_col[g][p] = s_col[g][r]; s_w[g][p] = s_w[g][r]; ++p; }
            }
            double sum = 0.0;
            for (int r = 0; r < p; ++r) sum = sum + s_w[g][r] * 1.0;
            if (sum == 0.0) sum = 1.0;
            const double rinv = 1.0 / sum;
            for (int r = 0; r < k; ++r) {
                const bool on = r < p;
                cols[i * k + r] = on ? s_col[g][r] : 0;
                vals[i * k + r] = on ? (float)(s_w[g][r] * rinv) : 0.f;
                if (conn) conn[i * k + r] = on ? s_w[g][r] : 0.0;
            }
            cntp[i] = p;
        }
    }
}

// Presence scores (evaluate.py:426-521): column sums of the un-normalised gaussian connectivities, overall (column 0)
// and per query group (columns 1..G).  One thread per (query, neighbour) entry, float64 atomics into [n_ref, G+1].
__global__ void presence_scatter(const double *__restrict__ dist, const int64_t *__restrict__ idx, int64_t nq, int k,
                                 const double *__restrict__ param_p, const int32_t *__restrict__ group, int n_groups,
                                 double *__restrict__ out) {
    int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nq * k) return;
    double dd = dist[g];
    int64_t c = idx[g];
    if (c == -1 || !isfinite(dd)) return;
    const double sigma = *param_p;
    const double w = exp(-(dd * dd) / (2.0 * (sigma * sigma)));
    double *row = out + c * (int64_t)(n_groups + 1);
    atomicAdd(&row[0], w);
    if (group) {
        int gi = group[g / k];
        if (gi >= 0 && gi < n_groups) atomicAdd(&row[1 + gi], w);
    }
}

// K3 for a caller-supplied canonical CSR: one thread per row (rows are short), float64 sequential sum.
__global__ void csr_normalize_rows(const int64_t *__restrict__ indptr, const double *__restrict__ data, int64_t nq,
                                   float *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    double s = 0.0;
    for (int64_t p = indptr[i]; p < indptr[i + 1]; ++p) s = s + data[p] * 1.0;
    if (s == 0.0) s = 1.0;
    const double r = 1.0 / s;
    for (int64_t p = indptr[i]; p < indptr[i + 1]; ++p) out[p] = (float)(data[p] * r);
}

// ------------------------------------------------------------------ exclusive scan int32 -> int64 ----
constexpr int SCAN_CHUNK = 1024;
__global__ __launch_bounds__(256) void scan_block_sums(const int32_t *__restrict__ in, int64_t n, int64_t *__restrict__ bsum) {
    int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK;
    int64_t s = 0;
    for (int t = threadIdx.x; t < SCAN_CHUNK; t += 256) { int64_t p = base + t; if (p < n) s += in[p]; }
    __shared__ int64_t sh[256];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) bsum[blockIdx.x] = sh[0];
}
__global__ void scan_of_sums(int64_t *bsum, int64_t nb) {  // single thread: nb = n/1024 is small
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int64_t run = 0;
        for (int64_t b = 0; b < nb; ++b) { int64_t v = bsum[b]; bsum[b] = run; run += v; }
    }
}
__global__ __launch_bounds__(256) void scan_apply(const int32_t *__restrict__ in, int64_t n, const int64_t *__restrict__ bsum,
                                                   int64_t *__restrict__ out /*n+1*/) {
    __shared__ int64_t sh[256];
    int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK;
    int64_t v[4];
    int64_t loc = 0;
    for (int e = 0; e < 4; ++e) { int64_t p = base + threadIdx.x * 4 + e; v[e] = p < n ? in[p] : 0; loc += v[e]; }
    sh[threadIdx.x] = loc;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {  // Hillis-Steele inclusive scan of the per-thread sums
        int64_t add = (int)threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
        __syncthreads();
        sh[threadIdx.x] += add;
        __syncthreads();
    }
    int64_t run = bsum[blockIdx.x] + sh[threadIdx.x] - loc;
    for (int e = 0; e < 4; ++e) {
        int64_t p = base + threadIdx.x * 4 + e;
        if (p < n) out[p] = run;
        run += v[e];
        if (p == n - 1) out[n] = run;
    }
}
__global__ void ell_compact(const int32_t *__restrict__ cols, const float *__restrict__ vals, const double *__restrict__ conn,
                            const int32_t *__restrict__ cnt, const int64_t *__restrict__ indptr, int64_t nq, int k,
                            int32_t *__restrict__ ocols, float *__restrict__ ovals, double *__restrict__ oconn) {
    int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nq * k) return;
    int64_t i = g / k;
    int t = (int)(g - i * k);
    if (t < cnt[i]) {
        int64_t o = indptr[i] + t;
        ocols[o] = cols[g];
        if (ovals) ovals[o] = vals[g];
        if (oconn) oconn[o] = conn[g];
    }
}

// ------------------------------------------------------------------ A0: label vote, one wave per row -
// Per category float32 sums accumulated in ascending column order (scipy csr_matmat), zero sums dropped,
// argmax ties -> lowest category code (scipy sparse argmax on sorted columns), empty -> (0, 0).
// G lanes per row: 32 for padded rows of at most 32 entries (two rows per wave), else 64.
template <int G>
__global__ __launch_bounds__(256) void vote_rows(const int64_t *__restrict__ indptr, const int32_t *__restrict__ cols,
                                                  const float *__restrict__ vals, int64_t nq, int ell_k,
                                                  const int32_t *__restrict__ cnt, const int32_t *__restrict__ codes,
                                                  int32_t *__restrict__ out_code, float *__restrict__ out_conf) {
    constexpr int RPB = 256 / G, CAP = G == 32 ? 32 : W_KMAX;
    __shared__ int s_code[RPB][CAP];
    __shared__ float s_w[RPB][CAP];
    const int lane = threadIdx.x % G, w = threadIdx.x / G;
    const int64_t i = (int64_t)blockIdx.x * RPB + w;
    if (i >= nq) return;
    int64_t p0;
    int n;
    if (indptr) { p0 = indptr[i]; n = (int)(indptr[i + 1] - p0); }
    else { p0 = i * ell_k; n = cnt ? cnt[i] : ell_k; }
    float best = 0.f;
    int bcode = 0x7fffffff;
    if (n <= CAP) {
        for (int t = lane; t < n; t += G) { s_code[w][t] = codes[cols[p0 + t]]; s_w[w][t] = vals[p0 + t]; }
        __builtin_amdgcn_wave_barrier();
        for (int t = lane; t < n; t += G) {
            int c = s_code[w][t];
            bool first = true;
            for (int j = 0; j < t && first; ++j) first = (s_code[w][j] != c);
            if (!first) continue;
            float sum = 0.f;
            for (int j = t; j < n; ++j) if (s_code[w][j] == c) sum = sum + s_w[w][j] * 1.0f;
            if (sum != 0.f && (sum > best || (sum == best && c < bcode) || bcode == 0x7fffffff)) { best = sum; bcode = c; }
        }
    } else {  // long rows (user-supplied mapping matrices): same algorithm straight from global memory
        for (int t = lane; t < n; t += G) {
            int c = codes[cols[p0 + t]];
            bool first = true;
            for (int j = 0; j < t && first; ++j) first = (codes[cols[p0 + j]] != c);
            if (!first) continue;
            float sum = 0.f;
            for (int j = t; j < n; ++j) if (codes[cols[p0 + j]] == c) sum = sum + vals[p0 + j] * 1.0f;
            if (sum != 0.f && (sum > best || (sum == best && c < bcode) || bcode == 0x7fffffff)) { best = sum; bcode = c; }
        }
    }
    for (int o = G / 2; o > 0; o >>= 1) {
        float ob = __shfl_xor(best, o);
        int oc = __shfl_xor(bcode, o);
        bool take = (oc != 0x7fffffff) && (bcode == 0x7fffffff || ob > best || (ob == best && oc < bcode));
        if (take) { best = ob; bcode = oc; }
    }
    if (lane == 0) {
        bool found = bcode != 0x7fffffff && best > 0.f;
        out_code[i] = found ? bcode : 0;
        out_conf[i] = found ? best : 0.f;
    }
}

// ------------------------------------------------------------------ A1/A2: CSR x dense ---------------
// Wide rows: one workgroup per (query row, 256*VEC-column chunk); every lane owns VEC consecutive output columns,
// reads the k gathered reference rows with 16-byte loads (row index / weight are wave-uniform scalars) and
// accumulates in ascending column order, multiply and add rounded separately (scipy csr_matvecs order).
template <typename T, int VEC>
struct VecT;
template <> struct VecT<float, 4> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct VecT<float, 1> { typedef float type; };
template <> struct VecT<double, 2> { typedef double type __attribute__((ext_vector_type(2))); };
template <> struct VecT<double, 1> { typedef double type; };

template <typename T> __device__ __forceinline__ T mul_rn(T a, T b);
template <> __device__ __forceinline__ float mul_rn<float>(float a, float b) { return __fmul_rn(a, b); }
template <> __device__ __forceinline__ double mul_rn<double>(double a, double b) { return __dmul_rn(a, b); }
template <typename T> __device__ __forceinline__ T add_rn(T a, T b);
template <> __device__ __forceinline__ float add_rn<float>(float a, float b) { return __fadd_rn(a, b); }
template <> __device__ __forceinline__ double add_rn<double>(double a, double b) { return __dadd_rn(a, b); }

template <typename T, int VEC>
__global__ __launch_bounds__(256) void spmm_wide(const int64_t *__restrict__ indptr, const int32_t *__restrict__ cols,
                                                  const float *__restrict__ vals, int64_t nq, const T *__restrict__ B,
                                                  int64_t n_out, T *__restrict__ out, int64_t nchunks) {
    typedef typename VecT<T, VEC>::type V;
    const int64_t tiles = nq * nchunks;
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t row = tile / nchunks;
        const int64_t c0 = (tile - row * nchunks) * (256 * VEC) + (int64_t)threadIdx.x * VEC;
        if (c0 >= n_out) continue;
        const int64_t p0 = indptr[row], p1 = indptr[row + 1];
        T acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = (T)0;
        int64_t p = p0;
        for (; p + 4 <= p1; p += 4) {
            V x[4];
            T a[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a[u] = (T)vals[p + u];
                x[u] = *reinterpret_cast<const V *>(B + (int64_t)cols[p + u] * n_out + c0);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const T *xv = reinterpret_cast<const T *>(&x[u]);
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[v] = add_rn<T>(acc[v], mul_rn<T>(a[u], xv[v]));
            }
        }
        for (; p < p1; ++p) {
            T a = (T)vals[p];
            V x = *reinterpret_cast<const V *>(B + (int64_t)cols[p] * n_out + c0);
            const T *xv = reinterpret_cast<const T *>(&x);
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] = add_rn<T>(acc[v], mul_rn<T>(a, xv[v]));
        }
        V o;
        T *ov = reinterpret_cast<T *>(&o);
#pragma unroll
        for (int v = 0; v < VEC; ++v) ov[v] = acc[v];
        __builtin_nontemporal_store(o, reinterpret_cast<V *>(out + row * n_out + c0));
    }
}

// Narrow payloads (obsm with a few columns, numeric obs with one): one thread per output element.
template <typename T>
__global__ __launch_bounds__(256) void spmm_narrow(const int64_t *__restrict__ indptr, const int32_t *__restrict__ cols,
                                                    const float *__restrict__ vals, int64_t nq, const T *__restrict__ B,
                                                    int64_t n_out, T *__restrict__ out) {
    int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nq * n_out) return;
    int64_t row = g / n_out, c = g - row * n_out;
    T acc = (T)0;
    for (int64_t p = indptr[row]; p < indptr[row + 1]; ++p)
        acc = add_rn<T>(acc, mul_rn<T>((T)vals[p], B[(int64_t)cols[p] * n_out + c]));
    out[g] = acc;
}

// ------------------------------------------------------------------ A3: CSR x CSR --------------------
// One workgroup per output row with a dense float32 accumulator in LDS for `ct` output columns at a time (one pass when
// n_out <= ct = 32 768 columns; wider payloads are processed in column passes and the row's output is emitted pass by
// pass, so columns stay ascending).  Every output element must see the same sequence of float32 adds as scipy's
// csr_matmat: neighbour rows in ascending column order of the mapping matrix -- a workgroup barrier separates the adds of
// consecutive neighbours (entries of one payload row have distinct columns and need none).  The gather itself is software
// pipelined: neighbours are taken in batches of SPG_NB; every thread keeps the first 1 024 entries' (column, value) pairs
// of the NEXT batch's payload rows in registers, so ~SPG_NB x 8 KB per workgroup are in flight while the current batch is
// added -- the barrier per neighbour no longer exposes the load latency.  The per-neighbour metadata
// (weight, payload row bounds) sits in LDS; the next row's is prefetched into registers.  Exact zeros are dropped like
// csr_matmat does.   MODE 0: count non-zeros per row.  MODE 1: emit (column, value) pairs in ascending column order.
// MODE 2 (single pass, n_out <= ct): count AND emit -- the row's place in the output comes from a chained scan over the rows
// (decoupled look-back: `chain[r]` = (state, value), state 1 = the row's own count, 2 = the inclusive prefix; a row
// publishes its count as soon as it is known, then wave 0 sums the published counts of its predecessors back to the
// nearest inclusive prefix).  Rows are handed out by a ticket counter (`chain[nq]`), so every predecessor of a row is held
// by a workgroup that is already running: the wait always ends.  The gather is done once instead of twice.  Rows whose end
// would pass `capacity` entries are counted but not written (the caller compares out_indptr[nq] with its capacity).
// The chained scan couples the workgroups (a row waits for the slowest of the ~256 rows in flight before it); with
// `out_rows` the rows are emitted in COMPLETION order instead: one 64-bit atomic (`chain[nq + 1]`: row slot << 40 | entries)
// hands a finished row its slot j and its first entry, out_rows[j] names the row -- nobody waits for anybody.
// acc[c] += x.  Measured alternative: one ds_add_f32 per entry (same bits: an IEEE single add inside the LDS unit) instead of
// load / v_add / store -- 2.5 x SLOWER on gfx950 (437 -> 1 097 ms at 1 M cells of config 5: LDS float atomics serialise),
// so the three-instruction form stays; -DCM_SPG_LDS_ATOMIC builds the other one.
#ifdef CM_SPG_LDS_ATOMIC
#define SPG_ADD(P, X) ((void)__hip_atomic_fetch_add((P), (X), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
#else
#define SPG_ADD(P, X) (*(P) = __fadd_rn(*(P), (X)))
#endif
constexpr int SPG_NB = 8;     // neighbours per batch (two batches of (column, value) x 2 chunks live in registers)
constexpr int SPG_MK = 512;   // neighbours per metadata chunk (one per thread)
template <int MODE>
__global__ __launch_bounds__(512) void spgemm_rows(const int64_t *__restrict__ indptr, const int32_t *__restrict__ cols,
                                                    const float *__restrict__ vals, int64_t nq,
                                                    const int64_t *__restrict__ bptr, const int32_t *__restrict__ bcols,
                                                    const float *__restrict__ bvals, int n_out, int ct,
                                                    int32_t *__restrict__ row_nnz, int64_t *out_indptr,
                                                    int64_t out_base, int32_t *__restrict__ ocols, float *__restrict__ ovals,
                                                    int dbg, const int32_t *__restrict__ order, unsigned long long *chain,
                                                    int64_t capacity, int32_t *__restrict__ out_rows) {
    // order != nullptr: output row r is the product row of mapping-matrix row order[r] (indptr is then the FULL array):
    // a processing order in which consecutive rows share neighbours keeps the gathered payload rows in L2.
    extern __shared__ __align__(16) unsigned char smem[];
    float *acc = reinterpret_cast<float *>(smem);                       // [ct] (ct is a multiple of 4)
    int64_t *m_b0 = reinterpret_cast<int64_t *>(acc + ct);              // [SPG_MK] first entry of the neighbour's payload row
    int *m_len = reinterpret_cast<int *>(m_b0 + SPG_MK);                // [SPG_MK] its length
    float *m_a = reinterpret_cast<float *>(m_len + SPG_MK);             // [SPG_MK] the neighbour's weight
    int *wsum = reinterpret_cast<int *>(m_a + SPG_MK);                  // [2][8] wave totals (by pass parity)
    long long *tk = reinterpret_cast<long long *>(wsum + 16);           // MODE 2: [0] next ticket, [1] the row's first entry
    float *dummy = reinterpret_cast<float *>(tk + 2);                   // [128] where lanes without an entry do their add
    const int nt = blockDim.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
    int parity = 0;
    const int dix = (int)(dummy - acc) + lane;   // the lane's own slot, as an accumulator index
    auto fetch_meta = [&](int64_t row, int64_t off, float &ma, int64_t &mb0, int &mlen) {
        ma = 0.f; mb0 = 0; mlen = 0;
        if (row < nq) {
            const int64_t src = order ? (int64_t)order[row] : row;
            const int64_t p = indptr[src] + off + tid;
            if (p < indptr[src + 1]) {
                const int64_t j = cols[p];
                ma = vals[p];
                mb0 = bptr[j];
                mlen = (int)(bptr[j + 1] - mb0);
            }
        }
    };
    float pre_a; int64_t pre_b0; int pre_len;      // first metadata chunk of the next row (in flight during this row)
    int64_t pre_deg = 0, pre_out = 0;              // ... its degree and (MODE 1) the start of its output
    auto fetch_row = [&](int64_t row) {
        fetch_meta(row, 0, pre_a, pre_b0, pre_len);
        const int64_t src = (order && row < nq) ? (int64_t)order[row] : row;
        pre_deg = row < nq ? indptr[src + 1] - indptr[src] : 0;
        pre_out = (MODE == 1 && row < nq) ? out_indptr[row] : 0;
    };
    int64_t row = blockIdx.x;
    // tickets only where a row waits for its predecessors (chained scan); completion order takes the static stride
    const bool ticketed = MODE == 2 && !out_rows;
    if (ticketed) {
        if (tid == 0) tk[0] = (long long)atomicAdd(chain + nq, 1ull);
        __syncthreads();
        row = tk[0];
        __syncthreads();
    }
    fetch_row(row);
    while (row < nq) {
        const int64_t deg = pre_deg;
        int64_t row_out = pre_out;
        const float f_a = pre_a; const int64_t f_b0 = pre_b0; const int f_len = pre_len;
        int64_t next_row = row + gridDim.x;
        if (ticketed) { if (tid == 0) tk[0] = (long long)atomicAdd(chain + nq, 1ull); }   // (read after the first barrier below)
        else fetch_row(next_row);
        int total = 0;   // non-zeros of this row emitted by earlier column passes
        for (int c0 = 0; c0 < n_out; c0 += ct) {
            const int cw = (n_out - c0 < ct) ? n_out - c0 : ct;
            {   // clear the accumulator (16 bytes per store; acc holds ct floats, ct a multiple of 4 >= cw)
                float4 *a4 = reinterpret_cast<float4 *>(acc);
                for (int c = tid; c < (cw + 3) / 4; c += nt) a4[c] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            for (int64_t off = 0; off < deg || off == 0; off += SPG_MK) {
                float ma; int64_t mb0; int mlen;
                if (off == 0) { ma = f_a; mb0 = f_b0; mlen = f_len; }
                else fetch_meta(row, off, ma, mb0, mlen);
                __syncthreads();   // previous chunk's items are done with the metadata (and: acc is zeroed)
                m_a[tid] = ma; m_b0[tid] = mb0; m_len[tid] = mlen;
                if (ticketed && off == 0) { next_row = tk[0]; fetch_row(next_row); }   // the next row's metadata is in flight from here
                __syncthreads();
                const int nchunk = (int)(deg - off < SPG_MK ? deg - off : SPG_MK);
                // ---- batches of SPG_NB neighbours, two register sets: the loads of batch i+1 are in flight while batch i
                // is added.  Everything per neighbour is straight-line code on uniform (scalar) row bounds: ~25 instructions
                // per wave and neighbour (the first version spent ~100, mostly on branches and register shuffling around
                // predicated loads -- the kernel is issue-bound, not bandwidth-bound).
                int ca[SPG_NB][2], cbv[SPG_NB][2];
                float va[SPG_NB][2], vb[SPG_NB][2];
                // metadata of a batch: lane l holds neighbour ub + (l & 7) (one LDS round trip per batch); a neighbour's values
                // reach the scalar registers by v_readlane
                struct BMeta { int len; int blo, bhi; float a; };
                auto batch_meta = [&](int ub) {
                    BMeta m;
                    const int u = ub + (lane & (SPG_NB - 1));
                    const bool in = u < nchunk;
                    const int64_t b = in ? m_b0[u] : 0;
                    m.len = in ? m_len[u] : 0;
                    m.blo = (int)(b & 0xffffffffll);
                    m.bhi = (int)(b >> 32);
                    m.a = in ? m_a[u] : 0.f;
                    return m;
                };
                auto uni = [&](const BMeta &m, int nidx, int64_t &b0, int &len, float &wa) {
                    len = __builtin_amdgcn_readlane(m.len, nidx);
                    b0 = ((int64_t)__builtin_amdgcn_readlane(m.bhi, nidx) << 32) | (uint32_t)__builtin_amdgcn_readlane(m.blo, nidx);
                    wa = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m.a), nidx));
                };
                auto load_batch = [&](int (&cc)[SPG_NB][2], float (&vv)[SPG_NB][2], const BMeta &bm) {
                    // unconditional loads: uniform row base + a lane index clamped to the row (an empty slot reads the first
                    // words of indptr instead -- always there); nothing here needs a loaded value, so all of a batch's loads
                    // are issued back to back
#pragma unroll
                    for (int n = 0; n < SPG_NB; ++n) {
                        int64_t b0; int len; float wa;
                        uni(bm, n, b0, len, wa);
                        const bool has = len > 0;
                        const int32_t *pc = has ? bcols + b0 : reinterpret_cast<const int32_t *>(indptr);
                        const float *pv = has ? bvals + b0 : reinterpret_cast<const float *>(indptr);
                        const int last = has ? len - 1 : 0;
                        // (uniform base + 32-bit byte offset: the scalar-base form of global_load, no 64-bit vector arithmetic)
                        const unsigned o_0 = (unsigned)min(tid, last) << 2, o_1 = (unsigned)min(tid + 512, last) << 2;
                        cc[n][0] = *reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(pc) + o_0);
                        vv[n][0] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(pv) + o_0);
                        cc[n][1] = *reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(pc) + o_1);
                        vv[n][1] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(pv) + o_1);
                    }
                };
                auto apply_batch = [&](int (&cc)[SPG_NB][2], float (&vv)[SPG_NB][2], const BMeta &bm) {
#pragma unroll
                    for (int n = 0; n < SPG_NB; ++n) {
                        int64_t b0; int len; float wa;
                        uni(bm, n, b0, len, wa);
                        if (len > 0) {   // (uniform)
                            // the previous neighbour's adds are done.  (Not __syncthreads(): its fence would also wait
                            // for the prefetched loads of the next batch -- only LDS traffic has to be ordered here.)
                            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                            {   // the two entries of a thread lie in one payload row: distinct columns, so both accumulator
                                // reads go out before either write (one LDS round trip per neighbour instead of two).  A lane
                                // without an entry (or with a column outside the pass) works on a slot of its own behind the
                                // accumulator: no branch
                                const int c_0 = cc[n][0] - c0, c_1 = cc[n][1] - c0;
                                const bool p_0 = tid < len && (unsigned)c_0 < (unsigned)cw;
                                const bool p_1 = 512 + tid < len && (unsigned)c_1 < (unsigned)cw;
                                float *a_0 = acc + (p_0 ? c_0 : dix), *a_1 = acc + (p_1 ? c_1 : dix + 64);
                                const float x_0 = *a_0, x_1 = *a_1;
                                *a_0 = __fadd_rn(x_0, __fmul_rn(wa, vv[n][0]));
                                *a_1 = __fadd_rn(x_1, __fmul_rn(wa, vv[n][1]));
                            }
                            if (len > 1024) {   // (uniform) payload rows longer than 1 024 entries
                                for (int64_t t = 1024 + tid; t < len; t += 512) {
                                    const int c = bcols[b0 + t] - c0;
                                    if ((unsigned)c < (unsigned)cw) SPG_ADD(acc + c, __fmul_rn(wa, bvals[b0 + t]));
                                }
                            }
                        }
                    }
                };
                BMeta ma_ = batch_meta(0), mb_;
                load_batch(ca, va, ma_);
                for (int ub = 0; ub < nchunk; ub += 2 * SPG_NB) {
                    mb_ = batch_meta(ub + SPG_NB);
                    load_batch(cbv, vb, mb_);
                    apply_batch(ca, va, ma_);
                    ma_ = batch_meta(ub + 2 * SPG_NB);
                    load_batch(ca, va, ma_);
                    apply_batch(cbv, vb, mb_);
                }
            }
            __syncthreads();   // all adds of the pass are done
            // ordered compaction: every wave owns a contiguous slice of the pass's columns and walks it 64 columns at a
            // time (ballot + prefix popcount): consecutive lanes write consecutive output entries
            const int cwr = (((cw + nw - 1) / nw) + 63) & ~63;
            const int lo = wave * cwr, hi = (lo + cwr < cw) ? lo + cwr : cw;
            int wcount = 0;
            for (int c = lo; c < hi; c += 64) {
                const bool nz = (c + lane < hi) && acc[c + lane] != 0.f;
                wcount += __popcll(__ballot(nz));
            }
            if (lane == 0) wsum[parity * 8 + wave] = wcount;
            __syncthreads();
            int before = 0, all = 0;
            for (int w = 0; w < nw; ++w) { const int s_ = wsum[parity * 8 + w]; all += s_; if (w < wave) before += s_; }
            parity ^= 1;
            bool fits = true;
            if (MODE == 2 && out_rows) {
                if (tid == 0) {
                    const unsigned long long old = atomicAdd(chain + nq + 1, (1ull << 40) | (unsigned long long)all);
                    const long long j = (long long)(old >> 40), o = (long long)(old & ((1ull << 40) - 1ull));
                    out_indptr[j] = o;
                    if (j == nq - 1) out_indptr[nq] = o + all;
                    out_rows[j] = (int32_t)row;
                    tk[1] = o;
                }
                __syncthreads();
                row_out = tk[1];
                fits = row_out + all <= capacity;
            } else if (MODE == 2) {
                constexpr unsigned long long VMASK = (1ull << 62) - 1ull;
                if (wave == 0) {
                    if (lane == 0)
                        __hip_atomic_store(chain + row, ((row == 0 ? 2ull : 1ull) << 62) | (unsigned long long)all, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                    long long excl = 0;
                    for (int64_t pos = (CM_A3_EXP & 16) ? -1 : row - 1; pos >= 0;) {   // 64 predecessors at a time, nearest on lane 0
                        const int64_t i = pos - lane;
                        unsigned long long v, empty, incl;
                        for (;;) {
                            v = i >= 0 ? __hip_atomic_load(chain + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (2ull << 62);
                            empty = __ballot((v >> 62) == 0ull);
                            incl = __ballot((v >> 62) == 2ull);
                            // usable when the nearest inclusive prefix has no unpublished row in front of it, or all 64 are counts
                            if (empty == 0ull || (incl != 0ull && __ffsll((long long)incl) < __ffsll((long long)empty))) break;
                            __builtin_amdgcn_s_sleep(2);
                        }
                        const int fi = incl ? __ffsll((long long)incl) - 1 : 63;
                        long long part = lane <= fi ? (long long)(v & VMASK) : 0ll;
#pragma unroll
                        for (int sft = 32; sft > 0; sft >>= 1) part += __shfl_xor(part, sft);
                        excl += part;
                        if (incl) break;
                        pos -= 64;
                    }
                    if (lane == 0) {
                        if (row > 0)
                            __hip_atomic_store(chain + row, (2ull << 62) | (unsigned long long)(excl + all), __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                        else out_indptr[0] = 0;
                        out_indptr[row + 1] = excl + all;
                        tk[1] = excl;
                    }
                }
                __syncthreads();
                row_out = tk[1];
                fits = row_out + all <= capacity;
            }
            if (MODE >= 1 && fits && !(dbg & 8)) {
                int64_t o = row_out - out_base + total + before;
                for (int c = lo; c < hi; c += 64) {
                    const float v = (c + lane < hi) ? acc[c + lane] : 0.f;
                    const bool nz = v != 0.f;
                    const unsigned long long m = __ballot(nz);
                    if (nz) {
                        const int64_t q = o + __popcll(m & ((1ull << lane) - 1ull));
                        ocols[q] = c0 + c + lane;
                        ovals[q] = v;
                    }
                    o += __popcll(m);
                }
            }
            total += all;
            __syncthreads();   // the accumulator is free for the next pass / row
        }
        if (MODE == 0 && tid == 0) row_nnz[row] = total;
        row = next_row;
    }
}

}  // namespace

// ======================================================================= host-side launchers ==========
namespace {
__global__ void widen_f32(const float *__restrict__ in, int64_t n, double *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = (double)in[i];
}
}  // namespace
cm_status cm_widen_f32_run(cm_handle_t h, const float *d_in, int64_t n, double *d_out) {
    if (n == 0) return CM_OK;
    hipLaunchKernelGGL(widen_f32, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 65536)), dim3(256), 0, h->stream, d_in, n, d_out);
    CM_HIP(h, hipGetLastError());
    return CM_OK;
}

cm_status cm_moments_run(cm_handle_t h, const double *d_dist, const int64_t *d_idx, int64_t n, const double *d_center,
                         double *d_out3) {
    double *partial = (double *)cm_ws_alloc(h, sizeof(double) * 3 * RED_BLOCKS);
    if (!partial) CM_FAIL(h, CM_E_NOMEM, "moments: device workspace exhausted");
    hipLaunchKernelGGL(moments_partial, dim3(RED_BLOCKS), dim3(256), 0, h->stream, d_dist, d_idx, n, d_center, partial);
    hipLaunchKernelGGL(moments_final, dim3(1), dim3(64), 0, h->stream, partial, RED_BLOCKS, d_out3);
    CM_HIP(h, hipGetLastError());
    return CM_OK;
}

cm_status cm_param_run(cm_handle_t h, const double *d_m3, int32_t kernel, double *d_param) {
    hipLaunchKernelGGL(param_kernel, dim3(1), dim3(64), 0, h->stream, d_m3, kernel, d_param);
    CM_HIP(h, hipGetLastError());
    return CM_OK;
}

cm_status cm_weights_run(cm_handle_t h, const double *d_dist, const int64_t *d_idx, int64_t nq, int32_t k,
                         int64_t n_ref, int32_t kernel, const double *d_param, double eps, int32_t *d_cols,
                         float *d_vals, int32_t *d_cnt, double *d_conn) {
    if (k < 1 || k > W_KMAX) CM_FAIL(h, CM_E_UNSUPPORTED, "mapping matrix: %d neighbours per row exceeds the supported %d", k, W_KMAX);
    if (kernel < 0 || kernel > CM_KERNEL_IDENTITY) CM_FAIL(h, CM_E_INVALID, "Unknown kernel id %d", kernel);
    if (nq == 0) return CM_OK;
    // (n_ref <= 0: unknown column range.  STRICTLY below 2^25: the sorted kernel's key is (column << 7) | position with 0xffffffff as
    // the invalid sentinel, which column 2^25 - 1 at position 127 would collide with)
    const bool reg_ok = n_ref > 0 && n_ref < (1LL << 25) && !cm_get_knobs().weights_lds;
    if (reg_ok && k <= 128) {   // register-resident sort: one row per 32 lanes, per wave, or per wave with two entries per lane
        if (k <= 32)
            hipLaunchKernelGGL((weights_rows_sorted<32, 1>), dim3((unsigned)((nq + 7) / 8)), dim3(256), 0, h->stream, d_dist, d_idx, nq, k,
                               kernel, d_param, eps, d_cols, d_vals, d_cnt, d_conn);
        else if (k <= 64)
            hipLaunchKernelGGL((weights_rows_sorted<64, 1>), dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, h->stream, d_dist, d_idx, nq, k,
                               kernel, d_param, eps, d_cols, d_vals, d_cnt, d_conn);
        else
            hipLaunchKernelGGL((weights_rows_sorted<64, 2>), dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, h->stream, d_dist, d_idx, nq, k,
                               kernel, d_param, eps, d_cols, d_vals, d_cnt, d_conn);
    } else if (k <= 32) {   // two rows per wave
        const size_t lds = 8 * ((size_t)k * 28 + 32) + 16;
        hipLaunchKernelGGL(weights_rows<32>, dim3((unsigned)((nq + 7) / 8)), dim3(256), lds, h->stream, d_dist, d_idx, nq, k, kernel,
                           d_param, eps, d_cols, d_vals, d_cnt, d_conn);
    } else {
        const size_t lds = 4 * ((size_t)k * 28 + 32) + 16;
        hipLaunchKernelGGL(weights_rows<64>, dim3((unsigned)((nq + 3) / 4)), dim3(256), lds, h->stream, d_dist, d_idx, nq, k, kernel,
                           d_param, eps, d_cols, d_vals, d_cnt, d_conn);
    }
    CM_HIP(h, hipGetLastError());
    return CM_OK;
}

cm_status cm_presence_run(cm_handle_t h, const double *d_dist, const int64_t *d_idx, int64_t nq, int32_t k,
                          const double *d_param, const int32_t *d_group, int32_t n_groups, double *d_out) {
    if (nq == 0) return CM_OK;
    int64_t tot = nq * k;
    hipLaunchKernelGGL(presence_scatter, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, d_dist, d_idx, nq, k,
                       d_param, d_group, n_groups, d_out);
    CM_HIP(h, hipGetLastError());
    return CM_OK;
}

cm_status cm_csr_normalize_run(cm_handle_t h, const int64_t *d_indptr, const double *d_data, int64_t nq, float *d_out) {
    if (nq == 0) return CM_OK;
    hipLaunchKernelGGL(csr_normalize_rows, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, h->stream, d_indptr, d_data, nq, d_out);
    CM_HIP(h, hipGetLastError());
    return CM_OK;
}

size_t cm_scan_ws_bytes(int64_t n) { return cm_align(sizeof(int64_t) * ((n + SCAN_CHUNK - 1) / SCAN_CHUNK + 1)); }

cm_status cm_exclusive_scan_i32_to_i64(cm_handle_t h, const int32_t *d_in, int64_t n, int64_t *d_out) {
    if (n == 0) { CM_HIP(h, hipMemsetAsync(d_out, 0, sizeof(int64_t), h->stream)); return CM_OK; }
    int64_t nb = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
    int64_t *bsum = (int64_t *)cm_ws_alloc(h, sizeof(int64_t) * (nb + 1));
    if (!bsum) CM_FAIL(h, CM_E_NOMEM, "scan: device workspace exhausted");
    hipLaunchKernelGGL(scan_block_sums, dim3((unsigned)nb), dim3(256), 0, h->stream, d_in, n, bsum);
    hipLaunchKernelGGL(scan_of_sums, dim3(1), dim3(64), 0, h->stream, bsum, nb);
    hipLaunchKernelGGL(scan_apply, dim3((unsigned)nb), dim3(256), 0, h->stream, d_in, n, bsum, d_out);
    CM_HIP(h, hipGetLastError());
    return CM_OK;
}

cm_status cm_ell_to_csr_run(cm_handle_t h, const int32_t *d_cols, const float *d_vals, const double *d_conn,
                            const int32_t *d_cnt, int64_t nq, int32_t k, int64_t *d_indptr, int32_t *d_out_cols,
                            float *d_out_vals, double *d_out_conn) {
    CM_TRY(cm_exclusive_scan_i32_to_i64(h, d_cnt, nq, d_indptr));
    if (nq == 0) return CM_OK;
    int64_t tot = nq * k;
    hipLaunchKernelGGL(ell_compact, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, d_cols, d_vals, d_conn, d_cnt,
                       d_indptr, nq, k, d_out_cols, d_out_vals, d_out_conn);
    CM_HIP(h, hipGetLastError());
    return CM_OK;
}

cm_status cm_vote_run(cm_handle_t h, const int64_t *d_indptr, const int32_t *d_cols, const float *d_vals, int64_t nq,
                      int32_t ell_k, const int32_t *d_cnt, const int32_t *d_codes, int32_t n_cat, int32_t *d_out_code,
                      float *d_out_conf) {
    if (n_cat < 1) CM_FAIL(h, CM_E_INVALID, "label vote: n_cat must be >= 1");
    if (nq == 0) return CM_OK;
    // rows of at most 32 entries (padded rows of the k-NN path, or a CSR mapping matrix built from a k <= 32 graph: ell_k is
    // then only the hint): two rows per wave.  Longer CSR rows are still handled by that kernel's global-memory path.
    if (ell_k > 0 && ell_k <= 32)
        hipLaunchKernelGGL(vote_rows<32>, dim3((unsigned)((nq + 7) / 8)), dim3(256), 0, h->stream, d_indptr, d_cols, d_vals, nq, ell_k,
                           d_cnt, d_codes, d_out_code, d_out_conf);
    else
        hipLaunchKernelGGL(vote_rows<64>, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, h->stream, d_indptr, d_cols, d_vals, nq, ell_k,
                           d_cnt, d_codes, d_out_code, d_out_conf);
    CM_HIP(h, hipGetLastError());
    return CM_OK;
}

template <typename T>
static cm_status spmm_dense_t(cm_handle_t h, const int64_t *d_indptr, const int32_t *d_cols, const float *d_vals, int64_t nq,
                              const T *B, int64_t n_out, T *out) {
    constexpr int VEC = 16 / sizeof(T);
    hipStream_t st = h->stream;
    bool aligned = (n_out % VEC == 0) && ((uintptr_t)B % 16 == 0) && ((uintptr_t)out % 16 == 0);
    if (n_out >= 64 && aligned) {
        int64_t nchunks = (n_out + 256 * VEC - 1) / (256 * VEC);
        int64_t tiles = nq * nchunks;
        unsigned grid = (unsigned)(tiles < 256 * 32 ? tiles : 256 * 32);
        hipLaunchKernelGGL((spmm_wide<T, VEC>), dim3(grid), dim3(256), 0, st, d_indptr, d_cols, d_vals, nq, B, n_out, out, nchunks);
    } else if (n_out >= 256) {
        int64_t nchunks = (n_out + 255) / 256;
        int64_t tiles = nq * nchunks;
        unsigned grid = (unsigned)(tiles < 256 * 32 ? tiles : 256 * 32);
        hipLaunchKernelGGL((spmm_wide<T, 1>), dim3(grid), dim3(256), 0, st, d_indptr, d_cols, d_vals, nq, B, n_out, out, nchunks);
    } else {
        int64_t tot = nq * n_out;
        hipLaunchKernelGGL(spmm_narrow<T>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, d_indptr, d_cols, d_vals, nq, B, n_out, out);
    }
    CM_HIP(h, hipGetLastError());
    return CM_OK;
}

cm_status cm_spmm_dense_run(cm_handle_t h, const int64_t *d_indptr, const int32_t *d_cols, const float *d_vals,
                            int64_t nq, const void *d_B, int64_t n_out, int32_t b_dtype, void *d_out) {
    if (n_out < 1) CM_FAIL(h, CM_E_INVALID, "spmm: payload has no columns");
    if (nq == 0) return CM_OK;
    if (nq * n_out / 256 > 0x7fffffffLL) CM_FAIL(h, CM_E_UNSUPPORTED, "spmm: output too large for one launch; tile the call");
    if (b_dtype == CM_F32) return spmm_dense_t<float>(h, d_indptr, d_cols, d_vals, nq, (const float *)d_B, n_out, (float *)d_out);
    if (b_dtype == CM_F64) return spmm_dense_t<double>(h, d_indptr, d_cols, d_vals, nq, (const double *)d_B, n_out, (double *)d_out);
    CM_FAIL(h, CM_E_INVALID, "spmm: unknown dtype id %d", b_dtype);
}

constexpr int SPG_CT = 32768;   // output columns per pass of the A3 accumulator (128 KB of LDS)

static cm_status spgemm_launch(cm_handle_t h, int mode, const int64_t *d_indptr, const int32_t *d_cols, const float *d_vals,
                               int64_t nq, const int64_t *d_bptr, const int32_t *d_bcols, const float *d_bvals, int64_t n_out,
                               int32_t *d_row_nnz, const int64_t *d_out_indptr, int64_t out_base, int32_t *d_out_cols,
                               float *d_out_vals, const int32_t *d_order = nullptr, unsigned long long *d_chain = nullptr,
                               int64_t capacity = 0, int32_t *d_out_rows = nullptr) {
    if (n_out < 1) CM_FAIL(h, CM_E_INVALID, "spmm_csr: payload has no columns");
    if (n_out > 0x7fffffffLL) CM_FAIL(h, CM_E_UNSUPPORTED, "spmm_csr: more than 2^31 output columns");
    if (nq == 0) return CM_OK;
    int ct = SPG_CT;
    { const int v = cm_get_knobs().spgemm_ct; if (v >= 4 && v <= SPG_CT) ct = v & ~3; }   // testing: force column passes
    if (n_out < ct) ct = (int)((n_out + 3) & ~3LL);
    const int dbg = cm_get_knobs().spgemm_dbg;   // timing experiments only (wrong results)
    size_t lds = (size_t)ct * 4 + (size_t)SPG_MK * 16 + 128 + 512;
    if (h->n_cu <= 0) {
        hipDeviceProp_t pr;
        CM_HIP(h, hipGetDeviceProperties(&pr, h->device));
        h->n_cu = pr.multiProcessorCount;
    }
    // resident workgroups: the accumulator decides how many fit a CU's 160 KB of LDS
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(4, (size_t)(160 * 1024) / (lds + 1024)));
    unsigned grid = (unsigned)std::min<int64_t>(nq, (int64_t)h->n_cu * per_cu);
    int64_t *optr = const_cast<int64_t *>(d_out_indptr);   // (written by the single-pass mode only)
    if (mode == 2 && (n_out > ct || !d_chain)) CM_FAIL(h, CM_E_INVALID, "spmm_csr: the single-pass product needs n_out <= %d", ct);
    if (mode == 0) {
        auto kern = spgemm_rows<0>;
        CM_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, h->stream, d_indptr, d_cols, d_vals, nq, d_bptr, d_bcols, d_bvals,
                           (int)n_out, ct, d_row_nnz, optr, out_base, d_out_cols, d_out_vals, dbg, d_order, d_chain, capacity, d_out_rows);
    } else if (mode == 1) {
        auto kern = spgemm_rows<1>;
        CM_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, h->stream, d_indptr, d_cols, d_vals, nq, d_bptr, d_bcols, d_bvals,
                           (int)n_out, ct, d_row_nnz, optr, out_base, d_out_cols, d_out_vals, dbg, d_order, d_chain, capacity, d_out_rows);
    } else {
        auto kern = spgemm_rows<2>;
        CM_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (d_out_rows && nq >= (1LL << 24)) CM_FAIL(h, CM_E_UNSUPPORTED, "spmm_csr: rows in completion order: at most 2^24 - 1 rows per call");
        CM_HIP(h, hipMemsetAsync(d_chain, 0, 8 * (size_t)(nq + 2), h->stream));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, h->stream, d_indptr, d_cols, d_vals, nq, d_bptr, d_bcols, d_bvals,
                           (int)n_out, ct, d_row_nnz, optr, out_base, d_out_cols, d_out_vals, dbg, d_order, d_chain, capacity, d_out_rows);
    }
    CM_HIP(h, hipGetLastError());
    return CM_OK;
}

cm_status cm_spmm_csr_count_run(cm_handle_t h, const int64_t *d_indptr, const int32_t *d_cols, const float *d_vals,
                                int64_t nq, const int64_t *d_bptr, const int32_t *d_bcols, const float *d_bvals,
                                int64_t n_out, int32_t *d_row_nnz, const int32_t *d_order) {
    return spgemm_launch(h, 0, d_indptr, d_cols, d_vals, nq, d_bptr, d_bcols, d_bvals, n_out, d_row_nnz, nullptr, 0, nullptr, nullptr, d_order);
}
// out_base: value subtracted from d_out_indptr[row] to address d_out_cols / d_out_vals (query tiles of a larger output:
// the tile's buffers start at the tile's first entry)
cm_status cm_spmm_csr_fill_run(cm_handle_t h, const int64_t *d_indptr, const int32_t *d_cols, const float *d_vals,
                               int64_t nq, const int64_t *d_bptr, const int32_t *d_bcols, const float *d_bvals,
                               int64_t n_out, const int64_t *d_out_indptr, int64_t out_base, int32_t *d_out_cols, float *d_out_vals,
                               const int32_t *d_order) {
    return spgemm_launch(h, 1, d_indptr, d_cols, d_vals, nq, d_bptr, d_bcols, d_bvals, n_out, nullptr, d_out_indptr, out_base, d_out_cols, d_out_vals,
                         d_order);
}

// Single pass (n_out <= SPG_CT): counts, output offsets (chained scan, or completion order with d_out_rows) and entries from
// ONE gather.  d_chain: nq + 2 words of scratch.  d_out_indptr[0..nq] is written in full; entries of rows that end beyond
// `capacity` are not.
bool cm_spmm_csr_single_pass_ok(int64_t n_out) { return n_out >= 1 && n_out <= SPG_CT && cm_get_knobs().spgemm_ct == 0; }
cm_status cm_spmm_csr_fused_run(cm_handle_t h, const int64_t *d_indptr, const int32_t *d_cols, const float *d_vals,
                                int64_t nq, const int64_t *d_bptr, const int32_t *d_bcols, const float *d_bvals,
                                int64_t n_out, int64_t *d_out_indptr, int32_t *d_out_cols, float *d_out_vals, int64_t capacity,
                                const int32_t *d_order, unsigned long long *d_chain, int32_t *d_out_rows) {
    if (nq == 0) { CM_HIP(h, hipMemsetAsync(d_out_indptr, 0, 8, h->stream)); return CM_OK; }
    return spgemm_launch(h, 2, d_indptr, d_cols, d_vals, nq, d_bptr, d_bcols, d_bvals, n_out, nullptr, d_out_indptr, 0, d_out_cols,
                         d_out_vals, d_order, d_chain, capacity, d_out_rows);
}

// ============================================================ A3, wave-sliced single pass ===============
// The barrier per neighbour of spgemm_rows exists because two neighbours' entries of one output column are added by
// different waves.  With a PAYLOAD INDEX -- for every payload row, the positions where its (ascending) columns cross the
// multiples of W -- wave w of a workgroup takes from every neighbour row exactly the entries of columns [w W, (w+1) W): an
// output column belongs to ONE wave, whose LDS operations are executed in program order, so the adds of a column happen in
// neighbour order (csr_matmat's sequence) with no synchronisation at all.  The waves run the whole gather independently
// (each keeps 2 x 8 neighbours' entries in flight in registers) and meet twice per row: for the row's entry count (sum of
// the waves' counts) and for its place in the output (chained scan or completion order, as in spgemm_rows<2>).  A wave
// fetches its own neighbour metadata (weight, slice start, slice length: lane l <-> neighbours l and l + 64), the next
// row's in four stages spread over the current row.
namespace {
// idx[j][b], b = 0..nb: number of entries of payload row j with column < b W.  One wave per row; also checks that the
// row's columns are strictly ascending and inside [0, n_out) (flag[0] = 1 otherwise).
constexpr int A3_HDR = 8;   // header words of a payload index: [0] longest slice, [1] bucket width, [2..3] non-zeros (int64), rest 0
__global__ __launch_bounds__(256) void payload_index(const int64_t *__restrict__ bptr, const int32_t *__restrict__ bcols, int64_t n_ref,
                                                     int n_out, int W, int nb, int32_t *__restrict__ idx, int *__restrict__ flag) {
    const int lane = threadIdx.x & 63;
    const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= n_ref) return;
    const int64_t lo = bptr[j], hi = bptr[j + 1];
    int32_t *out = idx + A3_HDR + j * (nb + 1);
    const int len = (int)(hi - lo);
    if (hi < lo || hi - lo > 0x7fffffffLL) { if (lane == 0) flag[0] = 1; return; }
    if (len == 0) { for (int b = lane; b <= nb; b += 64) out[b] = 0; return; }
    bool bad = false;
    for (int i = lane; i < len; i += 64) {
        const int c = bcols[lo + i];
        const int cn = i + 1 < len ? bcols[lo + i + 1] : 0x7fffffff;
        bad = bad || c < 0 || c >= n_out || (i + 1 < len && cn <= c);
        const int b0 = (unsigned)c < (unsigned)n_out ? c / W : nb;                  // bucket of this entry
        const int b1 = i + 1 < len ? ((unsigned)cn < (unsigned)n_out ? cn / W : nb) : nb;   // bucket of the next one (nb: past the end)
        if (i == 0) for (int b = 0; b <= b0 && b <= nb; ++b) out[b] = 0;
        for (int b = b0 + 1; b <= b1 && b <= nb; ++b) out[b] = i + 1;
    }
    if (__ballot(bad) != 0ull && lane == 0) flag[0] = 1;
}
// header: the longest slice (decides how many entries per lane the product kernel keeps in registers) and the entry count
__global__ __launch_bounds__(256) void payload_index_stats(int64_t n_ref, int nb, int W, int32_t *__restrict__ idx) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j == 0) idx[1] = W;
    int mx = 0; long long nz = 0;
    if (j < n_ref) {
        const int32_t *r = idx + A3_HDR + j * (nb + 1);
        for (int b = 0; b < nb; ++b) { const int l = r[b + 1] - r[b]; mx = l > mx ? l : mx; }
        nz = r[nb];
    }
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) { const int o = __shfl_xor(mx, sft); mx = o > mx ? o : mx; nz += __shfl_xor(nz, sft); }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(idx + 0, mx);
        atomicAdd(reinterpret_cast<unsigned long long *>(idx + 2), (unsigned long long)nz);
    }
}

// NCH: compile-time bound on the 64-column chunks of a wave's slice (W / 64 <= NCH).  The finished slice is read ONCE into
// NCH registers per lane (all LDS reads issued back to back, one wait), cleared behind those reads for the next row,
// counted and emitted from the registers: the count and the emission used to be two loops of read -> wait -> ballot
// with one LDS round trip per 64 columns each (round 3: ~2 x 59 x 130 cycles per row at 30 000 genes).
// X4 (round 3, the default; the name is historical): the BRANCH-FREE gather.  A lane holds EPL = 1, 2, 3, 4 (one group) or 8
// (two groups of 4) CONSECUTIVE entries of a slice, fetched with one 4- / 8- / 12- / 16-byte load per array, EPL chosen by
// the launcher as the smallest that covers the payload's longest slice: the accumulator updates -- one LDS read and one
// write instruction per entry of a lane, bank-conflicted on ~random columns -- are what the gather costs, not the bytes in
// flight (two or three register sets, entry-per-lane or vector loads: no difference; profiles/r03_a3_ablation.md).  A row
// whose slices all fit the registers takes a loop with no branch around any memory operation (FAST); a vector load may run
// up to three entries past the end of its slice (never used), so a slice that ends within three entries of the END OF THE
// ARRAYS (b_nnz) sends its row to the general loop.  !X4 (CELLMAP_A3_GATHER=1): round 2's entry-per-lane gather, for A/B.
// Narrow payloads (NCH <= 16) with at most two entries per lane are held to 128 registers: two workgroups per CU.
template <int EPL, int NCH, bool X4>
__global__ __launch_bounds__(512, (CM_A3_OCC4 && NCH <= CM_A3_OCCN && EPL <= CM_A3_OCC4 && X4) ? 4 : 2) void spgemm_sliced(const int64_t *__restrict__ indptr, const int32_t *__restrict__ cols,
                                                      const float *__restrict__ vals, int64_t nq,
                                                      const int64_t *__restrict__ bptr, const int32_t *__restrict__ bcols,
                                                      const float *__restrict__ bvals, const int32_t *__restrict__ bidx_, int nb1,
                                                      int n_out, int W, int64_t *out_indptr, int32_t *__restrict__ ocols,
                                                      float *__restrict__ ovals, const int32_t *__restrict__ order,
                                                      unsigned long long *chain, int64_t capacity, int32_t *__restrict__ out_rows,
                                                      int64_t b_nnz) {
    // EPL entries per lane and neighbour are held in registers (slices of up to 64 EPL entries; longer ones take a scalar
    // loop), NBATCH = 16 / EPL neighbours per register set: the same 2 x 32 registers of payload in flight per lane
    constexpr int NBATCH = 16 / EPL;
    const int32_t *__restrict__ bidx = bidx_ + A3_HDR;
    extern __shared__ __align__(16) unsigned char smem[];
    float *acc = reinterpret_cast<float *>(smem);                       // [8 W]
    int *wsum = reinterpret_cast<int *>(acc + 8 * W);                   // [2][8] wave totals (by row parity)
    long long *tk = reinterpret_cast<long long *>(wsum + 16);           // [0] ticket of the row after next, [1] the row's first entry
    float *dummy = reinterpret_cast<float *>(tk + 2);                   // [512] where lanes without an entry do their add
    float *zero = dummy + 512;                                          // [64] zeros: what a chunk beyond the slice reads
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wlo = wave * W;   // this wave's output columns: [wlo, min(wlo + W, n_out)) (possibly none)
    if (tid < 64) zero[tid] = 0.f;
    const int dix = (int)(dummy - acc) + lane;
    const bool ticketed = !out_rows;   // chained scan: rows by ticket (a predecessor is always held by a running workgroup)
    int parity = 0;

    int64_t row, nxt;
    if (ticketed) {
        // one ticket at a time (a workgroup's rows must interleave with the other workgroups' rows: a row waits for the counts
        // of ALL rows in front of it, and a ticket held back here is a row nobody works on)
        if (tid == 0) tk[0] = (long long)atomicAdd(chain + nq * CM_A3_CSTRIDE, 1ull);
        __syncthreads();
        row = tk[0]; nxt = nq;   // (the next row's ticket is taken when this one starts)
        __syncthreads();
    } else {
        row = blockIdx.x; nxt = row + gridDim.x;
        __syncthreads();   // (zero[] is written)
    }

    // ---- neighbour metadata of a row, per wave: lane l holds neighbours l and l + 64 (slice start, slice length, weight)
    struct Meta { int slo[2], shi[2], len[2]; float a[2]; int deg; int64_t p0; };
    struct Stage { int64_t src, p0; int deg; int j[2]; float a[2]; int64_t b0[2]; int s[2], e[2]; };
    auto stage0 = [&](Stage &g, int64_t r) { g.src = r < nq ? (order ? (int64_t)order[r] : r) : -1; };
    auto stage1 = [&](Stage &g) {
        g.p0 = 0; g.deg = 0;
        if (g.src >= 0) { g.p0 = indptr[g.src]; g.deg = (int)(indptr[g.src + 1] - g.p0); }
    };
    auto stage2 = [&](Stage &g) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int u = h * 64 + lane;
            const bool in = u < g.deg;
            g.j[h] = in ? cols[g.p0 + u] : -1;
            g.a[h] = in ? vals[g.p0 + u] : 0.f;
        }
    };
    auto stage3 = [&](Stage &g) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            g.b0[h] = 0; g.s[h] = 0; g.e[h] = 0;
            if (g.j[h] >= 0 && wave < nb1 - 1) {   // (a wave beyond the last bucket has no columns)
                g.b0[h] = bptr[g.j[h]];
                const int32_t *ip = bidx + (int64_t)g.j[h] * nb1 + wave;
                g.s[h] = ip[0]; g.e[h] = ip[1];
            }
        }
    };
    auto finish = [&](const Stage &g, Meta &m) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t st = g.b0[h] + g.s[h];
            m.slo[h] = (int)(st & 0xffffffffll); m.shi[h] = (int)(st >> 32);
            m.len[h] = g.e[h] - g.s[h];
            m.a[h] = g.a[h];
        }
        m.deg = g.deg; m.p0 = g.p0;
    };
    Stage g;
    Meta cur;
    stage0(g, row); stage1(g); stage2(g); stage3(g); finish(g, cur);
    if (!ticketed) stage0(g, nxt);

    const int nch = W / 64;   // chunks of this wave's slice (<= NCH; columns at or beyond n_out are never written: they stay 0)
    const int zix = (int)(zero - acc);
    // the row whose entries wait in registers for their place (deferred emission, see the end of the row loop)
    float rv[NCH];
    int64_t prev_row = 0;
    int prev_all = 0, prev_before = 0;
    bool have_prev = false;
    // wave 0: the sum of the counts of all rows in front of row r -- published counts back to the nearest published
    // inclusive prefix, 64 x CM_A3_LBW rows per step (that many loads per lane in flight; the nearest inclusive prefix is
    // a few hundred rows back: every workgroup places a row one row after counting it)
    // (Fetching the first window early -- an LDS-DMA of the 256 rows in front, issued from inline assembly in the last round of
    // the next row's gather so that the round trip runs under the last neighbours' updates -- was built and measured: the
    // window always resolves and the kernel is 8 % slower; profiles/r03_a3_ablation.md, section 4.)
    auto lookback = [&](int64_t r) -> long long {
        constexpr unsigned long long VMASK = (1ull << 62) - 1ull;
        constexpr int LBW = CM_A3_LBW;   // loads per lane and step
        long long excl = 0;
        for (int64_t pos = (CM_A3_EXP & 16) ? -1 : r - 1; pos >= 0; pos -= 64 * LBW) {
            unsigned long long v[LBW];
            int ustar = LBW, fi = 63;
            for (;;) {
#pragma unroll
                for (int u = 0; u < LBW; ++u) {
                    const int64_t i = pos - 64 * u - lane;
                    v[u] = i >= 0 ? __hip_atomic_load(chain + i * CM_A3_CSTRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (2ull << 62);
                }
                bool retry = false;
                ustar = LBW;
#pragma unroll
                for (int u = 0; u < LBW; ++u) {
                    const unsigned long long empty = __ballot((v[u] >> 62) == 0ull), incl = __ballot((v[u] >> 62) == 2ull);
                    if (ustar == LBW && !retry) {
                        if (incl != 0ull && (empty == 0ull || __ffsll((long long)incl) < __ffsll((long long)empty))) {
                            ustar = u; fi = __ffsll((long long)incl) - 1;
                        } else if (empty != 0ull) retry = true;
                    }
                }
                if (!retry) break;
                __builtin_amdgcn_s_sleep(2);
            }
            long long part = 0;
#pragma unroll
            for (int u = 0; u < LBW; ++u)
                if (u < ustar || (u == ustar && lane <= fi)) part += (long long)(v[u] & VMASK);
#pragma unroll
            for (int sft = 32; sft > 0; sft >>= 1) part += __shfl_xor(part, sft);
            excl += part;
            if (ustar < LBW) break;
        }
        return excl;
    };
    // the place of row r (count n) in the output -> tk[1]; also the row's indptr entry and, chained, its inclusive prefix
    auto place = [&](int64_t r, int n) {
        if (!ticketed) {
            if (tid == 0) {
                const unsigned long long old = atomicAdd(chain + nq * CM_A3_CSTRIDE + 1, (1ull << 40) | (unsigned long long)n);
                const long long j = (long long)(old >> 40), o = (long long)(old & ((1ull << 40) - 1ull));
                out_indptr[j] = o;
                if (j == nq - 1) out_indptr[nq] = o + n;
                out_rows[j] = (int32_t)r;
                tk[1] = o;
            }
        } else if (wave == 0) {
            const long long excl = lookback(r);
            if (lane == 0) {
                if (r > 0)
                    __hip_atomic_store(chain + r * CM_A3_CSTRIDE, (2ull << 62) | (unsigned long long)(excl + n), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else out_indptr[0] = 0;
                out_indptr[r + 1] = excl + n;
                tk[1] = excl;
            }
        }
    };
    // this wave's slice of a row (ascending columns; consecutive lanes write consecutive entries)
    auto emit = [&](float (&r_)[NCH], int64_t row_out, int before, int all) {
        if (row_out + all > capacity) return;
        int64_t o = row_out + before;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const float v = r_[i];
            const bool nz = v != 0.f;
            const unsigned long long m = __ballot(nz);
            if (m != 0ull) {   // (uniform)
                if (nz && !(CM_A3_EXP & 8)) {
                    const int64_t q = o + __popcll(m & ((1ull << lane) - 1ull));
                    ocols[q] = wlo + 64 * i + lane;
                    ovals[q] = v;
                }
                o += __popcll(m);
            }
        }
    };
    {   // the slice starts cleared; every row clears it again right behind its read-out
        float4 *a4 = reinterpret_cast<float4 *>(acc + wlo);
        for (int c = lane; c < W / 4; c += 64) a4[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    while (row < nq) {
        // the next row: static stride, its metadata fetched in four stages spread over this row -- or, with tickets, taken
        // now, known to the waves after the first barrier and fetched during the emission (a ticket held for longer is a
        // row the other workgroups wait for)
        const int64_t nxt2 = nxt + gridDim.x;
        if (ticketed && tid == 0) tk[0] = (long long)atomicAdd(chain + nq * CM_A3_CSTRIDE, 1ull);
        if (!ticketed) stage1(g);
        // ---- the gather: batches of NBATCH neighbours, two register sets
        const int nn = cur.deg < 128 ? cur.deg : 128;
        int ca[NBATCH][EPL], cb[NBATCH][EPL];
        float va[NBATCH][EPL], vb[NBATCH][EPL];
        struct BM { int slo, shi, len; float a; int base; };
        auto pick = [&](int ub) {   // the metadata registers that hold neighbours ub .. ub + NBATCH - 1 (ub is a multiple of NBATCH)
            BM m;
            const bool hi = ub >= 64, none = ub >= 128;
            m.slo = hi ? cur.slo[1] : cur.slo[0];
            m.shi = hi ? cur.shi[1] : cur.shi[0];
            m.len = none ? 0 : (hi ? cur.len[1] : cur.len[0]);
            m.a = hi ? cur.a[1] : cur.a[0];
            m.base = ub & 63;
            return m;
        };
        auto uni = [&](const BM &m, int n, int64_t &st, int &len, float &wa) {
            const int l = m.base + n;
            len = __builtin_amdgcn_readlane(m.len, l);
            st = ((int64_t)__builtin_amdgcn_readlane(m.shi, l) << 32) | (uint32_t)__builtin_amdgcn_readlane(m.slo, l);
            wa = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m.a), l));
        };
        auto load_batch = [&](int (&cc)[NBATCH][EPL], float (&vv)[NBATCH][EPL], const BM &bm) {
#pragma unroll
            for (int n = 0; n < NBATCH; ++n) {
                int64_t st; int len; float wa;
                uni(bm, n, st, len, wa);
                const bool has = len > 0;
                const int32_t *pc = has ? bcols + st : reinterpret_cast<const int32_t *>(indptr);
                const float *pv = has ? bvals + st : reinterpret_cast<const float *>(indptr);
                const int last = has ? len - 1 : 0;
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    const unsigned o_ = (unsigned)min(lane + 64 * e, last) << 2;
                    cc[n][e] = *reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(pc) + o_);
                    vv[n][e] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(pv) + o_);
                }
            }
        };
        auto apply_batch = [&](int (&cc)[NBATCH][EPL], float (&vv)[NBATCH][EPL], const BM &bm) {
#pragma unroll
            for (int n = 0; n < NBATCH; ++n) {
                int64_t st; int len; float wa;
                uni(bm, n, st, len, wa);
                if (len > 0) {   // (uniform)
                    // the entries of one neighbour row have distinct columns: all reads go out before the writes
                    // (skipping, under a uniform branch, the entry sets that lie wholly beyond a slice was measured: 131 ->
                    // 158 ms per million rows -- a branch around LDS operations makes the compiler's wait counting conservative)
                    float *ap_[EPL]; float x_[EPL];
#pragma unroll
                    for (int e = 0; e < EPL; ++e) {
                        ap_[e] = acc + (lane + 64 * e < len ? cc[n][e] : dix + 64 * e);
                        x_[e] = *ap_[e];
                    }
#pragma unroll
                    for (int e = 0; e < EPL; ++e) *ap_[e] = __fadd_rn(x_[e], __fmul_rn(wa, vv[n][e]));
                    if (len > 64 * EPL) {   // (uniform) a slice longer than the registers hold
                        for (int t = 64 * EPL + lane; t < len; t += 64) {
                            float *ap = acc + bcols[st + t];
                            *ap = __fadd_rn(*ap, __fmul_rn(wa, bvals[st + t]));
                        }
                    }
                }
            }
        };
        if constexpr (!X4) {
        BM ma = pick(0), mb;
        load_batch(ca, va, ma);
        for (int ub = 0; ub < nn; ub += 2 * NBATCH) {
            mb = pick(ub + NBATCH);
            load_batch(cb, vb, mb);
            apply_batch(ca, va, ma);
            if (ub == 0 && !ticketed) stage2(g);
            ma = pick(ub + 2 * NBATCH);
            load_batch(ca, va, ma);
            apply_batch(cb, vb, mb);
        }
        } else {
            // EPL = 2, 3, 4 or 8 entries per lane: one group of 2 / 3 / 4 CONSECUTIVE entries (one 8- / 12- / 16-byte load
            // per array) or two groups of 4.  The LDS updates are what the gather costs (bank conflicts of ~random columns),
            // one read and one write instruction per entry of a lane: three entries per lane cover config 5's slices (longest
            // 171) with 3 + 3 instead of 4 + 4 LDS instructions per neighbour.
            constexpr int EPG = (EPL + 3) / 4, GE = EPL / EPG, NB = (CM_A3_OCC4 && NCH <= CM_A3_OCCN && EPL >= 2 && EPL <= CM_A3_OCC4) ? 4 / EPG : ((CM_A3_NB8 && EPL <= 2) ? 8 : 4 / EPG);   // groups per lane, entries per group, neighbours per set
            static_assert(GE * EPG == EPL && GE >= 1 && GE <= 4, "entries per lane: 1, 2, 3, 4 or 8");
            typedef int ivu_t __attribute__((ext_vector_type(GE), aligned(4)));
            typedef float fvu_t __attribute__((ext_vector_type(GE), aligned(4)));
            typedef int iv_t __attribute__((ext_vector_type(GE)));
            typedef float fv_t __attribute__((ext_vector_type(GE)));
            // (uniform) the vector loads of this slice stay inside the payload arrays
            auto safe = [&](int64_t st, int len) { const int r = (len + GE - 1) / GE * GE; return st + (r < GE ? GE : r) <= b_nnz; };
            // FAST: every slice of this row and wave fits the registers and none touches the end of the arrays -- the loop
            // then holds no load besides the register sets' (a conditional entry-per-lane loop between two neighbours makes
            // the compiler's load counting conservative: it waited for part of the NEXT set before every neighbour)
            auto gather = [&](auto fast_) {
                constexpr bool FAST = decltype(fast_)::value;
                iv_t c0[NB][EPG], c1[NB][EPG], c2[NB][EPG];
                fv_t v0[NB][EPG], v1[NB][EPG], v2[NB][EPG];
                auto load_set = [&](iv_t (&cc)[NB][EPG], fv_t (&vv)[NB][EPG], const BM &bm) {
#pragma unroll
                    for (int n = 0; n < NB; ++n) {
                        int64_t st; int len; float wa;
                        uni(bm, n, st, len, wa);
                        const bool has = len > 0 && (FAST || safe(st, len));
                        const char *pc = reinterpret_cast<const char *>(has ? bcols + st : reinterpret_cast<const int32_t *>(indptr));
                        const char *pv = reinterpret_cast<const char *>(has ? bvals + st : reinterpret_cast<const float *>(indptr));
                        const int lim = has ? len : 0;
#pragma unroll
                        for (int q = 0; q < EPG; ++q) {
                            const int f = 64 * GE * q + GE * lane;
                            const unsigned o_ = f < lim ? (unsigned)f << 2 : 0u;
                            cc[n][q] = *reinterpret_cast<const ivu_t *>(pc + o_);
                            vv[n][q] = *reinterpret_cast<const fvu_t *>(pv + o_);
                        }
                    }
                };
                auto apply_set = [&](iv_t (&cc)[NB][EPG], fv_t (&vv)[NB][EPG], const BM &bm) {
#pragma unroll
                    for (int n = 0; n < NB; ++n) {
                        int64_t st; int len; float wa;
                        uni(bm, n, st, len, wa);
                        if (CM_A3_EXP & 2) {
#pragma unroll
                            for (int q = 0; q < EPG; ++q) asm volatile("" ::"v"(cc[n][q]), "v"(vv[n][q]));
                        } else if (FAST || len > 0) {   // (uniform; FAST: no branch at all -- a neighbour slot without entries works on the
                                                        //  lanes' private slots only, and the updates of consecutive neighbours are one
                                                        //  straight stretch the compiler can schedule: 132 -> 129 ms per million rows)
                            int t0 = 0;
                            if (FAST || safe(st, len)) {   // (uniform)
                                float *ap_[EPL]; float x_[EPL];
#pragma unroll
                                for (int q = 0; q < EPG; ++q)
#pragma unroll
                                    for (int e = 0; e < GE; ++e) {
                                        ap_[GE * q + e] = acc + (64 * GE * q + GE * lane + e < len ? cc[n][q][e] : dix + 64 * (GE * q + e));
                                        x_[GE * q + e] = *ap_[GE * q + e];
                                    }
#pragma unroll
                                for (int q = 0; q < EPG; ++q)
#pragma unroll
                                    for (int e = 0; e < GE; ++e) *ap_[GE * q + e] = __fadd_rn(x_[GE * q + e], __fmul_rn(wa, vv[n][q][e]));
                                t0 = 64 * EPL;
                            }
                            if constexpr (!FAST) {
                                if (len > t0) {   // (uniform) beyond the registers, or a slice at the very end of the arrays
                                    for (int t = t0 + lane; t < len; t += 64) {
                                        float *ap = acc + bcols[st + t];
                                        *ap = __fadd_rn(*ap, __fmul_rn(wa, bvals[st + t]));
                                    }
                                }
                            }
                        }
                    }
                };
                if constexpr (CM_A3_NSET == 3) {
                    BM m0 = pick(0), m1 = pick(NB), m2;
                    load_set(c0, v0, m0);
                    load_set(c1, v1, m1);
                    for (int ub = 0; ub < nn; ub += 3 * NB) {
                        m2 = pick(ub + 2 * NB);
                        load_set(c2, v2, m2);
                        apply_set(c0, v0, m0);
                        if (ub == 0 && !ticketed) stage2(g);
                        m0 = pick(ub + 3 * NB);
                        load_set(c0, v0, m0);
                        apply_set(c1, v1, m1);
                        m1 = pick(ub + 4 * NB);
                        load_set(c1, v1, m1);
                        apply_set(c2, v2, m2);
                    }
                } else {
                    BM m0 = pick(0), m1;
                    load_set(c0, v0, m0);
                    for (int ub = 0; ub < nn; ub += 2 * NB) {
                        m1 = pick(ub + NB);
                        load_set(c1, v1, m1);
                        apply_set(c0, v0, m0);
                        if (ub == 0 && !ticketed) stage2(g);
                        m0 = pick(ub + 2 * NB);
                        load_set(c0, v0, m0);
                        apply_set(c1, v1, m1);
                    }
                }
            };
            bool slow_ = false;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int64_t st = ((int64_t)cur.shi[h] << 32) | (uint32_t)cur.slo[h];
                slow_ = slow_ || cur.len[h] > 64 * EPL || (cur.len[h] > 0 && !safe(st, cur.len[h]));
            }
            if (__ballot(slow_) == 0ull) gather(std::integral_constant<bool, true>{});
            else gather(std::integral_constant<bool, false>{});
        }
        if (nn == 0 && !ticketed) stage2(g);   // (a row without neighbours: the loop above did not run)
        // neighbours beyond the 128 held in registers (rows of very high degree): one at a time
        for (int u = 128; u < cur.deg; ++u) {
            const int64_t jn = cols[cur.p0 + u];
            const float wa = vals[cur.p0 + u];
            const int64_t b0 = bptr[jn];
            const bool own = wave < nb1 - 1;
            const int s_ = own ? bidx[jn * nb1 + wave] : 0, e_ = own ? bidx[jn * nb1 + wave + 1] : 0;
            for (int t = s_ + lane; t < e_; t += 64) {
                float *ap = acc + bcols[b0 + t];
                *ap = __fadd_rn(*ap, __fmul_rn(wa, bvals[b0 + t]));
            }
        }
        if (!ticketed) stage3(g);
        // ---- the row in front of this one: its entries have been waiting in registers while this row was gathered
        // (deferred emission, round 3).  A row's place is the sum of the counts of ALL rows in front of it, and the ~255 rows
        // right in front are the ones the other workgroups hold: placed at once, a row waited for the slowest of them -- a
        // barrier across the whole grid per row (a quarter of the kernel's time).  One row later they have all published
        // their counts.  The stores go out BEFORE the read-out and the count of this row (LDS only), so they have left
        // the memory pipeline when the next row's first loads are waited for.
        if (have_prev) place(prev_row, prev_all);
        __syncthreads();
        if (ticketed) { nxt = tk[0]; stage0(g, nxt); stage1(g); }
        if (have_prev) emit(rv, tk[1], prev_before, prev_all);
        // ---- read the finished slice out into registers, clear it for the next row, count its non-zeros
        int nch_ = nch;
        asm volatile("" : "+s"(nch_));   // (the NCH chunk offsets are formed here, not hoisted out of the row loop into NCH registers)
#pragma unroll
        for (int i = 0; i < NCH; ++i) rv[i] = acc[(i < nch_ ? wlo + 64 * i : zix) + lane];   // (uniform select, no branch)
        {
            float4 *a4 = reinterpret_cast<float4 *>(acc + wlo);
            for (int c = lane; c < W / 4; c += 64) a4[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        int wcount = 0;
#pragma unroll
        for (int i = 0; i < NCH; ++i) wcount += __popcll(__ballot(rv[i] != 0.f));
#pragma unroll
        for (int i = 0; i < NCH; ++i) asm volatile("" : "+v"(rv[i]));   // (the emission tests again: 64 masks would not stay in SGPRs)
        if (lane == 0) wsum[parity * 8 + wave] = wcount;
        if (ticketed) stage2(g);
        __syncthreads();
        int before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < 8; ++w) { const int s_ = wsum[parity * 8 + w]; all += s_; if (w < wave) before += s_; }
        parity ^= 1;
        // this row's count is published at once (row 0: it is its own inclusive prefix)
        if (ticketed && wave == 0 && lane == 0)
            __hip_atomic_store(chain + row * CM_A3_CSTRIDE, ((row == 0 ? 2ull : 1ull) << 62) | (unsigned long long)all, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        prev_row = row; prev_all = all; prev_before = before; have_prev = true;
        // ---- next row
        if (ticketed) stage3(g);
        finish(g, cur);
        row = nxt;
        if (!ticketed) { nxt = nxt2; stage0(g, nxt); }
    }
    // ---- the last row of this workgroup
    if (have_prev) {
        place(prev_row, prev_all);
        __syncthreads();
        emit(rv, tk[1], prev_before, prev_all);
    }
}
}  // namespace

size_t cm_a3_chain_bytes(int64_t nq) { return 8 * ((size_t)(nq > 0 ? nq : 0) * CM_A3_CSTRIDE + 2); }
int cm_a3_bucket_width(int64_t n_out) {
    const int64_t w = (((n_out + 7) / 8) + 63) & ~63LL;
    return (int)(w < 64 ? 64 : w);
}
int64_t cm_payload_index_len_impl(int64_t n_ref, int64_t n_out) {
    if (n_ref < 0 || n_out < 1 || n_out > SPG_CT) return -1;
    const int W = cm_a3_bucket_width(n_out);
    const int nb = (int)((n_out + W - 1) / W);
    return A3_HDR + n_ref * (nb + 1);
}
// d_flag: one int of scratch, 1 afterwards if a payload row is not canonical (columns strictly ascending, inside [0, n_out))
cm_status cm_payload_index_run(cm_handle_t h, const int64_t *d_bptr, const int32_t *d_bcols, int64_t n_ref, int64_t n_out,
                               int32_t *d_index, int *d_flag) {
    if (n_out < 1 || n_out > SPG_CT) CM_FAIL(h, CM_E_INVALID, "payload index: 1 .. %d output columns", SPG_CT);
    const int W = cm_a3_bucket_width(n_out);
    const int nb = (int)((n_out + W - 1) / W);
    CM_HIP(h, hipMemsetAsync(d_flag, 0, 4, h->stream));
    CM_HIP(h, hipMemsetAsync(d_index, 0, 4 * A3_HDR, h->stream));
    if (n_ref > 0) {
        hipLaunchKernelGGL(payload_index, dim3((unsigned)((n_ref + 3) / 4)), dim3(256), 0, h->stream, d_bptr, d_bcols, n_ref, (int)n_out, W, nb,
                           d_index, d_flag);
        hipLaunchKernelGGL(payload_index_stats, dim3((unsigned)((n_ref + 255) / 256)), dim3(256), 0, h->stream, n_ref, nb, W, d_index);
    }
    CM_HIP(h, hipGetLastError());
    return CM_OK;
}
// The header of an index (host copy): read once per index and kept on the handle.
cm_status cm_payload_index_header(cm_handle_t h, const int32_t *d_index, int *max_slice, long long *nnz) {
    if (h->a3_index != d_index) {
        int32_t hdr[A3_HDR];
        CM_HIP(h, hipMemcpyAsync(hdr, d_index, sizeof(hdr), hipMemcpyDeviceToHost, h->stream));
        CM_HIP(h, hipStreamSynchronize(h->stream));
        h->a3_index = d_index;
        h->a3_max_slice = hdr[0];
        h->a3_nnz = (long long)(((unsigned long long)(uint32_t)hdr[3] << 32) | (uint32_t)hdr[2]);
    }
    *max_slice = h->a3_max_slice;
    *nnz = h->a3_nnz;
    return CM_OK;
}
cm_status cm_spmm_csr_sliced_run(cm_handle_t h, const int64_t *d_indptr, const int32_t *d_cols, const float *d_vals,
                                 int64_t nq, const int64_t *d_bptr, const int32_t *d_bcols, const float *d_bvals, int64_t n_ref,
                                 const int32_t *d_index, int64_t n_out, int64_t *d_out_indptr, int32_t *d_out_cols,
                                 float *d_out_vals, int64_t capacity, const int32_t *d_order, unsigned long long *d_chain,
                                 int32_t *d_out_rows) {
    if (n_out < 1 || n_out > SPG_CT) CM_FAIL(h, CM_E_INVALID, "spmm_csr (single pass): 1 .. %d output columns", SPG_CT);
    if (nq == 0) { CM_HIP(h, hipMemsetAsync(d_out_indptr, 0, 8, h->stream)); return CM_OK; }
    if (d_out_rows && nq >= (1LL << 24)) CM_FAIL(h, CM_E_UNSUPPORTED, "spmm_csr: rows in completion order: at most 2^24 - 1 rows per call");
    const int W = cm_a3_bucket_width(n_out);
    const int nb = (int)((n_out + W - 1) / W);
    // entries per lane held in registers: enough for the longest slice when that is at most 512 entries, else for 1.5 x the
    // mean slice (longer slices take the scalar loop)
    int max_slice = 0; long long nnz = 0;
    CM_TRY(cm_payload_index_header(h, d_index, &max_slice, &nnz));
    int want = max_slice;
    if (max_slice > 512 && n_ref > 0) want = (int)std::min<long long>(512, (3 * nnz) / (2 * n_ref * nb) + 1);
    const size_t lds = (size_t)8 * W * 4 + 64 + 16 + 2048 + 256;
    if (h->n_cu <= 0) {
        hipDeviceProp_t pr;
        CM_HIP(h, hipGetDeviceProperties(&pr, h->device));
        h->n_cu = pr.multiProcessorCount;
    }
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(4, (size_t)(160 * 1024) / (lds + 1024)));
    const unsigned grid = (unsigned)std::min<int64_t>(nq, (int64_t)h->n_cu * per_cu);
    const int nch = W / 64;
    // slices of more than 64 entries: the 16-byte gather (X4) with 4 or 8 entries per lane; shorter ones (and
    // CELLMAP_A3_GATHER=1, the A/B switch) keep the entry-per-lane gather
    const bool x4 = cm_get_knobs().a3_gather == 0 && (want > 64 || CM_A3_X1);
    // entry-per-lane gather: 1 entry per lane when no slice is longer than 64 (narrow or very sparse payloads, gene-column
    // shards of a multi-GPU run): one LDS read + write per neighbour instead of two
    const int epl = x4 ? (want <= 64 ? 1 : (want <= 128 ? 2 : (want <= 192 ? 3 : (want <= 256 ? 4 : 8))))
                       : (want <= 64 && CM_A3_EPL1 ? 1 : (want <= 128 ? 2 : (want <= 256 ? 4 : 8)));
    auto pick_nch = [&](auto e, auto x) {
        constexpr int E = decltype(e)::value;
        constexpr bool X = decltype(x)::value;
        return nch <= 16 ? spgemm_sliced<E, 16, X> : (nch <= 32 ? spgemm_sliced<E, 32, X> : spgemm_sliced<E, 64, X>);
    };
    typedef std::integral_constant<bool, true> T_; typedef std::integral_constant<bool, false> F_;
    auto kern = x4 ? (epl == 1 ? pick_nch(std::integral_constant<int, 1>{}, T_{}) : epl == 2 ? pick_nch(std::integral_constant<int, 2>{}, T_{})
                               : (epl == 3 ? pick_nch(std::integral_constant<int, 3>{}, T_{})
                                           : (epl == 4 ? pick_nch(std::integral_constant<int, 4>{}, T_{}) : pick_nch(std::integral_constant<int, 8>{}, T_{}))))
                   : (epl == 1 ? pick_nch(std::integral_constant<int, 1>{}, F_{}) : epl == 2 ? pick_nch(std::integral_constant<int, 2>{}, F_{})
                               : (epl == 4 ? pick_nch(std::integral_constant<int, 4>{}, F_{}) : pick_nch(std::integral_constant<int, 8>{}, F_{})));
    CM_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CM_HIP(h, hipMemsetAsync(d_chain, 0, cm_a3_chain_bytes(nq), h->stream));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, h->stream, d_indptr, d_cols, d_vals, nq, d_bptr, d_bcols, d_bvals, d_index, nb + 1,
                       (int)n_out, W, d_out_indptr, d_out_cols, d_out_vals, d_order, d_chain, capacity, d_out_rows, (int64_t)nnz);
    CM_HIP(h, hipGetLastError());
    return CM_OK;
}

// ---- locality order of the mapping-matrix rows: rows grouped by their smallest neighbour column (counting sort over the
// column range).  Queries that share a neighbour sit close to each other, so consecutive rows of this order gather mostly
// the same payload rows.  Within a group the order is the arrival order of the scatter (any order serves the purpose).
namespace {
__global__ void lo_hist(const int64_t *__restrict__ indptr, const int32_t *__restrict__ cols, int64_t nq, int64_t n_ref, int *__restrict__ hist) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const int64_t p = indptr[i];
    const int64_t key = p < indptr[i + 1] ? (int64_t)cols[p] : n_ref;   // empty rows last
    atomicAdd(&hist[key], 1);
}
__global__ void lo_scatter(const int64_t *__restrict__ indptr, const int32_t *__restrict__ cols, int64_t nq, int64_t n_ref,
                           const int64_t *__restrict__ start, int *__restrict__ cursor, int32_t *__restrict__ order) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const int64_t p = indptr[i];
    const int64_t key = p < indptr[i + 1] ? (int64_t)cols[p] : n_ref;
    order[start[key] + atomicAdd(&cursor[key], 1)] = (int32_t)i;
}
}  // namespace
size_t cm_locality_order_ws_bytes(int64_t n_ref) { return 2 * cm_align(4 * (size_t)(n_ref + 1)) + cm_align(8 * (size_t)(n_ref + 2)) + cm_scan_ws_bytes(n_ref + 1) + 4096; }
cm_status cm_locality_order_run(cm_handle_t h, const int64_t *d_indptr, const int32_t *d_cols, int64_t nq, int64_t n_ref, int32_t *d_order) {
    if (nq == 0) return CM_OK;
    int *hist = (int *)cm_ws_alloc(h, 4 * (size_t)(n_ref + 1));
    int *cursor = (int *)cm_ws_alloc(h, 4 * (size_t)(n_ref + 1));
    int64_t *start = (int64_t *)cm_ws_alloc(h, 8 * (size_t)(n_ref + 2));
    if (!hist || !cursor || !start) CM_FAIL(h, CM_E_NOMEM, "locality order: device workspace exhausted");
    CM_HIP(h, hipMemsetAsync(hist, 0, 4 * (size_t)(n_ref + 1), h->stream));
    CM_HIP(h, hipMemsetAsync(cursor, 0, 4 * (size_t)(n_ref + 1), h->stream));
    hipLaunchKernelGGL(lo_hist, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, h->stream, d_indptr, d_cols, nq, n_ref, hist);
    CM_TRY(cm_exclusive_scan_i32_to_i64(h, hist, n_ref + 1, start));
    hipLaunchKernelGGL(lo_scatter, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, h->stream, d_indptr, d_cols, nq, n_ref, (const int64_t *)start,
                       cursor, d_order);
    CM_HIP(h, hipGetLastError());
    return CM_OK;
}

// ------------------------------------------------------------------ T1: precomputed-distance ingest --
// Square CSR distance matrix (adata.obsp['distances']) -> padded (indices, distances) rows with -1 / inf padding
// (reference utils.py:129-219: two Python loops over all cells).  include_self: -1 keep as stored, 0 drop the
// diagonal entry, 1 append (self, 0.0) when the row has no diagonal entry.  A row whose distances are already
// non-decreasing keeps its stored order; otherwise it is sorted by distance (ties: stored position -- the reference's
// np.argsort leaves tie order unspecified).
namespace {
constexpr int IG_MAX = 1024;   // longest row (+1) the in-LDS sort handles

__global__ void ingest_count(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices, int64_t n,
                             int include_self, int *__restrict__ rowlen, int *__restrict__ kmax) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int len = 0;
    if (i < n) {
        const int64_t lo = indptr[i], hi = indptr[i + 1];
        len = (int)(hi - lo);
        if (include_self >= 0) {
            bool has = false;
            for (int64_t p = lo; p < hi; ++p) has = has || indices[p] == (int32_t)i;
            if (include_self == 1 && !has) ++len;
            if (include_self == 0 && has) {   // every diagonal entry is dropped (duplicates are not canonical CSR, but legal)
                int c = 0;
                for (int64_t p = lo; p < hi; ++p) c += indices[p] == (int32_t)i;
                len -= c;
            }
        }
        rowlen[i] = len;
    }
    for (int o = 32; o > 0; o >>= 1) len = max(len, __shfl_xor(len, o));
    if ((threadIdx.x & 63) == 0 && len > 0) atomicMax(kmax, len);
}

__global__ __launch_bounds__(256) void ingest_fill(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                   const double *__restrict__ data, int64_t n, int include_self, int kmax,
                                                   int64_t *__restrict__ out_idx, double *__restrict__ out_dist) {
    __shared__ double s_d[4][IG_MAX];
    __shared__ int s_c[4][IG_MAX];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 4 + w;
    if (i >= n) return;
    const int64_t lo = indptr[i], hi = indptr[i + 1];
    // gather (stable compaction of the kept entries, wave-wide)
    int m = 0;
    for (int64_t base = lo; base < hi; base += 64) {
        const int64_t p = base + lane;
        const bool in = p < hi;
        const int c = in ? indices[p] : 0;
        const bool keep = in && !(include_self == 0 && c == (int32_t)i);
        const unsigned long long b = __ballot(keep);
        if (keep) {
            const int pos = m + __popcll(b & ((1ull << lane) - 1ull));
            s_d[w][pos] = data[p];
            s_c[w][pos] = c;
        }
        m += __popcll(b);
    }
    if (include_self == 1) {
        bool has = false;
        for (int s = lane; s < m; s += 64) has = has || s_c[w][s] == (int32_t)i;
        if (__ballot(has) == 0ull) {
            if (lane == 0) { s_d[w][m] = 0.0; s_c[w][m] = (int)i; }
            ++m;
        }
    }
    __builtin_amdgcn_wave_barrier();
    // already non-decreasing?  (np.all(np.diff(d) >= 0): a NaN difference counts as unsorted)
    bool bad = false;
    for (int s = lane; s + 1 < m; s += 64) bad = bad || !(s_d[w][s + 1] - s_d[w][s] >= 0.0);
    const bool sorted = __ballot(bad) == 0ull;
    int64_t *oi = out_idx + i * (int64_t)kmax;
    double *od = out_dist + i * (int64_t)kmax;
    for (int s = lane; s < m; s += 64) {
        const double v = s_d[w][s];
        int rk = s;
        if (!sorted) {
            rk = 0;
            for (int j = 0; j < m; ++j) {
                const double u = s_d[w][j];
                rk += (u < v || (u == v && j < s)) ? 1 : 0;
            }
        }
        oi[rk] = s_c[w][s];
        od[rk] = v;
    }
    for (int s = m + lane; s < kmax; s += 64) { oi[s] = -1; od[s] = INFINITY; }
}
// Rows with 1 024 or more stored neighbours (no limit upstream, utils.py:129-219): one WORKGROUP per row, the row in
// dynamic LDS (up to IG_LONG_MAX entries), wave 0 gathers (stable compaction as above), all 256 threads rank.
constexpr int IG_LONG_MAX = CM_INGEST_LDS_MAX;
__global__ __launch_bounds__(256) void ingest_fill_long(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                        const double *__restrict__ data, int64_t n, int include_self, int kmax,
                                                        int cap, int64_t *__restrict__ out_idx, double *__restrict__ out_dist) {
    extern __shared__ __align__(16) unsigned char ig_smem[];
    double *s_d = reinterpret_cast<double *>(ig_smem);
    int *s_c = reinterpret_cast<int *>(s_d + cap);
    __shared__ int s_m;
    const int tid = threadIdx.x, lane = tid & 63;
    for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
        const int64_t lo = indptr[i], hi = indptr[i + 1];
        if (tid < 64) {
            int m = 0;
            for (int64_t base = lo; base < hi; base += 64) {
                const int64_t p = base + lane;
                const bool in = p < hi;
                const int c = in ? indices[p] : 0;
                const bool keep = in && !(include_self == 0 && c == (int32_t)i);
                const unsigned long long b = __ballot(keep);
                if (keep) {
                    const int pos = m + __popcll(b & ((1ull << lane) - 1ull));
                    s_d[pos] = data[p];
                    s_c[pos] = c;
                }
                m += __popcll(b);
            }
            if (include_self == 1) {
                bool has = false;
                for (int s_ = lane; s_ < m; s_ += 64) has = has || s_c[s_] == (int32_t)i;
                if (__ballot(has) == 0ull) {
                    if (lane == 0) { s_d[m] = 0.0; s_c[m] = (int)i; }
                    ++m;
                }
            }
            if (lane == 0) s_m = m;
        }
        __syncthreads();
        const int m = s_m;
        bool bad = false;
        for (int s_ = tid; s_ + 1 < m; s_ += 256) bad = bad || !(s_d[s_ + 1] - s_d[s_] >= 0.0);
        const bool sorted = __syncthreads_or(bad ? 1 : 0) == 0;
        int64_t *oi = out_idx + i * (int64_t)kmax;
        double *od = out_dist + i * (int64_t)kmax;
        for (int s_ = tid; s_ < m; s_ += 256) {
            const double v = s_d[s_];
            int rk = s_;
            if (!sorted) {
                rk = 0;
                for (int j = 0; j < m; ++j) {
                    const double u = s_d[j];
                    rk += (u < v || (u == v && j < s_)) ? 1 : 0;
                }
            }
            oi[rk] = s_c[s_];
            od[rk] = v;
        }
        for (int s_ = m + tid; s_ < kmax; s_ += 256) { oi[s_] = -1; od[s_] = INFINITY; }
        __syncthreads();
    }
}
}  // namespace

cm_status cm_ingest_count_run(cm_handle_t h, const int64_t *d_indptr, const int32_t *d_indices, int64_t n, int include_self,
                              int *d_rowlen, int *d_kmax) {
    if (n > 0) hipLaunchKernelGGL(ingest_count, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, d_indptr, d_indices,
                                  n, include_self, d_rowlen, d_kmax);
    CM_HIP(h, hipGetLastError());
    return CM_OK;
}

cm_status cm_ingest_fill_run(cm_handle_t h, const int64_t *d_indptr, const int32_t *d_indices, const double *d_data, int64_t n,
                             int include_self, int kmax, int64_t *d_out_idx, double *d_out_dist) {
    if (kmax + 1 > IG_LONG_MAX)
        CM_FAIL(h, CM_E_UNSUPPORTED, "precomputed distances: a row has %d entries (maximum %d)", kmax, IG_LONG_MAX - 1);
    if (n > 0 && kmax > 0 && kmax + 1 <= IG_MAX)
        hipLaunchKernelGGL(ingest_fill, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, h->stream, d_indptr, d_indices, d_data, n,
                           include_self, kmax, d_out_idx, d_out_dist);
    else if (n > 0 && kmax > 0) {   // long rows: one workgroup per row, the row in dynamic LDS
        const int cap = kmax + 1;
        const size_t lds = (size_t)cap * 12 + 16;
        auto kern = ingest_fill_long;
        CM_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)std::min<int64_t>(n, 4096)), dim3(256), lds, h->stream, d_indptr, d_indices, d_data, n,
                           include_self, kmax, cap, d_out_idx, d_out_dist);
    }
    CM_HIP(h, hipGetLastError());
    return CM_OK;
}

// ------------------------------------------------------------------ E1: expression-transfer evaluation --
// Per-gene agreement between measured and imputed expression (reference evaluate.py:236-323 loops over genes in
// Python calling scipy per gene): Pearson r, RMSE of the z-scored vectors, Jensen-Shannon distance (base 10), over
// all cells or over the cells of one group.  Inputs are row-major float32 [n, G]; a thread owns one gene (coalesced
// row reads), a workgroup a chunk of rows; per-chunk float64 partial sums are combined in chunk order (deterministic).
// Two passes: sums (means, clipped sums), then CENTRED second moments -- a constant column gets variance exactly 0,
// as scipy's centred arithmetic does -- and the Jensen-Shannon terms.
namespace {
constexpr int EX_N1 = 5;   // pass 1: n, sum a, sum b, sum max(a,0), sum max(b,0)
constexpr int EX_N2 = 4;   // pass 2: sum (a-ma)^2, sum (b-mb)^2, sum (a-ma)(b-mb), JS terms

__global__ __launch_bounds__(256) void expr_pass1(const float *__restrict__ A, const float *__restrict__ B, int64_t n,
                                                  int64_t G, const int32_t *__restrict__ group, int sel,
                                                  int64_t rows_per_chunk, double *__restrict__ part) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk, r1 = r0 + rows_per_chunk < n ? r0 + rows_per_chunk : n;
    if (g >= G) return;
    double s[EX_N1] = {0, 0, 0, 0, 0};
    for (int64_t r = r0; r < r1; ++r) {
        if (sel >= 0 && group[r] != sel) continue;
        const double a = (double)A[r * G + g], b = (double)B[r * G + g];
        s[0] += 1.0; s[1] += a; s[2] += b;
        s[3] += a > 0.0 ? a : 0.0; s[4] += b > 0.0 ? b : 0.0;
    }
    double *o = part + ((int64_t)blockIdx.y * G + g) * EX_N1;
#pragma unroll
    for (int e = 0; e < EX_N1; ++e) o[e] = s[e];
}

template <int NS>
__global__ void expr_combine(const double *__restrict__ part, int64_t G, int chunks, double *__restrict__ tot) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    double s[NS];
#pragma unroll
    for (int e = 0; e < NS; ++e) s[e] = 0.0;
    for (int c = 0; c < chunks; ++c)
#pragma unroll
        for (int e = 0; e < NS; ++e) s[e] += part[((int64_t)c * G + g) * NS + e];
#pragma unroll
    for (int e = 0; e < NS; ++e) tot[g * NS + e] = s[e];
}

__global__ __launch_bounds__(256) void expr_pass2(const float *__restrict__ A, const float *__restrict__ B, int64_t n,
                                                  int64_t G, const int32_t *__restrict__ group, int sel,
                                                  int64_t rows_per_chunk, const double *__restrict__ tot1, int want_js,
                                                  double *__restrict__ part) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk, r1 = r0 + rows_per_chunk < n ? r0 + rows_per_chunk : n;
    if (g >= G) return;
    const double cnt = tot1[g * EX_N1], sa = tot1[g * EX_N1 + 3], sb = tot1[g * EX_N1 + 4];
    const double ma = cnt > 0.0 ? tot1[g * EX_N1 + 1] / cnt : 0.0, mb = cnt > 0.0 ? tot1[g * EX_N1 + 2] / cnt : 0.0;
    const bool js_ok = want_js && sa > 0.0 && sb > 0.0;
    double s[EX_N2] = {0, 0, 0, 0};
    for (int64_t r = r0; r < r1; ++r) {
        if (sel >= 0 && group[r] != sel) continue;
        const double a = (double)A[r * G + g], b = (double)B[r * G + g];
        const double da = a - ma, db = b - mb;
        s[0] = fma(da, da, s[0]); s[1] = fma(db, db, s[1]); s[2] = fma(da, db, s[2]);
        if (js_ok) {   // p log(p/m) + q log(q/m), p = a+/sum a+, q = b+/sum b+, m = (p+q)/2; 0 log 0 = 0
            const double p = (a > 0.0 ? a : 0.0) / sa, q = (b > 0.0 ? b : 0.0) / sb, m = 0.5 * (p + q);
            if (p > 0.0) s[3] += p * log(p / m);
            if (q > 0.0) s[3] += q * log(q / m);
        }
    }
    double *o = part + ((int64_t)blockIdx.y * G + g) * EX_N2;
#pragma unroll
    for (int e = 0; e < EX_N2; ++e) o[e] = s[e];
}

__global__ void expr_finish(const double *__restrict__ tot1, const double *__restrict__ tot2, int64_t G, int method,
                            float *__restrict__ out) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    const double n = tot1[g * EX_N1];
    double res = NAN;
    if (n > 0.0) {
        const double va = tot2[g * EX_N2] / n, vb = tot2[g * EX_N2 + 1] / n, cov = tot2[g * EX_N2 + 2] / n;  // population
        if (method == 0) {            // scipy.stats.pearsonr: nan for a constant input, clipped to [-1, 1]
            if (va > 0.0 && vb > 0.0 && n >= 2.0) {
                res = cov / (sqrt(va) * sqrt(vb));
                res = res > 1.0 ? 1.0 : (res < -1.0 ? -1.0 : res);
            }
        } else if (method == 1) {     // RMSE of the z-scored vectors, std == 0 -> 1 (evaluate.py:40-63)
            const double sda = va > 0.0 ? sqrt(va) : 1.0, sdb = vb > 0.0 ? sqrt(vb) : 1.0;
            const double ms = va / (sda * sda) + vb / (sdb * sdb) - 2.0 * cov / (sda * sdb);
            res = sqrt(ms > 0.0 ? ms : 0.0);
        } else {                      // scipy.spatial.distance.jensenshannon(p, q, base=10) on the clipped vectors
            if (tot1[g * EX_N1 + 3] > 0.0 && tot1[g * EX_N1 + 4] > 0.0) {
                const double js = tot2[g * EX_N2 + 3] / 2.302585092994046;   // ln 10
                res = sqrt(js > 0.0 ? js * 0.5 : 0.0);
            }
        }
    }
    out[g] = (float)res;
}

int expr_chunks(int64_t n) {
    int chunks = (int)((n + 4095) / 4096);
    return chunks < 1 ? 1 : (chunks > 512 ? 512 : chunks);
}
}  // namespace

cm_status cm_expr_metrics_run(cm_handle_t h, const float *d_a, const float *d_b, int64_t n, int64_t G, const int32_t *d_group,
                              int n_groups, int method, float *d_out) {
    if (n < 0 || G < 1 || method < 0 || method > 2 || n_groups < 0) CM_FAIL(h, CM_E_INVALID, "expression metrics: bad arguments");
    const int chunks = expr_chunks(n);
    const int64_t rpc = (n + chunks - 1) / chunks;
    double *part = (double *)cm_ws_alloc(h, sizeof(double) * (size_t)chunks * G * EX_N1);
    double *tot1 = (double *)cm_ws_alloc(h, sizeof(double) * (size_t)G * EX_N1);
    double *tot2 = (double *)cm_ws_alloc(h, sizeof(double) * (size_t)G * EX_N2);
    if (!part || !tot1 || !tot2) CM_FAIL(h, CM_E_NOMEM, "expression metrics: device workspace exhausted");
    const dim3 grid((unsigned)((G + 255) / 256), (unsigned)chunks), gg((unsigned)((G + 255) / 256));
    for (int sel = -1; sel < n_groups; ++sel) {   // row 0 of the output: all cells; row 1 + c: cells of group c
        if (sel >= 0 && !d_group) break;
        hipLaunchKernelGGL(expr_pass1, grid, dim3(256), 0, h->stream, d_a, d_b, n, G, d_group, sel, rpc, part);
        hipLaunchKernelGGL(expr_combine<EX_N1>, gg, dim3(256), 0, h->stream, part, G, chunks, tot1);
        hipLaunchKernelGGL(expr_pass2, grid, dim3(256), 0, h->stream, d_a, d_b, n, G, d_group, sel, rpc, tot1, method == 2 ? 1 : 0, part);
        hipLaunchKernelGGL(expr_combine<EX_N2>, gg, dim3(256), 0, h->stream, part, G, chunks, tot2);
        hipLaunchKernelGGL(expr_finish, gg, dim3(256), 0, h->stream, tot1, tot2, G, method, d_out + (int64_t)(sel + 1) * G);
    }
    CM_HIP(h, hipGetLastError());
    return CM_OK;
}

size_t cm_expr_metrics_ws_bytes(int64_t n, int64_t G) {
    return cm_align(sizeof(double) * (size_t)expr_chunks(n) * G * EX_N1) + 2 * cm_align(sizeof(double) * (size_t)G * EX_N1) + 4096;
}

// code-object preload (cm_create): one empty launch makes the runtime load this translation unit's kernels onto the
// device ahead of the first real call (the load used to land inside the first call's stage times)
static __global__ void cm_touch_graph() {}
void cm_preload_graph(hipStream_t st) { hipLaunchKernelGGL(cm_touch_graph, dim3(1), dim3(64), 0, st); }
