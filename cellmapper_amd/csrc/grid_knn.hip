// Uniform-grid candidate search for 1 <= d <= 3 (spatial coordinates; BASELINE config 5: 4 M cells, d = 2, k = 90).
//
// A brute-force filter visits n_ref rows per query; on a grid whose cells hold ~k/4 points a query touches a few hundred.
// The stage produces exactly what the brute-force filters produce -- K' candidate rows per query and a threshold tau
// with the invariant "every non-candidate has fp32 filter value (direct squared distance) > tau" -- so the float64
// re-rank, the certificate and the escalation of uncertified queries (knn_kernels.hip) are shared:
//   1. reference points are bucketed by cell (counting sort: histogram, exclusive scan, scatter of fp32 coordinates and
//      original row numbers);
//   2. one wave per query grows a cube of cells around the query's cell until it holds >= k points (cell occupancies come
//      from the scan, two loads per cell row), evaluates those points and selects the k-th smallest distance d_k (radix
//      select over LDS): an upper bound of the true k-th neighbour distance;
//   3. it then collects every point within tau = d_k (1 + margin) from the cube that contains that ball -- by construction
//      the complete set of rows with filter value <= tau.
// Queries whose cube holds more than GRID_CAP points (dense clusters) or whose ball holds more than K' rows (ties) get
// tau = -inf: the re-rank flags them and they continue on the brute-force direct-difference kernel.
#include "cm_internal.h"

namespace {

constexpr int GRID_CAP = 2048;   // points a wave evaluates in step 2 (LDS: 4 waves x 8 KB)

struct GridInfo {
    float lo[3];
    float h, inv_h;
    int g[3];
    int d;
};

__device__ __forceinline__ unsigned f2ord(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

template <typename T>
__global__ __launch_bounds__(256) void grid_bbox(const T *__restrict__ X, int64_t n, int d, unsigned *__restrict__ bb) {
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        for (int c = 0; c < d; ++c) {
            const float v = (float)X[i * d + c];
            mn[c] = fminf(mn[c], v);
            mx[c] = fmaxf(mx[c], v);
        }
    for (int c = 0; c < d; ++c) {
        for (int o = 32; o > 0; o >>= 1) { mn[c] = fminf(mn[c], __shfl_xor(mn[c], o)); mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], o)); }
        if ((threadIdx.x & 63) == 0) { atomicMin(&bb[c], f2ord(mn[c])); atomicMax(&bb[3 + c], f2ord(mx[c])); }
    }
}

__global__ void grid_init_bbox(unsigned *bb) {
    if (threadIdx.x < 3) { bb[threadIdx.x] = 0xffffffffu; bb[3 + threadIdx.x] = 0u; }
}

__global__ void grid_setup(const unsigned *__restrict__ bb, int d, double ncell_target, int64_t ncell_max, GridInfo *gi) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float lo[3] = {0, 0, 0}, ex[3] = {0, 0, 0};
    double vol = 1.0;
    int nd = 0;   // dimensions with a non-degenerate extent
    for (int c = 0; c < d; ++c) {
        lo[c] = ord2f(bb[c]);
        ex[c] = ord2f(bb[3 + c]) - lo[c];
        if (ex[c] > 0.f) { vol *= (double)ex[c]; ++nd; }
    }
    double h = nd > 0 ? pow(vol / ncell_target, 1.0 / nd) : 1.0;
    if (!(h > 0.0)) h = 1.0;
    int g[3] = {1, 1, 1};
    for (int it = 0; it < 200; ++it) {
        double prod = 1.0;
        for (int c = 0; c < d; ++c) {
            double gc = floor((double)ex[c] / h) + 1.0;
            if (gc > 2.0e9) gc = 2.0e9;
            g[c] = (int)gc;
            prod *= gc;
            if (gc > 65536.0) prod = 1.0e300;   // <= 2^16 cells per dimension: fp32 cell coordinates stay good to 0.016 cells
        }
        if (prod <= (double)ncell_max) break;
        h *= 1.1;
    }
    for (int c = 0; c < 3; ++c) { gi->lo[c] = lo[c]; gi->g[c] = c < d ? g[c] : 1; }
    gi->h = (float)h;
    gi->inv_h = (float)(1.0 / h);
    gi->d = d;
}

__device__ __forceinline__ void cell_of_point(const GridInfo &gi, const float *p, int d, int *cc) {
    cc[0] = cc[1] = cc[2] = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {   // (constant indices: the three-element arrays stay in registers, not in scratch)
        if (c < d) {
            int v = (int)floorf((p[c] - gi.lo[c]) * gi.inv_h);
            cc[c] = v < 0 ? 0 : (v >= gi.g[c] ? gi.g[c] - 1 : v);
        }
    }
}
__device__ __forceinline__ int64_t cell_linear(const GridInfo &gi, int x, int y, int z) {
    return ((int64_t)z * gi.g[1] + y) * gi.g[0] + x;
}

template <typename T>
__global__ void grid_count(const T *__restrict__ X, int64_t n, int d, const GridInfo *__restrict__ gip, int *__restrict__ cell_of,
                           int *__restrict__ counts) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const GridInfo gi = *gip;
    float p[3] = {0, 0, 0};
    for (int c = 0; c < d; ++c) p[c] = (float)X[i * d + c];
    int cc[3];
    cell_of_point(gi, p, d, cc);
    const int64_t lin = cell_linear(gi, cc[0], cc[1], cc[2]);
    cell_of[i] = (int)lin;
    atomicAdd(&counts[lin], 1);
}

template <typename T>
__global__ void grid_scatter(const T *__restrict__ X, int64_t n, int d, const int *__restrict__ cell_of,
                             const int64_t *__restrict__ cell_start, int *__restrict__ cursor, float *__restrict__ sxyz,
                             int *__restrict__ sidx) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = cell_of[i];
    const int64_t pos = cell_start[c] + atomicAdd(&cursor[c], 1);
    for (int e = 0; e < d; ++e) sxyz[pos * d + e] = (float)X[i * d + e];
    sidx[pos] = (int)i;
}

// wave-wide: k-th smallest (1-based) of the n non-negative floats in v[] (LDS); radix select on the bit patterns.
// wave_kth_regs: the same selection with the values read ONCE into NR registers per lane (n <= 64 NR): the 31 bit passes are
// then compare / ballot / count only -- the LDS version re-reads every value in every pass (round 3: a third of grid_query).
template <int NR>
__device__ __forceinline__ float wave_kth_regs(const float *v, int n, int k, int lane) {
    unsigned key[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) { const int i = r * 64 + lane; key[r] = i < n ? __float_as_uint(v[i]) : 0xffffffffu; }
    // invariant: `below` keys are smaller than `prefix`; the k-th smallest has the high bits of `prefix`.  One unsigned
    // compare per register and bit: the keys with those high bits and a 0 at `bit` are the ones below prefix | 1 << bit
    // that are not below prefix.  (Sign bit 0 for every entry; a padding key is 0xffffffff: never below anything tested.)
    unsigned prefix = 0u;
    int below = 0;
    for (int bit = 30; bit >= 0; --bit) {
        const unsigned pivot = prefix | (1u << bit);
        int c = 0;
#pragma unroll
        for (int r = 0; r < NR; ++r) c += __popcll(__ballot(key[r] < pivot));
        if (k > c) { prefix = pivot; below = c; }   // fewer than k keys below the pivot: the k-th has this bit set
    }
    (void)below;
    return __uint_as_float(prefix);
}
__device__ __forceinline__ float wave_kth_lds(const float *v, int n, int k, int lane);
__device__ __forceinline__ float wave_kth(const float *v, int n, int k, int lane) {
    if (n <= 256) return wave_kth_regs<4>(v, n, k, lane);
    if (n <= 512) return wave_kth_regs<8>(v, n, k, lane);
    if (n <= 1024) return wave_kth_regs<16>(v, n, k, lane);
    return wave_kth_lds(v, n, k, lane);
}
__device__ __forceinline__ float wave_kth_lds(const float *v, int n, int k, int lane) {
    unsigned prefix = 0u;
    int remaining = k;
    for (int bit = 30; bit >= 0; --bit) {   // sign bit is 0 for every entry
        const unsigned himask = ~((2u << bit) - 1u);
        int cnt0 = 0;
        for (int base = 0; base < n; base += 64) {
            const int i = base + lane;
            const unsigned key = i < n ? __float_as_uint(v[i]) : 0xffffffffu;
            const bool m = i < n && ((key & himask) == prefix) && !((key >> bit) & 1u);
            cnt0 += __popcll(__ballot(m));
        }
        if (remaining > cnt0) { prefix |= (1u << bit); remaining -= cnt0; }
    }
    return __uint_as_float(prefix);
}

template <typename T>
__global__ __launch_bounds__(256) void grid_query(const T *__restrict__ Q, int64_t nq, int d, const int *__restrict__ rowmap,
                                                  const GridInfo *__restrict__ gip, const int64_t *__restrict__ cell_start,
                                                  const float *__restrict__ sxyz, const int *__restrict__ sidx, int k, int Kc,
                                                  const unsigned *__restrict__ gmax, int is_f64, int *__restrict__ cand,
                                                  float *__restrict__ cthr, unsigned *__restrict__ dbgc) {
    __shared__ float s_v[4][GRID_CAP];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 4 + w;
    if (i >= nq) return;
    const GridInfo gi = *gip;
    const int64_t io = rowmap ? (int64_t)rowmap[i] : i;
    float q[3] = {0, 0, 0};
#pragma unroll
    for (int c = 0; c < 3; ++c)
        if (c < d) q[c] = (float)Q[io * d + c];
    // squared distance to stored point p, coordinates in order (unrolled over the at most three dimensions: `q` indexed by a
    // run-time c lived in scratch memory, 40 bytes per lane, and every point evaluation went through it)
    auto dist2 = [&](int64_t p) {
        float s_ = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c)
            if (c < d) { const float t = q[c] - sxyz[p * d + c]; s_ = fmaf(t, t, s_); }
        return s_;
    };
    int cc[3];
    cell_of_point(gi, q, d, cc);
    int *crow = cand + i * (int64_t)Kc;
    const int gmaxdim = max(gi.g[0], max(gi.g[1], gi.g[2]));

    // ---- step 2a: smallest cube of cells around the query's cell with >= k points
    int r = 0, cnt = 0;
    for (;; ++r) {
        const int y0 = max(cc[1] - r, 0), y1 = min(cc[1] + r, gi.g[1] - 1), z0 = max(cc[2] - r, 0), z1 = min(cc[2] + r, gi.g[2] - 1);
        const int x0 = max(cc[0] - r, 0), x1 = min(cc[0] + r, gi.g[0] - 1);
        const int ny = y1 - y0 + 1, nrows = ny * (z1 - z0 + 1);
        int c = 0;
        for (int t = lane; t < nrows; t += 64) {
            const int z = z0 + t / ny, y = y0 + t % ny;
            c += (int)(cell_start[cell_linear(gi, x1, y, z) + 1] - cell_start[cell_linear(gi, x0, y, z)]);
        }
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
        cnt = c;
        if (cnt >= k || r >= gmaxdim) break;
    }
    bool give_up = cnt < k || cnt > GRID_CAP;
    float tau = -INFINITY;
    int ncand = 0;
    if (!give_up) {
        // ---- step 2b: distances of the cube's points, k-th smallest
        const int y0 = max(cc[1] - r, 0), y1 = min(cc[1] + r, gi.g[1] - 1), z0 = max(cc[2] - r, 0), z1 = min(cc[2] + r, gi.g[2] - 1);
        const int x0 = max(cc[0] - r, 0), x1 = min(cc[0] + r, gi.g[0] - 1);
        int base = 0;
        for (int z = z0; z <= z1; ++z)
            for (int y = y0; y <= y1; ++y) {
                const int64_t p0 = cell_start[cell_linear(gi, x0, y, z)], p1 = cell_start[cell_linear(gi, x1, y, z) + 1];
                for (int64_t p = p0 + lane; p < p1; p += 64) {
                    s_v[w][base + (int)(p - p0)] = dist2(p);
                }
                base += (int)(p1 - p0);
            }
        __builtin_amdgcn_wave_barrier();
        const float dk = wave_kth(s_v[w], cnt, k, lane);
        // float64 coordinates were rounded to fp32: widen by what that can move a squared distance (knn_rerank's bound)
        const float dlt = is_f64 ? 2.0f * 5.9604645e-8f * __uint_as_float(gmax[4]) : 0.f;
        tau = dk * (1.0f + 1.0e-5f) + 4.04f * (2.0f * dlt * sqrtf((float)d * dk) + (float)d * dlt * dlt) + 1.0e-30f;
        // ---- step 3: d_k may overestimate the k-th neighbour distance (the cube of step 2 need not contain the whole
        // ball); the cube that DOES contain the ball of radius sqrt(tau) lies at most r2 cells away in every direction.
        // Gather the distances <= tau from it, take the k-th smallest -- now the filter's true k-th value -- and emit
        // everything within that (a handful beyond k: ties and the margin).
        // (+0.02: two fp32 roundings per cell coordinate, at most 2 * 2^16 * 2^-23 cells)
        const int r2 = (int)(sqrtf(tau) * gi.inv_h * 1.000001f + 0.02f) + 1;
        const int Y0 = max(cc[1] - r2, 0), Y1 = min(cc[1] + r2, gi.g[1] - 1), Z0 = max(cc[2] - r2, 0), Z1 = min(cc[2] + r2, gi.g[2] - 1);
        const int X0 = max(cc[0] - r2, 0), X1 = min(cc[0] + r2, gi.g[0] - 1);
        int m = 0;
        for (int z = Z0; z <= Z1 && !give_up; ++z)
            for (int y = Y0; y <= Y1 && !give_up; ++y) {
                const int64_t p0 = cell_start[cell_linear(gi, X0, y, z)], p1 = cell_start[cell_linear(gi, X1, y, z) + 1];
                for (int64_t pb = p0; pb < p1; pb += 64) {
                    const int64_t p = pb + lane;
                    float s = INFINITY;
                    if (p < p1) s = dist2(p);
                    const bool in = s <= tau;
                    const unsigned long long b = __ballot(in);
                    const int pos = m + __popcll(b & ((1ull << lane) - 1ull));
                    if (in && pos < GRID_CAP) s_v[w][pos] = s;
                    m += __popcll(b);
                }
                if (m > GRID_CAP) give_up = true;
            }
        if (!give_up) {
            __builtin_amdgcn_wave_barrier();
            const float dk2 = wave_kth(s_v[w], m, k, lane);   // m >= k: the k points behind d_k are among them
            tau = dk2 * (1.0f + 1.0e-5f) + 4.04f * (2.0f * dlt * sqrtf((float)d * dk2) + (float)d * dlt * dlt) + 1.0e-30f;
            for (int z = Z0; z <= Z1 && !give_up; ++z)
                for (int y = Y0; y <= Y1 && !give_up; ++y) {
                    const int64_t p0 = cell_start[cell_linear(gi, X0, y, z)], p1 = cell_start[cell_linear(gi, X1, y, z) + 1];
                    for (int64_t pb = p0; pb < p1; pb += 64) {
                        const int64_t p = pb + lane;
                        bool in = false;
                        if (p < p1) in = dist2(p) <= tau;
                        const unsigned long long b = __ballot(in);
                        const int pos = ncand + __popcll(b & ((1ull << lane) - 1ull));
                        if (in && pos < Kc) crow[pos] = sidx[p];
                        ncand += __popcll(b);
                    }
                    if (ncand > Kc) give_up = true;   // more rows within tau than the candidate list holds (ties / duplicates)
                }
        }
    }
    if (give_up && lane == 0 && dbgc) atomicAdd(&dbgc[cnt < k ? 0 : (cnt > GRID_CAP ? 1 : 2)], 1u);
    if (give_up) { tau = -INFINITY; ncand = 0; }
    for (int s = ncand + lane; s < Kc; s += 64) crow[s] = -1;
    if (lane == 0) cthr[i] = tau;
}

}  // namespace

size_t cm_grid_ws_bytes(int64_t n_ref, int64_t ncell_max) {
    return cm_align(sizeof(int) * (size_t)n_ref) * 2 + cm_align(sizeof(float) * (size_t)n_ref * 3) + cm_align(sizeof(int) * (size_t)(ncell_max + 1)) * 2 +
           cm_align(sizeof(int64_t) * (size_t)(ncell_max + 2)) + cm_scan_ws_bytes(ncell_max + 1) + 4096;
}

int64_t cm_grid_ncell_max(int64_t n_ref, int k) {
    int64_t target = n_ref / (k / 4 > 1 ? k / 4 : 1);
    if (target < 1) target = 1;
    if (target > (1 << 25)) target = 1 << 25;
    return target * 2 + 1024;   // room for the rounding of the per-dimension cell counts
}

// Builds the grid over the reference rows.  All buffers come from the handle's workspace (valid until the next begin_call).
template <typename T>
cm_status cm_grid_build(cm_handle_t h, const T *d_ref, int64_t n_ref, int d, int k, void **gi_out, int64_t **cell_start_out,
                        float **sxyz_out, int **sidx_out) {
    hipStream_t st = h->stream;
    const int64_t ncell_max = cm_grid_ncell_max(n_ref, k);
    const double target = (double)((ncell_max - 1024) / 2);
    unsigned *bb = (unsigned *)cm_ws_alloc(h, 64);
    GridInfo *gi = (GridInfo *)cm_ws_alloc(h, sizeof(GridInfo));
    int *cell_of = (int *)cm_ws_alloc(h, sizeof(int) * (size_t)n_ref);
    int *counts = (int *)cm_ws_alloc(h, sizeof(int) * (size_t)(ncell_max + 1));
    int *cursor = (int *)cm_ws_alloc(h, sizeof(int) * (size_t)(ncell_max + 1));
    int64_t *cell_start = (int64_t *)cm_ws_alloc(h, sizeof(int64_t) * (size_t)(ncell_max + 2));
    float *sxyz = (float *)cm_ws_alloc(h, sizeof(float) * (size_t)n_ref * d);
    int *sidx = (int *)cm_ws_alloc(h, sizeof(int) * (size_t)n_ref);
    if (!bb || !gi || !cell_of || !counts || !cursor || !cell_start || !sxyz || !sidx)
        CM_FAIL(h, CM_E_NOMEM, "grid k-NN: device workspace exhausted");
    hipLaunchKernelGGL(grid_init_bbox, dim3(1), dim3(64), 0, st, bb);
    hipLaunchKernelGGL(grid_bbox<T>, dim3(1024), dim3(256), 0, st, d_ref, n_ref, d, bb);
    hipLaunchKernelGGL(grid_setup, dim3(1), dim3(64), 0, st, bb, d, target, ncell_max, gi);
    CM_HIP(h, hipMemsetAsync(counts, 0, sizeof(int) * (size_t)(ncell_max + 1), st));
    CM_HIP(h, hipMemsetAsync(cursor, 0, sizeof(int) * (size_t)(ncell_max + 1), st));
    hipLaunchKernelGGL(grid_count<T>, dim3((unsigned)((n_ref + 255) / 256)), dim3(256), 0, st, d_ref, n_ref, d, gi, cell_of, counts);
    CM_TRY(cm_exclusive_scan_i32_to_i64(h, counts, ncell_max + 1, cell_start));
    hipLaunchKernelGGL(grid_scatter<T>, dim3((unsigned)((n_ref + 255) / 256)), dim3(256), 0, st, d_ref, n_ref, d, cell_of, cell_start,
                       cursor, sxyz, sidx);
    CM_HIP(h, hipGetLastError());
    *gi_out = gi; *cell_start_out = cell_start; *sxyz_out = sxyz; *sidx_out = sidx;
    return CM_OK;
}

template <typename T>
cm_status cm_grid_query(cm_handle_t h, const T *d_qry, int64_t nq, int d, const int *rowmap, const void *gi,
                        const int64_t *cell_start, const float *sxyz, const int *sidx, int k, int Kc, const unsigned *gmax,
                        int *cand, float *cthr) {
    if (nq > 0)
        hipLaunchKernelGGL(grid_query<T>, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, h->stream, d_qry, nq, d, rowmap,
                           (const GridInfo *)gi, cell_start, sxyz, sidx, k, Kc, gmax, sizeof(T) == 8 ? 1 : 0, cand, cthr,
                           cm_get_knobs().dbg ? const_cast<unsigned *>(gmax) + 12 : (unsigned *)nullptr);
    CM_HIP(h, hipGetLastError());
    return CM_OK;
}

template cm_status cm_grid_build<float>(cm_handle_t, const float *, int64_t, int, int, void **, int64_t **, float **, int **);
template cm_status cm_grid_build<double>(cm_handle_t, const double *, int64_t, int, int, void **, int64_t **, float **, int **);
template cm_status cm_grid_query<float>(cm_handle_t, const float *, int64_t, int, const int *, const void *, const int64_t *,
                                        const float *, const int *, int, int, const unsigned *, int *, float *);
template cm_status cm_grid_query<double>(cm_handle_t, const double *, int64_t, int, const int *, const void *, const int64_t *,
                                         const float *, const int *, int, int, const unsigned *, int *, float *);
