// Spearman agreement between measured and imputed expression, per gene.
// Reference: evaluate.py:276-277 (scipy.stats.spearmanr per gene over a cell mask) = Pearson correlation of the AVERAGE
// ranks of the two columns (tied values share the mean of their positions), nan when either column is constant.
//
// Per cell selection (all cells / one group) and per chunk of genes: the selected rows of both [n, G] matrices are
// gathered into gene-major segments, every segment is sorted (rocPRIM segmented radix sort, keys = values, payload =
// position in the selection), one wave per segment finds the tie runs of the sorted keys (wave scan with a carry), the
// average ranks are scattered back to selection order in float64 and a workgroup per gene correlates the two rank
// vectors (their mean is (m+1)/2 exactly).  Everything is deterministic: no atomics.
#include <cstring>

#include <rocprim/device/device_segmented_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "cm_internal.h"

namespace {

constexpr int64_t SP_EMAX = (int64_t)1 << 27;   // elements (cells x genes) ranked per chunk

struct SegOff {
    unsigned m;
    __host__ __device__ unsigned operator()(unsigned g) const { return g * m; }
};
using OffIt = rocprim::transform_iterator<rocprim::counting_iterator<unsigned>, SegOff, unsigned>;

// rows of group `sel`, in ascending order (one workgroup: ordered compaction with a running base)
__global__ __launch_bounds__(1024) void sp_select_rows(const int32_t *__restrict__ group, int64_t n, int sel,
                                                       int *__restrict__ rows, int *__restrict__ count) {
    __shared__ int s_scan[1024];
    __shared__ int s_base;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    for (int64_t t0 = 0; t0 < n; t0 += 1024) {
        const int64_t i = t0 + threadIdx.x;
        const int f = (i < n && group[i] == sel) ? 1 : 0;
        s_scan[threadIdx.x] = f;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const int v = (int)threadIdx.x >= o ? s_scan[threadIdx.x - o] : 0;
            __syncthreads();
            s_scan[threadIdx.x] += v;
            __syncthreads();
        }
        const int incl = s_scan[threadIdx.x], base = s_base;
        if (f) rows[base + incl - 1] = (int)i;
        __syncthreads();
        if (threadIdx.x == 1023) s_base = base + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *count = s_base;
}

// keys[g][i] = M[rows[i]][g0 + g] (-0 -> +0: the sort must see them as the tie scipy sees), vals[g][i] = i
__global__ __launch_bounds__(256) void sp_gather(const float *__restrict__ M, int64_t G, const int *__restrict__ rows,
                                                 unsigned m, int64_t g0, int Gc, float *__restrict__ keys,
                                                 int *__restrict__ vals) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const unsigned i0 = blockIdx.x * 32u;
    const int gb = blockIdx.y * 32;
    for (int r = ty; r < 32; r += 8) {
        const unsigned i = i0 + r;
        const int g = gb + tx;
        float v = 0.f;
        if (i < m && g < Gc) v = M[(int64_t)(rows ? rows[i] : (int)i) * G + g0 + g] + 0.0f;
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int g = gb + r;
        const unsigned i = i0 + tx;
        if (g < Gc && i < m) {
            keys[(size_t)g * m + i] = tile[tx][r];
            vals[(size_t)g * m + i] = (int)i;
        }
    }
}

// tie runs of a sorted segment, one wave per segment: start[i] = first position of i's run, runend[s] = last position
// of the run that starts at s
__global__ __launch_bounds__(256) void sp_runs(const float *__restrict__ skeys, unsigned m, int Gc, int *__restrict__ start,
                                               int *__restrict__ runend) {
    const int seg = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (seg >= Gc) return;
    const float *k = skeys + (size_t)seg * m;
    int *st = start + (size_t)seg * m, *re = runend + (size_t)seg * m;
    int carry_start = 0;
    float carry_key = 0.f;
    for (unsigned base = 0; base < m; base += 64) {
        const unsigned i = base + lane;
        const bool in = i < m;
        const float key = in ? k[i] : 0.f;
        float pk = __shfl_up(key, 1);
        if (lane == 0) pk = carry_key;
        const bool flag = in && (i == 0 || key != pk);
        int bs = flag ? (int)i : -1;
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(bs, o);
            if (lane >= o) bs = bs > v ? bs : v;
        }
        const int s = bs > carry_start ? bs : carry_start;
        int sp = __shfl_up(s, 1);
        if (lane == 0) sp = carry_start;
        if (flag && i > 0) re[sp] = (int)i - 1;
        if (in) st[i] = s;
        const unsigned left = m - base;
        const int last = left >= 64 ? 63 : (int)left - 1;
        carry_start = __shfl(s, last);
        carry_key = __shfl(key, last);
    }
    if (lane == 0) re[carry_start] = (int)m - 1;
}

__global__ void sp_ranks(const int *__restrict__ start, const int *__restrict__ runend, const int *__restrict__ svals,
                         unsigned m, size_t total, double *__restrict__ rank) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const size_t seg = t / m;
    const int s = start[t], e = runend[seg * m + s];
    rank[seg * m + svals[t]] = 0.5 * (double)(s + e) + 1.0;
}

__global__ __launch_bounds__(256) void sp_corr(const double *__restrict__ ra, const double *__restrict__ rb, unsigned m,
                                               float *__restrict__ out) {
    __shared__ double s_r[3][256];
    const double mu = 0.5 * ((double)m + 1.0);
    const double *a = ra + (size_t)blockIdx.x * m, *b = rb + (size_t)blockIdx.x * m;
    double saa = 0.0, sbb = 0.0, sab = 0.0;
    for (unsigned i = threadIdx.x; i < m; i += 256) {
        const double x = a[i] - mu, y = b[i] - mu;
        saa = fma(x, x, saa); sbb = fma(y, y, sbb); sab = fma(x, y, sab);
    }
    s_r[0][threadIdx.x] = saa; s_r[1][threadIdx.x] = sbb; s_r[2][threadIdx.x] = sab;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
            for (int q = 0; q < 3; ++q) s_r[q][threadIdx.x] += s_r[q][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double va = s_r[0][0], vb = s_r[1][0];
        out[blockIdx.x] = (va > 0.0 && vb > 0.0) ? (float)(s_r[2][0] / sqrt(va * vb)) : __builtin_nanf("");
    }
}

__global__ void sp_fill_nan(float *__restrict__ p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = __builtin_nanf("");
}

inline int64_t sp_chunk_genes(int64_t m, int64_t G) {
    int64_t gc = SP_EMAX / (m > 0 ? m : 1);
    if (gc < 1) gc = 1;
    return gc < G ? gc : G;
}

inline size_t sp_sort_temp(unsigned size, unsigned segments, unsigned m) {
    size_t bytes = 0;
    OffIt b0(rocprim::counting_iterator<unsigned>(0), SegOff{m}), b1(rocprim::counting_iterator<unsigned>(1), SegOff{m});
    (void)rocprim::segmented_radix_sort_pairs(nullptr, bytes, (const float *)nullptr, (float *)nullptr, (const int *)nullptr,
                                              (int *)nullptr, size, segments, b0, b1);
    return bytes;
}

}  // namespace

size_t cm_expr_spearman_ws_bytes(int64_t n, int64_t G) {
    if (n < 1) n = 1;
    const int64_t gc = sp_chunk_genes(n, G);
    const size_t E = (size_t)gc * (size_t)n;
    return 4 * cm_align(4 * E) + 2 * cm_align(8 * E) + cm_align(sp_sort_temp((unsigned)E, (unsigned)gc, (unsigned)n)) + cm_align(4 * (size_t)n) + 65536;
}

cm_status cm_expr_spearman_run(cm_handle_t h, const float *d_a, const float *d_b, int64_t n, int64_t G, const int32_t *d_group,
                               int n_groups, float *d_out) {
    if (n < 0 || G < 1 || n_groups < 0) CM_FAIL(h, CM_E_INVALID, "expression metrics: bad arguments");
    if (n > 0x7fffffffLL) CM_FAIL(h, CM_E_UNSUPPORTED, "spearman: more than 2^31 cells");
    hipStream_t st = h->stream;
    const int64_t gc_max = sp_chunk_genes(n, G);
    const size_t E = (size_t)gc_max * (size_t)(n > 0 ? n : 1);
    if (E > 0xffffffffull) CM_FAIL(h, CM_E_UNSUPPORTED, "spearman: a single gene column exceeds the sort's 2^32 element limit");
    float *k_in = (float *)cm_ws_alloc(h, 4 * E), *k_out = (float *)cm_ws_alloc(h, 4 * E);
    int *v_in = (int *)cm_ws_alloc(h, 4 * E), *v_out = (int *)cm_ws_alloc(h, 4 * E);
    double *rk_a = (double *)cm_ws_alloc(h, 8 * E), *rk_b = (double *)cm_ws_alloc(h, 8 * E);
    const size_t temp_cap = sp_sort_temp((unsigned)E, (unsigned)gc_max, (unsigned)(n > 0 ? n : 1));
    void *temp = cm_ws_alloc(h, temp_cap ? temp_cap : 16);
    int *rows = (int *)cm_ws_alloc(h, 4 * (size_t)(n > 0 ? n : 1));
    int *d_count = (int *)cm_ws_alloc(h, 64);
    if (!k_in || !k_out || !v_in || !v_out || !rk_a || !rk_b || !temp || !rows || !d_count)
        CM_FAIL(h, CM_E_NOMEM, "expression metrics: device workspace exhausted");
    for (int sel = -1; sel < n_groups; ++sel) {   // row 0 of the output: all cells; row 1 + c: cells of group c
        if (sel >= 0 && !d_group) break;
        float *out_row = d_out + (int64_t)(sel + 1) * G;
        int64_t m = n;
        const int *rsel = nullptr;
        if (sel >= 0) {
            hipLaunchKernelGGL(sp_select_rows, dim3(1), dim3(1024), 0, st, d_group, n, sel, rows, d_count);
            CM_HIP(h, hipMemcpyAsync(h->d_nflag_pinned + 8, d_count, sizeof(int), hipMemcpyDeviceToHost, st));
            CM_HIP(h, hipStreamSynchronize(st));
            m = h->d_nflag_pinned[8];
            rsel = rows;
        }
        if (m < 2) {
            hipLaunchKernelGGL(sp_fill_nan, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, st, out_row, G);
            continue;
        }
        const int64_t gstep = gc_max;   // (buffers and the sort's temporary storage are sized for gc_max segments of n rows)
        for (int64_t g0 = 0; g0 < G; g0 += gstep) {
            const int Gc = (int)(G - g0 < gstep ? G - g0 : gstep);
            const size_t total = (size_t)Gc * (size_t)m;
            OffIt b0(rocprim::counting_iterator<unsigned>(0), SegOff{(unsigned)m}), b1(rocprim::counting_iterator<unsigned>(1), SegOff{(unsigned)m});
            for (int side = 0; side < 2; ++side) {
                hipLaunchKernelGGL(sp_gather, dim3((unsigned)((m + 31) / 32), (unsigned)((Gc + 31) / 32)), dim3(256), 0, st,
                                   side ? d_b : d_a, G, rsel, (unsigned)m, g0, Gc, k_in, v_in);
                size_t tb = temp_cap;
                CM_HIP(h, rocprim::segmented_radix_sort_pairs(temp, tb, (const float *)k_in, k_out, (const int *)v_in, v_out,
                                                              (unsigned)total, (unsigned)Gc, b0, b1, 0, 32, st));
                int *start = reinterpret_cast<int *>(k_in), *runend = v_in;   // (the unsorted copies are no longer needed)
                hipLaunchKernelGGL(sp_runs, dim3((unsigned)((Gc + 3) / 4)), dim3(256), 0, st, (const float *)k_out, (unsigned)m, Gc, start, runend);
                hipLaunchKernelGGL(sp_ranks, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const int *)start, (const int *)runend,
                                   (const int *)v_out, (unsigned)m, total, side ? rk_b : rk_a);
            }
            hipLaunchKernelGGL(sp_corr, dim3((unsigned)Gc), dim3(256), 0, st, (const double *)rk_a, (const double *)rk_b, (unsigned)m, out_row + g0);
        }
    }
    CM_HIP(h, hipGetLastError());
    return CM_OK;
}


// ------------------------------------------------------------------ precomputed-distance ingest: rows beyond the in-LDS sorts
// utils.py:129-219 sets no limit on the stored neighbours of a cell (a dense distance matrix has n of them); rows of up to
// 12 287 entries are rank-sorted in LDS (graph_kernels.hip: ingest_fill / ingest_fill_long).  Longer rows (round 4): the kept
// entries of every row are compacted in stored order into staging arrays (value, column), a STABLE segmented radix sort by
// value (rocPRIM; ties keep their stored order, an already non-decreasing row comes out as stored -- the rule of the LDS
// kernels) writes them straight into the output arrays, a last kernel pads the tails with -1 / inf.  Row batches keep the
// sort below 2^31 elements.
namespace {
__global__ __launch_bounds__(256) void ig_gather(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                 const double *__restrict__ data, int64_t row0, int rows, int64_t n_total, int include_self,
                                                 int kmax, double *__restrict__ k_in, int64_t *__restrict__ v_in, unsigned *__restrict__ ends) {
    // one wave per row (rows are long: the loop over 64-entry pieces is the work)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + w;
    if (r >= rows) return;
    const int64_t i = row0 + r;
    (void)n_total;
    const int64_t lo = indptr[i], hi = indptr[i + 1];
    double *kd = k_in + (size_t)r * kmax;
    int64_t *vd = v_in + (size_t)r * kmax;
    int m = 0;
    bool has = false;
    for (int64_t base = lo; base < hi; base += 64) {
        const int64_t p = base + lane;
        const bool in = p < hi;
        const int c = in ? indices[p] : -1;
        has = has || (in && c == (int32_t)i);
        const bool keep = in && !(include_self == 0 && c == (int32_t)i);
        const unsigned long long b = __ballot(keep);
        if (keep) {
            const int pos = m + __popcll(b & ((1ull << lane) - 1ull));
            kd[pos] = data[p];
            vd[pos] = c;
        }
        m += __popcll(b);
    }
    if (include_self == 1 && __ballot(has) == 0ull) {
        if (lane == 0) { kd[m] = 0.0; vd[m] = i; }
        ++m;
    }
    if (lane == 0) { ends[r] = (unsigned)((size_t)r * kmax + m); ends[rows + r] = (unsigned)((size_t)r * kmax); }   // [0, rows): ends, [rows, 2 rows): begins
}
__global__ void ig_pad(const unsigned *__restrict__ ends, int rows, int kmax, int64_t *__restrict__ out_idx, double *__restrict__ out_dist) {
    const int r = blockIdx.y;
    const unsigned m = ends[r] - (unsigned)((size_t)r * kmax);
    for (unsigned s_ = m + blockIdx.x * blockDim.x + threadIdx.x; s_ < (unsigned)kmax; s_ += gridDim.x * blockDim.x) {
        out_idx[(size_t)r * kmax + s_] = -1;
        out_dist[(size_t)r * kmax + s_] = INFINITY;
    }
}
inline int64_t ig_batch_rows(int64_t n, int kmax) {   // (fewer than 2^31 elements per sort, at most 32 768 rows per launch: grid y)
    return std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(n, 32768), ((int64_t)1 << 31) / std::max(kmax, 1) - 1));
}
inline size_t ig_sort_temp(unsigned size, unsigned segments, unsigned m) {
    size_t bytes = 0;
    (void)m;
    (void)rocprim::segmented_radix_sort_pairs(nullptr, bytes, (const double *)nullptr, (double *)nullptr, (const int64_t *)nullptr,
                                              (int64_t *)nullptr, size, segments, (const unsigned *)nullptr, (const unsigned *)nullptr);
    return bytes;
}
}  // namespace

// device bytes the long-row ingest needs besides its outputs: staging (value + column per output slot of a batch), the
// batch's segment ends, the sort's temporary storage
size_t cm_ingest_huge_ws_bytes(int64_t n, int kmax) {
    const int64_t B = ig_batch_rows(n, kmax);
    const size_t E = (size_t)B * (size_t)kmax;
    return 2 * cm_align(8 * E) + cm_align(8 * (size_t)B) + cm_align(ig_sort_temp((unsigned)E, (unsigned)B, (unsigned)kmax)) + 4096;
}

cm_status cm_ingest_fill_huge_run(cm_handle_t h, const int64_t *d_indptr, const int32_t *d_indices, const double *d_data, int64_t n,
                                  int include_self, int kmax, void *d_ws, size_t ws_bytes, int64_t *d_out_idx, double *d_out_dist) {
    if (n < 1 || kmax < 1) return CM_OK;
    if (ws_bytes < cm_ingest_huge_ws_bytes(n, kmax)) CM_FAIL(h, CM_E_INTERNAL, "precomputed distances: workspace too small");
    const int64_t B = ig_batch_rows(n, kmax);
    const size_t E = (size_t)B * (size_t)kmax;
    char *p = (char *)d_ws;
    double *k_in = (double *)p; p += cm_align(8 * E);
    int64_t *v_in = (int64_t *)p; p += cm_align(8 * E);
    unsigned *ends = (unsigned *)p; p += cm_align(8 * (size_t)B);   // ends + begins of the batch's segments
    void *temp = p;
    const size_t temp_cap = ig_sort_temp((unsigned)E, (unsigned)B, (unsigned)kmax);
    for (int64_t r0 = 0; r0 < n; r0 += B) {
        const int rows = (int)std::min<int64_t>(B, n - r0);
        hipLaunchKernelGGL(ig_gather, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, h->stream, d_indptr, d_indices, d_data, r0, rows, n, include_self,
                           kmax, k_in, v_in, ends);
        size_t tb = temp_cap;
        CM_HIP(h, rocprim::segmented_radix_sort_pairs(temp, tb, (const double *)k_in, d_out_dist + (size_t)r0 * kmax, (const int64_t *)v_in,
                                                      d_out_idx + (size_t)r0 * kmax, (unsigned)((size_t)rows * kmax), (unsigned)rows,
                                                      (const unsigned *)(ends + rows), (const unsigned *)ends, 0, 64, h->stream));
        hipLaunchKernelGGL(ig_pad, dim3(8, (unsigned)rows), dim3(256), 0, h->stream, (const unsigned *)ends, rows, kmax,
                           d_out_idx + (size_t)r0 * kmax, d_out_dist + (size_t)r0 * kmax);
    }
    CM_HIP(h, hipGetLastError());
    return CM_OK;
}
