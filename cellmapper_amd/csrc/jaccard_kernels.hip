// Shared-neighbour counts for the jaccard / hnoca mapping methods.
// Reference: CellMapper.compute_mapping_matrix, /root/reference/src/cellmapper/model/cellmapper.py:287-301:
//     xx, yy, xy, yx = knn.get_adjacency_matrices()            (0/1 CSR adjacencies, knn.py:228-266,467-483)
//     jaccard = (yx @ xx.T) + (yy @ xy.T)                       (scipy SpGEMM)
//     jaccard.data /= 4k - jaccard.data      |  hnoca: (data / (2k - data))^2
// J[i, j] = |N_yx(i) ∩ N_xx(j)| + |N_yy(i) ∩ N_xy(j)|  (query i, reference j).  GPU formulation: build the transposed
// neighbour lists T_xx[r] = {j : r in N_xx(j)}, T_xy[q] = {j : q in N_xy(j)} with a counting sort, then for every query
// row concatenate T_xx[r] over r in N_yx(i) and T_xy[q] over q in N_yy(i), sort the multiset (bitonic, LDS) and
// run-length encode it: distinct values are the columns (ascending = canonical CSR), run lengths are the counts.
#include "cm_internal.h"

#include <math.h>
#include <algorithm>

namespace {

constexpr int SN_CAP = 8192;  // multiset entries per query row handled in LDS (32 KB)

__global__ void t_count(const int64_t *__restrict__ idx, int64_t n, int *__restrict__ deg, int64_t ncols) {
    int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    int64_t c = idx[g];
    if (c >= 0 && c < ncols) atomicAdd(&deg[c], 1);
}
__global__ void t_fill(const int64_t *__restrict__ idx, int64_t rows, int k, const int64_t *__restrict__ tptr,
                       int *__restrict__ cursor, int *__restrict__ tlist, int64_t ncols) {
    int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= rows * k) return;
    int64_t c = idx[g];
    if (c >= 0 && c < ncols) {
        int pos = atomicAdd(&cursor[c], 1);
        tlist[tptr[c] + pos] = (int)(g / k);
    }
}
// per query row: total multiset size (sum of transposed-list lengths), and the maximum over rows
__global__ void sn_row_totals(const int64_t *__restrict__ yx, const int64_t *__restrict__ yy, int64_t nq, int k,
                              const int64_t *__restrict__ txx_ptr, const int64_t *__restrict__ txy_ptr,
                              int *__restrict__ tot, int *__restrict__ maxtot) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    int64_t s = 0;
    for (int t = 0; t < k; ++t) {
        int64_t r = yx[i * k + t];
        if (r >= 0) s += txx_ptr[r + 1] - txx_ptr[r];
        int64_t q = yy[i * k + t];
        if (q >= 0) s += txy_ptr[q + 1] - txy_ptr[q];
    }
    tot[i] = (int)(s > 0x7fffffff ? 0x7fffffff : s);
    atomicMax(maxtot, tot[i]);
}

// MODE 0: distinct count per row.  MODE 1: emit (column, transformed count).  transform: 0 raw, 1 jaccard, 2 hnoca.
template <int MODE>
__global__ __launch_bounds__(256) void sn_rows(const int64_t *__restrict__ yx, const int64_t *__restrict__ yy, int64_t nq,
                                                int k, const int64_t *__restrict__ txx_ptr, const int *__restrict__ txx,
                                                const int64_t *__restrict__ txy_ptr, const int *__restrict__ txy,
                                                int *__restrict__ gscratch, int gcap, int32_t *__restrict__ row_nnz,
                                                const int64_t *__restrict__ out_indptr, int32_t *__restrict__ ocols,
                                                double *__restrict__ ovals, int transform) {
    __shared__ int lbuf[SN_CAP];
    __shared__ int s_off[2050];  // prefix of list lengths over the 2k neighbour slots (k <= 1024)
    __shared__ int s_scan[256];
    const int tid = threadIdx.x;
    for (int64_t i = blockIdx.x; i < nq; i += gridDim.x) {
        // 1. lengths of the 2k source lists and their exclusive prefix
        for (int t = tid; t < 2 * k; t += 256) {
            int64_t v = t < k ? yx[i * k + t] : yy[i * k + (t - k)];
            const int64_t *ptr = t < k ? txx_ptr : txy_ptr;
            s_off[t + 1] = v >= 0 ? (int)(ptr[v + 1] - ptr[v]) : 0;
        }
        if (tid == 0) s_off[0] = 0;
        __syncthreads();
        if (tid == 0) for (int t = 0; t < 2 * k; ++t) s_off[t + 1] += s_off[t];
        __syncthreads();
        const int M = s_off[2 * k];
        int P = 1;
        while (P < M) P <<= 1;
        int *buf = (P <= SN_CAP) ? lbuf : gscratch + (size_t)blockIdx.x * gcap;
        // 2. gather the multiset (one wave-sized stride per source list), pad to a power of two
        for (int t = 0; t < 2 * k; ++t) {
            const int len = s_off[t + 1] - s_off[t];
            if (len == 0) continue;
            int64_t v = t < k ? yx[i * k + t] : yy[i * k + (t - k)];
            const int *src = (t < k ? txx + txx_ptr[v] : txy + txy_ptr[v]);
            for (int e = tid; e < len; e += 256) buf[s_off[t] + e] = src[e];
        }
        for (int e = M + tid; e < P; e += 256) buf[e] = 0x7fffffff;
        __syncthreads();
        // 3. bitonic sort, ascending
        for (int size = 2; size <= P; size <<= 1) {
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                for (int e = tid; e < (P >> 1); e += 256) {
                    int lo = 2 * e - (e & (stride - 1));
                    int hi = lo + stride;
                    bool up = ((lo & size) == 0);
                    int a = buf[lo], b = buf[hi];
                    if ((a > b) == up) { buf[lo] = b; buf[hi] = a; }
                }
                __syncthreads();
            }
        }
        // 4. run-length encode: heads are positions whose predecessor differs
        const int per = (M + 255) / 256;
        const int e0 = tid * per, e1 = (e0 + per < M) ? e0 + per : M;
        int mine = 0;
        for (int e = e0; e < e1; ++e) mine += (e == 0 || buf[e] != buf[e - 1]) ? 1 : 0;
        s_scan[tid] = mine;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            int add = tid >= o ? s_scan[tid - o] : 0;
            __syncthreads();
            s_scan[tid] += add;
            __syncthreads();
        }
        if (MODE == 0) {
            if (tid == 255) row_nnz[i] = s_scan[255];
        } else {
            int64_t o = out_indptr[i] + s_scan[tid] - mine;
            for (int e = e0; e < e1; ++e) {
                if (e == 0 || buf[e] != buf[e - 1]) {
                    int v = buf[e], run = 1;
                    while (e + run < M && buf[e + run] == v) ++run;
                    double cnt = (double)run, val = cnt;
                    if (transform == 1) val = cnt / (4.0 * k - cnt);
                    else if (transform == 2) { double r = cnt / (2.0 * k - cnt); val = r * r; }
                    ocols[o] = v;
                    ovals[o] = val;
                    ++o;
                }
            }
        }
        __syncthreads();
    }
}

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    cm_status alloc(cm_handle_t h, size_t bytes) {
        hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
        if (e != hipSuccess) { p = nullptr; CM_FAIL(h, CM_E_NOMEM, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e)); }
        return CM_OK;
    }
    template <typename T> T *as() { return reinterpret_cast<T *>(p); }
};

cm_status build_transpose(cm_handle_t h, const int64_t *d_idx, int64_t rows, int k, int64_t ncols, DevBuf &tptr, DevBuf &tlist) {
    DevBuf deg, cursor;
    CM_TRY(deg.alloc(h, 4 * (size_t)ncols));
    CM_TRY(cursor.alloc(h, 4 * (size_t)ncols));
    CM_TRY(tptr.alloc(h, 8 * ((size_t)ncols + 1)));
    CM_TRY(tlist.alloc(h, 4 * (size_t)rows * k));
    CM_HIP(h, hipMemsetAsync(deg.p, 0, 4 * (size_t)ncols, h->stream));
    CM_HIP(h, hipMemsetAsync(cursor.p, 0, 4 * (size_t)ncols, h->stream));
    int64_t n = rows * k;
    hipLaunchKernelGGL(t_count, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, d_idx, n, deg.as<int>(), ncols);
    CM_TRY(cm_exclusive_scan_i32_to_i64(h, deg.as<int>(), ncols, tptr.as<int64_t>()));
    hipLaunchKernelGGL(t_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, d_idx, rows, k, tptr.as<int64_t>(),
                       cursor.as<int>(), tlist.as<int>(), ncols);
    CM_HIP(h, hipGetLastError());
    CM_HIP(h, hipStreamSynchronize(h->stream));  // deg / cursor are released on return
    return CM_OK;
}

}  // namespace

extern "C" cm_status cm_shared_neighbors(cm_handle_t h, const int64_t *yx, const int64_t *xx, const int64_t *yy,
                                         const int64_t *xy, int64_t nq, int64_t nr, int32_t k, int32_t transform,
                                         int64_t *out_indptr, int32_t *out_cols, double *out_vals, int64_t capacity,
                                         int64_t *out_nnz) {
    if (!h) return CM_E_INVALID;
    h->err.clear();
    if (k < 1 || k > 1024 || nq < 0 || nr < 1) CM_FAIL(h, CM_E_INVALID, "shared neighbours: bad shape (k=%d)", k);
    if (transform < 0 || transform > 2) CM_FAIL(h, CM_E_INVALID, "shared neighbours: unknown transform %d", transform);
    CM_HIP(h, hipSetDevice(h->device));
    cm_ws_reset(h);
    CM_TRY(cm_ws_reserve(h, cm_scan_ws_bytes(std::max(nq, nr)) * 2 + 65536));
    DevBuf dyx, dxx, dyy, dxy, txx_ptr, txx, txy_ptr, txy, tot, rn, optr, oc, ov, gs, mx;
    CM_TRY(dyx.alloc(h, 8 * (size_t)nq * k)); CM_TRY(dyy.alloc(h, 8 * (size_t)nq * k));
    CM_TRY(dxx.alloc(h, 8 * (size_t)nr * k)); CM_TRY(dxy.alloc(h, 8 * (size_t)nr * k));
    CM_HIP(h, hipMemcpyAsync(dyx.p, yx, 8 * (size_t)nq * k, hipMemcpyHostToDevice, h->stream));
    CM_HIP(h, hipMemcpyAsync(dyy.p, yy, 8 * (size_t)nq * k, hipMemcpyHostToDevice, h->stream));
    CM_HIP(h, hipMemcpyAsync(dxx.p, xx, 8 * (size_t)nr * k, hipMemcpyHostToDevice, h->stream));
    CM_HIP(h, hipMemcpyAsync(dxy.p, xy, 8 * (size_t)nr * k, hipMemcpyHostToDevice, h->stream));
    CM_TRY(build_transpose(h, dxx.as<int64_t>(), nr, k, nr, txx_ptr, txx));  // columns: reference cells
    cm_ws_reset(h);
    CM_TRY(build_transpose(h, dxy.as<int64_t>(), nr, k, nq, txy_ptr, txy));  // columns: query cells
    cm_ws_reset(h);
    CM_TRY(tot.alloc(h, 4 * (size_t)nq)); CM_TRY(rn.alloc(h, 4 * (size_t)nq)); CM_TRY(optr.alloc(h, 8 * ((size_t)nq + 1)));
    CM_TRY(mx.alloc(h, 16));
    CM_HIP(h, hipMemsetAsync(mx.p, 0, 16, h->stream));
    if (nq) hipLaunchKernelGGL(sn_row_totals, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, h->stream, dyx.as<int64_t>(),
                               dyy.as<int64_t>(), nq, k, txx_ptr.as<int64_t>(), txy_ptr.as<int64_t>(), tot.as<int>(), mx.as<int>());
    int maxtot = 0;
    CM_HIP(h, hipMemcpyAsync(&maxtot, mx.p, 4, hipMemcpyDeviceToHost, h->stream));
    CM_HIP(h, hipStreamSynchronize(h->stream));
    if (maxtot == 0x7fffffff) CM_FAIL(h, CM_E_UNSUPPORTED, "shared neighbours: a row gathers more than 2^31 entries");
    unsigned grid = (unsigned)(nq < 4096 ? (nq > 0 ? nq : 1) : 4096);
    int gcap = 0;
    if (maxtot > SN_CAP) {  // rows too long for LDS sort in a per-workgroup global scratch
        gcap = 1;
        while (gcap < maxtot) gcap <<= 1;
        CM_TRY(gs.alloc(h, 4 * (size_t)grid * gcap));
    }
    if (nq) hipLaunchKernelGGL(sn_rows<0>, dim3(grid), dim3(256), 0, h->stream, dyx.as<int64_t>(), dyy.as<int64_t>(), nq, k,
                               txx_ptr.as<int64_t>(), txx.as<int>(), txy_ptr.as<int64_t>(), txy.as<int>(), gs.as<int>(), gcap,
                               rn.as<int32_t>(), (const int64_t *)nullptr, (int32_t *)nullptr, (double *)nullptr, transform);
    CM_TRY(cm_exclusive_scan_i32_to_i64(h, rn.as<int32_t>(), nq, optr.as<int64_t>()));
    CM_HIP(h, hipMemcpyAsync(out_indptr, optr.p, 8 * ((size_t)nq + 1), hipMemcpyDeviceToHost, h->stream));
    CM_HIP(h, hipStreamSynchronize(h->stream));
    int64_t nnz = out_indptr[nq];
    if (out_nnz) *out_nnz = nnz;
    if (!out_cols || !out_vals || capacity < nnz) return CM_OK;  // sizing call: the caller allocates nnz entries and calls again
    CM_TRY(oc.alloc(h, 4 * (size_t)nnz)); CM_TRY(ov.alloc(h, 8 * (size_t)nnz));
    if (nq) hipLaunchKernelGGL(sn_rows<1>, dim3(grid), dim3(256), 0, h->stream, dyx.as<int64_t>(), dyy.as<int64_t>(), nq, k,
                               txx_ptr.as<int64_t>(), txx.as<int>(), txy_ptr.as<int64_t>(), txy.as<int>(), gs.as<int>(), gcap,
                               rn.as<int32_t>(), optr.as<int64_t>(), oc.as<int32_t>(), ov.as<double>(), transform);
    CM_HIP(h, hipGetLastError());
    CM_HIP(h, hipMemcpyAsync(out_cols, oc.p, 4 * (size_t)nnz, hipMemcpyDeviceToHost, h->stream));
    CM_HIP(h, hipMemcpyAsync(out_vals, ov.p, 8 * (size_t)nnz, hipMemcpyDeviceToHost, h->stream));
    CM_HIP(h, hipStreamSynchronize(h->stream));
    return CM_OK;
}
