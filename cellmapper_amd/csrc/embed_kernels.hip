// Joint representations (second half): the dense linear algebra behind the reference's
// `use_rep=None` fallbacks -- fast CCA (embedding.py:50-185, utils.py:230-398: truncated SVD of the cross-covariance
// (X - rowmean) (Y - rowmean)^T / (G - 1)) and joint PCA (embedding.py:14-48).  Both reduce to two primitives over a tall
// cell x gene matrix A (dense float32 / float64 or CSR float32), evaluated in float64 on the f64 matrix cores
// (v_mfma_f64_16x16x4_f64):
//     cm_gram    : G = A'^T A'       [p, p]      A' = A, or A with every ROW centred at its mean
//     cm_project : Z = A' P          [n, nc]     P [p, nc] float64;  A' = A, row-centred, or shifted by a column vector
// Everything that is p x p (eigen-decomposition, the small SVD) is the host's business (cellmapper_amd/embedding.py).
//
// Data flow: row tiles of A are staged into a float64 tile D [rows, pp] (pp = p padded to 64; centring applied here, CSR
// rows scattered into a cleared tile), so one Gram / projection kernel serves every input format.
#include "cm_internal.h"

#include <algorithm>
#include <cstdlib>

namespace {
typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int EB_ROWS = 32768;   // rows per staged tile (x pp x 8 B; bounded below for wide matrices)
constexpr int EB_KC = 32;        // rows per LDS chunk of the Gram kernel

// ---- staging: D[r][c] = A[r0 + r][c] - shift,  shift = mean of the row (mode 1) / col_shift[c] (mode 2) / 0
template <typename T>
__global__ __launch_bounds__(256) void stage_dense(const T *__restrict__ A, int64_t n, int p, int pp, int64_t r0, int rows, int mode,
                                                    const double *__restrict__ col_shift, double *__restrict__ D,
                                                    double *__restrict__ row_mean_out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wave;
    if (r >= rows) return;
    const int64_t gr = r0 + r;
    double *drow = D + (size_t)r * pp;
    if (gr >= n) { for (int c = lane; c < pp; c += 64) drow[c] = 0.0; return; }
    const T *arow = A + gr * (int64_t)p;
    double mean = 0.0;
    if (mode == 1) {   // numpy's mean(axis=1) is a pairwise sum; float64 sequential-by-lane + tree is within 1e-15 of it
        double s = 0.0;
        for (int c = lane; c < p; c += 64) s += (double)arow[c];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        mean = s / (double)p;
        if (row_mean_out && lane == 0) row_mean_out[gr] = mean;
    }
    for (int c = lane; c < pp; c += 64) {
        double v = 0.0;
        if (c < p) v = (double)arow[c] - (mode == 1 ? mean : (mode == 2 ? col_shift[c] : 0.0));
        drow[c] = v;
    }
}

__global__ __launch_bounds__(256) void stage_csr(const int64_t *__restrict__ indptr, const int32_t *__restrict__ cols,
                                                  const float *__restrict__ vals, int64_t n, int p, int pp, int64_t r0, int rows,
                                                  int mode, const double *__restrict__ col_shift, double *__restrict__ D) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wave;
    if (r >= rows) return;
    const int64_t gr = r0 + r;
    double *drow = D + (size_t)r * pp;
    if (gr >= n) { for (int c = lane; c < pp; c += 64) drow[c] = 0.0; return; }
    const int64_t b0 = indptr[gr], b1 = indptr[gr + 1];
    double mean = 0.0;
    if (mode == 1) {
        double s = 0.0;
        for (int64_t t = b0 + lane; t < b1; t += 64) s += (double)vals[t];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        mean = s / (double)p;
    }
    for (int c = lane; c < pp; c += 64) drow[c] = c < p ? -(mode == 1 ? mean : (mode == 2 ? col_shift[c] : 0.0)) : 0.0;
    // the stored entries overwrite their cells with (value - shift): plain stores, after the fill has been acknowledged
    // (a different lane wrote the cell above; canonical CSR: distinct columns within a row, so no cell is written twice)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int64_t t = b0 + lane; t < b1; t += 64) {
        const int c = cols[t];
        if (c >= 0 && c < p) drow[c] = (double)vals[t] - (mode == 1 ? mean : (mode == 2 ? col_shift[c] : 0.0));
    }
}

// ---- Gram: partial[split][mi*64 .. , ni*64 ..] += D[rows of the split]^T D[...]   (upper block triangle, mi <= ni)
// Workgroup = 4 waves; output tile 64 x 64; wave w owns the 32 x 32 quadrant (w >> 1, w & 1) = 2 x 2 MFMA tiles.
// v_mfma_f64_16x16x4_f64: A operand lane l = A[m = l & 15][k = l >> 4], B operand lane l = B[k = l >> 4][n = l & 15],
// C lane l, element v = C[m = 4 * v + (l >> 4)][n = l & 15]   (probed on the device; tests/test_gpu_boundary.py pins it).
__global__ __launch_bounds__(256) void gram_tiles(const double *__restrict__ D, int rows, int pp, int nblk, int rows_per_split,
                                                   double *__restrict__ partial) {
    __shared__ double sa[EB_KC][64 + 2], sb[EB_KC][64 + 2];
    // blockIdx.x -> (mi, ni) with mi <= ni
    int t = blockIdx.x, mi = 0;
    while (t >= nblk - mi) { t -= nblk - mi; ++mi; }
    const int ni = mi + t;
    const int split = blockIdx.y;
    const int ra = split * rows_per_split, rb = std::min(rows, ra + rows_per_split);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int qm = (wave >> 1) * 32, qn = (wave & 1) * 32;
    f64x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};
    for (int r0 = ra; r0 < rb; r0 += EB_KC) {
        for (int e = tid; e < EB_KC * 64; e += 256) {
            const int rr = e >> 6, cc = e & 63;
            const bool in = r0 + rr < rb;
            sa[rr][cc] = in ? D[(size_t)(r0 + rr) * pp + mi * 64 + cc] : 0.0;
            sb[rr][cc] = in ? D[(size_t)(r0 + rr) * pp + ni * 64 + cc] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int k0 = 0; k0 < EB_KC; k0 += 4) {
            const int kk = k0 + (lane >> 4);
            const double a0 = sa[kk][qm + (lane & 15)], a1 = sa[kk][qm + 16 + (lane & 15)];
            const double b0 = sb[kk][qn + (lane & 15)], b1 = sb[kk][qn + 16 + (lane & 15)];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }
    double *out = partial + (size_t)split * pp * pp;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int m = mi * 64 + qm + i * 16 + 4 * v + (lane >> 4), nn = ni * 64 + qn + j * 16 + (lane & 15);
                out[(size_t)m * pp + nn] += acc[i][j][v];
            }
}

// out[p, p] = sum over splits of the upper block triangle, mirrored
__global__ void gram_finish(const double *__restrict__ partial, int nsplit, int pp, int p, double *__restrict__ out) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (int64_t)p * p) return;
    const int i = (int)(g / p), j = (int)(g % p);
    const int a = (i / 64 <= j / 64) ? i : j, b = (i / 64 <= j / 64) ? j : i;   // stored block: row block <= column block
    double s = 0.0;
    for (int sidx = 0; sidx < nsplit; ++sidx) s += partial[(size_t)sidx * pp * pp + (size_t)a * pp + b];
    out[g] = s;
}

// ---- projection: Z[rows, nc] = D[rows, pp] P[pp, ncp]   (ncp = nc padded to 16, <= 64).  Workgroup = 4 waves, 64 rows;
// wave w owns rows 16 w .. 16 w + 15 and all ncp / 16 column tiles.
__global__ __launch_bounds__(256) void project_rows(const double *__restrict__ D, int rows, int pp, const double *__restrict__ P,
                                                     int ncp, int nc, double *__restrict__ Z, int64_t z_row0, int64_t n) {
    __shared__ double sd[64][EB_KC + 1], sp[EB_KC][64 + 2];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int rbase = blockIdx.x * 64;
    const int ntile = ncp / 16;
    f64x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f64x4{0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < pp; k0 += EB_KC) {
        for (int e = tid; e < 64 * EB_KC; e += 256) {
            const int rr = e / EB_KC, kk = e % EB_KC;
            sd[rr][kk] = rbase + rr < rows ? D[(size_t)(rbase + rr) * pp + k0 + kk] : 0.0;
        }
        for (int e = tid; e < EB_KC * 64; e += 256) {
            const int kk = e >> 6, cc = e & 63;
            sp[kk][cc] = cc < ncp ? P[(size_t)(k0 + kk) * ncp + cc] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int kq = 0; kq < EB_KC; kq += 4) {
            const int kk = kq + (lane >> 4);
            const double a = sd[wave * 16 + (lane & 15)][kk];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < ntile) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, sp[kk][j * 16 + (lane & 15)], acc[j], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (j < ntile)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int r = rbase + wave * 16 + 4 * v + (lane >> 4), c = j * 16 + (lane & 15);
                if (r < rows && c < nc && z_row0 + r < n) Z[(z_row0 + r) * (int64_t)nc + c] = acc[j][v];
            }
}

struct EDev {   // plain RAII device buffer (these calls are rare and large: no pooling)
    void *p = nullptr;
    ~EDev() { if (p) (void)hipFree(p); }
    cm_status alloc(cm_handle_t h, size_t bytes) {
        hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
        if (e != hipSuccess) { p = nullptr; CM_FAIL(h, CM_E_NOMEM, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e)); }
        return CM_OK;
    }
    template <typename T> T *as() { return reinterpret_cast<T *>(p); }
};

struct Input {   // one tall matrix on the device, dense or CSR
    EDev dense, ptr, col, val;
    bool is_csr = false;
    int dtype = CM_F32;
};

cm_status upload_input(cm_handle_t h, const void *X, const int64_t *indptr, const int32_t *cols, const float *vals, int64_t n, int32_t p,
                       int32_t dtype, Input &in) {
    hipStream_t st = h->stream;
    if (X) {
        const size_t bytes = (size_t)n * p * (dtype == CM_F64 ? 8 : 4);
        CM_TRY(in.dense.alloc(h, bytes));
        CM_HIP(h, hipMemcpyAsync(in.dense.p, X, bytes, hipMemcpyHostToDevice, st));
        in.dtype = dtype;
    } else {
        if (!indptr || !cols || !vals) CM_FAIL(h, CM_E_INVALID, "joint representation: neither a dense matrix nor CSR arrays given");
        const int64_t nnz = indptr[n];
        in.is_csr = true;
        CM_TRY(in.ptr.alloc(h, 8 * ((size_t)n + 1))); CM_TRY(in.col.alloc(h, 4 * (size_t)nnz)); CM_TRY(in.val.alloc(h, 4 * (size_t)nnz));
        CM_HIP(h, hipMemcpyAsync(in.ptr.p, indptr, 8 * ((size_t)n + 1), hipMemcpyHostToDevice, st));
        CM_HIP(h, hipMemcpyAsync(in.col.p, cols, 4 * (size_t)nnz, hipMemcpyHostToDevice, st));
        CM_HIP(h, hipMemcpyAsync(in.val.p, vals, 4 * (size_t)nnz, hipMemcpyHostToDevice, st));
    }
    return CM_OK;
}

cm_status stage_tile(cm_handle_t h, Input &in, int64_t n, int p, int pp, int64_t r0, int rows, int mode, const double *d_shift, double *D) {
    const unsigned grid = (unsigned)((rows + 3) / 4);
    if (in.is_csr)
        hipLaunchKernelGGL(stage_csr, dim3(grid), dim3(256), 0, h->stream, in.ptr.as<int64_t>(), in.col.as<int32_t>(), in.val.as<float>(), n, p, pp,
                           r0, rows, mode, d_shift, D);
    else if (in.dtype == CM_F64)
        hipLaunchKernelGGL(stage_dense<double>, dim3(grid), dim3(256), 0, h->stream, in.dense.as<double>(), n, p, pp, r0, rows, mode, d_shift, D,
                           (double *)nullptr);
    else
        hipLaunchKernelGGL(stage_dense<float>, dim3(grid), dim3(256), 0, h->stream, in.dense.as<float>(), n, p, pp, r0, rows, mode, d_shift, D,
                           (double *)nullptr);
    CM_HIP(h, hipGetLastError());
    return CM_OK;
}

int tile_rows(int pp) {   // rows per staged tile: at most 1 GiB of float64
    int64_t r = ((int64_t)1 << 30) / ((int64_t)pp * 8);
    r = std::max<int64_t>(64, std::min<int64_t>(EB_ROWS, r / 64 * 64));
    return (int)r;
}
}  // namespace

extern "C" {

cm_status cm_gram(cm_handle_t h, const void *X, const int64_t *indptr, const int32_t *cols, const float *vals, int64_t n, int32_t p,
                  int32_t dtype, int32_t center_rows, double *out) {
    if (!h) return CM_E_INVALID;
    h->err.clear();
    if (n < 1 || p < 1 || !out || (X && dtype != CM_F32 && dtype != CM_F64)) CM_FAIL(h, CM_E_INVALID, "gram: bad arguments");
    CM_HIP(h, hipSetDevice(h->device));
    const int pp = (p + 63) / 64 * 64, nblk = pp / 64;
    const int ntile2 = nblk * (nblk + 1) / 2;
    const int T = tile_rows(pp);
    // few output tiles: split the rows of a staged tile over several workgroups (each with its own partial sum: the
    // reduction order is fixed, results are deterministic)
    const int nsplit = std::max(1, std::min(16, 1024 / ntile2));
    Input in;
    EDev D, part, dout;
    CM_TRY(upload_input(h, X, indptr, cols, vals, n, p, dtype, in));
    CM_TRY(D.alloc(h, (size_t)T * pp * 8));
    CM_TRY(part.alloc(h, (size_t)nsplit * pp * pp * 8));
    CM_TRY(dout.alloc(h, (size_t)p * p * 8));
    CM_HIP(h, hipMemsetAsync(part.p, 0, (size_t)nsplit * pp * pp * 8, h->stream));
    for (int64_t r0 = 0; r0 < n; r0 += T) {
        const int rows = (int)std::min<int64_t>(T, n - r0);
        CM_TRY(stage_tile(h, in, n, p, pp, r0, rows, center_rows ? 1 : 0, nullptr, D.as<double>()));
        const int rps = ((rows + nsplit - 1) / nsplit + EB_KC - 1) / EB_KC * EB_KC;
        hipLaunchKernelGGL(gram_tiles, dim3(ntile2, nsplit), dim3(256), 0, h->stream, D.as<double>(), rows, pp, nblk, rps, part.as<double>());
        CM_HIP(h, hipGetLastError());
    }
    hipLaunchKernelGGL(gram_finish, dim3((unsigned)(((int64_t)p * p + 255) / 256)), dim3(256), 0, h->stream, part.as<double>(), nsplit, pp, p,
                       dout.as<double>());
    CM_HIP(h, hipGetLastError());
    CM_HIP(h, hipMemcpyAsync(out, dout.p, (size_t)p * p * 8, hipMemcpyDeviceToHost, h->stream));
    CM_HIP(h, hipStreamSynchronize(h->stream));
    return CM_OK;
}

cm_status cm_project(cm_handle_t h, const void *X, const int64_t *indptr, const int32_t *cols, const float *vals, int64_t n, int32_t p,
                     int32_t dtype, int32_t center_mode, const double *col_shift, const double *P, int32_t nc, double *out) {
    if (!h) return CM_E_INVALID;
    h->err.clear();
    if (n < 1 || p < 1 || nc < 1 || nc > 64 || !P || !out || center_mode < 0 || center_mode > 2 || (center_mode == 2 && !col_shift) ||
        (X && dtype != CM_F32 && dtype != CM_F64))
        CM_FAIL(h, CM_E_INVALID, "project: bad arguments (1 <= nc <= 64)");
    CM_HIP(h, hipSetDevice(h->device));
    const int pp = (p + 63) / 64 * 64, ncp = (nc + 15) / 16 * 16;
    const int T = tile_rows(pp);
    Input in;
    EDev D, dP, dS, dZ;
    CM_TRY(upload_input(h, X, indptr, cols, vals, n, p, dtype, in));
    CM_TRY(D.alloc(h, (size_t)T * pp * 8));
    CM_TRY(dP.alloc(h, (size_t)pp * ncp * 8));
    CM_TRY(dZ.alloc(h, (size_t)n * nc * 8));
    CM_HIP(h, hipMemsetAsync(dP.p, 0, (size_t)pp * ncp * 8, h->stream));
    CM_HIP(h, hipMemcpy2DAsync(dP.p, (size_t)ncp * 8, P, (size_t)nc * 8, (size_t)nc * 8, (size_t)p, hipMemcpyHostToDevice, h->stream));
    if (center_mode == 2) {
        CM_TRY(dS.alloc(h, (size_t)p * 8));
        CM_HIP(h, hipMemcpyAsync(dS.p, col_shift, (size_t)p * 8, hipMemcpyHostToDevice, h->stream));
    }
    for (int64_t r0 = 0; r0 < n; r0 += T) {
        const int rows = (int)std::min<int64_t>(T, n - r0);
        CM_TRY(stage_tile(h, in, n, p, pp, r0, rows, center_mode, dS.as<double>(), D.as<double>()));
        hipLaunchKernelGGL(project_rows, dim3((unsigned)((rows + 63) / 64)), dim3(256), 0, h->stream, D.as<double>(), rows, pp, dP.as<double>(), ncp,
                           nc, dZ.as<double>(), r0, n);
        CM_HIP(h, hipGetLastError());
    }
    CM_HIP(h, hipMemcpyAsync(out, dZ.p, (size_t)n * nc * 8, hipMemcpyDeviceToHost, h->stream));
    CM_HIP(h, hipStreamSynchronize(h->stream));
    return CM_OK;
}

}  // extern "C"
