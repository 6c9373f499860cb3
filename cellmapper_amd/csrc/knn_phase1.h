// Phase 1 of the exact k-NN (K0): fp32 MFMA distance GEMM fused with a per-query top-K' candidate filter.
//
// Replaces the distance computation + heap of sklearn's EuclideanArgKmin that the reference calls at
// knn.py:428-440.  The result of this phase is NOT final: it is a certified superset filter
// (K' = k + slack candidates per query and the threshold tau below which every candidate was kept);
// phase 2 (knn_rerank) recomputes float64 distances with sklearn's formula, and any query whose top-k
// cannot be certified against the fp32 error bound goes through an exact float64 scan.
//
// gfx950 mapping: one workgroup = NW waves, each wave owns 32 queries whose prepared rows live in VGPRs
// as the B operand of v_mfma_f32_32x32x2_f32; a 128-row reference tile is staged in LDS (A operand) and
// shared by all waves.  Prepared rows fold the norms into the contraction:
//     q' = [q, 1, 0..],  y' = [-2y, |y|^2, 0..]   =>   q'.y' = |y|^2 - 2 q.y   (= d^2 - |q|^2)
// so the MFMA accumulator is already the ranking value.  With queries on the accumulator's lane (column)
// index each lane owns ONE query and 16 reference rows per 32x32 tile: the filter is a min-tree, one
// compare against the query's threshold register, and (rarely) an LDS append.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

namespace cmk {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BR = 128;  // reference rows per LDS tile

__host__ __device__ constexpr int lds_row_stride(int KP) { return ((KP / 4) & 1) ? KP : KP + 4; }

__host__ inline size_t phase1_lds_bytes(int KP, int NW, int C) {
    size_t QB = size_t(NW) * 32;
    return size_t(BR) * lds_row_stride(KP) * 4 + QB * 8 + QB * size_t(C) * 8 + size_t(NW) * C * 8;
}

// Keep the Kc smallest (value, index) pairs of one query's candidate buffer, sorted ascending, and refresh
// its threshold.  Executed by one whole wave (wave-uniform arguments).  Rank-by-counting: every pair's rank is
// the number of pairs that order before it; ranks are unique because reference indices are.
__device__ __forceinline__ void wave_prune(uint2 *qbuf, uint2 *tmp, int *cntp, float *thrp, int Kc, int lane) {
    const int n = *cntp;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        float v = INFINITY;
        int id = 0x7fffffff;
        if (i < n) {
            uint2 x = qbuf[i];
            v = __uint_as_float(x.x);
            id = (int)x.y;
        }
        int rk = 0;
        for (int j = 0; j < n; ++j) {
            uint2 x = qbuf[j];  // same address in every lane: LDS broadcast
            float vj = __uint_as_float(x.x);
            int ij = (int)x.y;
            rk += (vj < v || (vj == v && ij < id)) ? 1 : 0;
        }
        if (i < n && rk < Kc) tmp[rk] = make_uint2(__float_as_uint(v), (unsigned)id);
    }
    __builtin_amdgcn_wave_barrier();
    const int m = n < Kc ? n : Kc;
    for (int i = lane; i < m; i += 64) qbuf[i] = tmp[i];
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        *cntp = m;
        if (n >= Kc) *thrp = __uint_as_float(tmp[Kc - 1].x);
    }
    __builtin_amdgcn_wave_barrier();
}

template <int KP, int NW>
__global__ __launch_bounds__(NW * 64, 1) void knn_cand_mfma(const float *__restrict__ Q, const float *__restrict__ R,
                                                              int tiles_per_split, int n_tiles, int Kc, int C,
                                                              int *__restrict__ cand, float *__restrict__ cthr) {
    constexpr int KS = lds_row_stride(KP);
    constexpr int G8 = KP / 8;
    constexpr bool TAIL = (KP % 8) != 0;
    constexpr int QB = NW * 32;
    constexpr int NT = NW * 64;
    constexpr int KP4 = KP / 4;
    constexpr int PIECES = BR * KP4;
    constexpr int PPT = (PIECES + NT - 1) / NT;

    extern __shared__ __align__(16) unsigned char smem[];
    float *refs = reinterpret_cast<float *>(smem);       // [BR][KS]
    int *cnt = reinterpret_cast<int *>(refs + BR * KS);  // [QB]
    float *thrs = reinterpret_cast<float *>(cnt + QB);   // [QB]
    uint2 *buf = reinterpret_cast<uint2 *>(thrs + QB);   // [QB][C]   (value bits, reference row)
    uint2 *tmpbuf = buf + size_t(QB) * C;                // [NW][C]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, hh = lane >> 5;
    const int qloc = wave * 32 + c;
    const int64_t qglob = (int64_t)blockIdx.x * QB + qloc;
    const int split = blockIdx.y, S = gridDim.y;
    const int t0 = split * tiles_per_split;
    const int t1 = (t0 + tiles_per_split < n_tiles) ? t0 + tiles_per_split : n_tiles;

    // query fragment (B operand): lane (c,hh) holds k = 8g+4hh+j for MFMA (g,j); the same k permutation is
    // applied to the reference fragment, so the contraction is unchanged.
    float qf[KP / 2];
    {
        const float *qrow = Q + qglob * KP;
#pragma unroll
        for (int g = 0; g < G8; ++g) {
            float4 v = *reinterpret_cast<const float4 *>(qrow + 8 * g + 4 * hh);
            qf[4 * g + 0] = v.x; qf[4 * g + 1] = v.y; qf[4 * g + 2] = v.z; qf[4 * g + 3] = v.w;
        }
        if (TAIL) {
            float2 v = *reinterpret_cast<const float2 *>(qrow + 8 * G8 + 2 * hh);
            qf[4 * G8 + 0] = v.x; qf[4 * G8 + 1] = v.y;
        }
    }
    if (tid < QB) { cnt[tid] = 0; thrs[tid] = INFINITY; }
    __syncthreads();
    float thr = INFINITY;
    uint2 *mybuf = buf + size_t(qloc) * C;
    uint2 *wtmp = tmpbuf + size_t(wave) * C;

    float4 pre[PPT];
#define CM_LOAD_TILE(t_)                                                                       \
    {                                                                                          \
        const float4 *src_ = reinterpret_cast<const float4 *>(R + (int64_t)(t_) * BR * KP);    \
        _Pragma("unroll") for (int i = 0; i < PPT; ++i) {                                      \
            int p = tid + i * NT;                                                              \
            pre[i] = (PIECES % NT == 0 || p < PIECES) ? src_[p] : make_float4(0.f, 0.f, 0.f, 0.f); \
        }                                                                                      \
    }
    if (t0 < t1) CM_LOAD_TILE(t0);

    for (int t = t0; t < t1; ++t) {
        __syncthreads();  // all waves are done with the previous tile (and with the cnt/thr init)
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            int p = tid + i * NT;
            if (PIECES % NT == 0 || p < PIECES) {
                int row = p / KP4, c4 = p - row * KP4;
                *reinterpret_cast<float4 *>(refs + row * KS + c4 * 4) = pre[i];
            }
        }
        __syncthreads();
        if (t + 1 < t1) CM_LOAD_TILE(t + 1);  // in flight while this tile is consumed

#pragma unroll 1
        for (int st = 0; st < BR / 32; ++st) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float *arow = refs + (st * 32 + c) * KS;
#pragma unroll
            for (int g = 0; g < G8; ++g) {
                float4 a = *reinterpret_cast<const float4 *>(arow + 8 * g + 4 * hh);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, qf[4 * g + 0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, qf[4 * g + 1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, qf[4 * g + 2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, qf[4 * g + 3], acc, 0, 0, 0);
            }
            if (TAIL) {
                float2 a = *reinterpret_cast<const float2 *>(arow + 8 * G8 + 2 * hh);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, qf[4 * G8 + 0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, qf[4 * G8 + 1], acc, 0, 0, 0);
            }
            // acc[r] = q'.y' for query column c and reference row (r&3) + 8*(r>>2) + 4*hh of this 32-row sub-tile
            float m0 = fminf(fminf(acc[0], acc[1]), fminf(acc[2], acc[3]));
            float m1 = fminf(fminf(acc[4], acc[5]), fminf(acc[6], acc[7]));
            float m2 = fminf(fminf(acc[8], acc[9]), fminf(acc[10], acc[11]));
            float m3 = fminf(fminf(acc[12], acc[13]), fminf(acc[14], acc[15]));
            float m = fminf(fminf(m0, m1), fminf(m2, m3));
            if (__ballot(m < thr) != 0ull) {  // wave-uniform: something in this sub-tile beats a threshold
                const int rbase = (t * BR + st * 32) + 4 * hh;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[r];
                    if (v < thr) {
                        int slot = atomicAdd(&cnt[qloc], 1);
                        mybuf[slot] = make_uint2(__float_as_uint(v), (unsigned)(rbase + (r & 3) + 8 * (r >> 2)));
                    }
                }
                __builtin_amdgcn_wave_barrier();
                int n = cnt[qloc];
                unsigned long long need = __ballot(n > C - 32) & 0xffffffffull;
                while (need) {
                    int ql = __ffsll((long long)need) - 1;
                    need &= need - 1;
                    int qq = wave * 32 + ql;
                    wave_prune(buf + size_t(qq) * C, wtmp, &cnt[qq], &thrs[qq], Kc, lane);
                }
                thr = thrs[qloc];
            }
        }
    }

    // final: sort every query's survivors, emit K' candidate rows (-1 padded) and the certified threshold
    for (int ql = 0; ql < 32; ++ql) {
        int qq = wave * 32 + ql;
        wave_prune(buf + size_t(qq) * C, wtmp, &cnt[qq], &thrs[qq], Kc, lane);
        int64_t qg = (int64_t)blockIdx.x * QB + qq;
        int n = cnt[qq];
        int *dst = cand + (qg * S + split) * Kc;
        const uint2 *qb = buf + size_t(qq) * C;
        for (int i = lane; i < Kc; i += 64) dst[i] = (i < n) ? (int)qb[i].y : -1;
        if (lane == 0) cthr[qg * S + split] = thrs[qq];
    }
}

#undef CM_LOAD_TILE

// ---- wide rows (round 3: 260 <= d <= 2047; round 4: no upper limit).  The same filter with the contraction CHUNKED over the feature dimension: neither
// the queries' prepared rows (KP / 2 registers per lane) nor a 128-row reference tile (128 x KP floats of LDS) fit on chip
// beyond KP = 260, so a tile's four 32 x 32 accumulators (64 registers) stay live while KCW = 64-wide chunks of the tile (LDS,
// 34 KB) and of the wave's 32 query rows (registers, reloaded per tile: L2 hits) stream through; the filter runs after the
// last chunk.  The MFMA chain of a (row, query) pair is the same sequential fp32 accumulation over k as in knn_cand_mfma, so
// the certificate's bound ((KP + 16) 2^-24 |q'||y'|) is unchanged.  Prepared rows: KP a multiple of 64, zero padded.  The next
// chunk's reference pieces and query fragments are fetched into registers while the current chunk is multiplied.
constexpr int KCW = 64;                        // chunk width (floats)
constexpr int KSW = KCW + 4;                   // LDS row stride of a chunk (bank spread, as lds_row_stride)
__host__ inline size_t phase1_wide_lds_bytes(int NW, int C) {
    size_t QB = size_t(NW) * 32;
    return size_t(BR) * KSW * 4 + QB * 8 + QB * size_t(C) * 8 + size_t(NW) * C * 8;
}

template <int NW>
__global__ __launch_bounds__(NW * 64, 1) void knn_cand_mfma_wide(const float *__restrict__ Q, const float *__restrict__ R, int KP,
                                                                   int tiles_per_split, int n_tiles, int Kc, int C,
                                                                   int *__restrict__ cand, float *__restrict__ cthr) {
    constexpr int QB = NW * 32;
    constexpr int NT = NW * 64;
    constexpr int PIECES = BR * (KCW / 4);          // float4 pieces of one tile chunk
    constexpr int PPT = (PIECES + NT - 1) / NT;
    const int n_chunks = KP / KCW;

    extern __shared__ __align__(16) unsigned char smem[];
    float *refs = reinterpret_cast<float *>(smem);        // [BR][KSW]: one chunk of the current tile
    int *cnt = reinterpret_cast<int *>(refs + BR * KSW);  // [QB]
    float *thrs = reinterpret_cast<float *>(cnt + QB);    // [QB]
    uint2 *buf = reinterpret_cast<uint2 *>(thrs + QB);    // [QB][C]   (value bits, reference row)
    uint2 *tmpbuf = buf + size_t(QB) * C;                 // [NW][C]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, hh = lane >> 5;
    const int qloc = wave * 32 + c;
    const int64_t qglob = (int64_t)blockIdx.x * QB + qloc;
    const int split = blockIdx.y, S = gridDim.y;
    const int t0 = split * tiles_per_split;
    const int t1 = (t0 + tiles_per_split < n_tiles) ? t0 + tiles_per_split : n_tiles;
    const float *qrow = Q + qglob * (int64_t)KP;

    if (tid < QB) { cnt[tid] = 0; thrs[tid] = INFINITY; }
    __syncthreads();
    float thr = INFINITY;
    uint2 *mybuf = buf + size_t(qloc) * C;
    uint2 *wtmp = tmpbuf + size_t(wave) * C;

    // one chunk ahead in registers: the reference pieces this thread stores to LDS and the wave's query fragments
    float4 pre[PPT], qn[KCW / 8];
#define CM_WIDE_FETCH(t_, ch_)                                                                                  \
    {                                                                                                           \
        const float *src_ = R + ((int64_t)(t_) * BR) * KP + (ch_) * KCW;                                        \
        _Pragma("unroll") for (int i = 0; i < PPT; ++i) {                                                       \
            const int p = tid + i * NT;                                                                         \
            const int row = p / (KCW / 4), c4 = p - row * (KCW / 4);                                            \
            pre[i] = (PIECES % NT == 0 || p < PIECES) ? *reinterpret_cast<const float4 *>(src_ + (int64_t)row * KP + c4 * 4) \
                                                      : make_float4(0.f, 0.f, 0.f, 0.f);                        \
        }                                                                                                       \
        _Pragma("unroll") for (int g = 0; g < KCW / 8; ++g)                                                     \
            qn[g] = *reinterpret_cast<const float4 *>(qrow + (ch_) * KCW + 8 * g + 4 * hh);                     \
    }
    if (t0 < t1) CM_WIDE_FETCH(t0, 0);

    for (int t = t0; t < t1; ++t) {
        f32x16 acc[BR / 32];
#pragma unroll
        for (int st = 0; st < BR / 32; ++st)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[st][r] = 0.f;
        for (int ch = 0; ch < n_chunks; ++ch) {
            __syncthreads();   // every wave is done with the previous chunk
#pragma unroll
            for (int i = 0; i < PPT; ++i) {
                const int p = tid + i * NT;
                if (PIECES % NT == 0 || p < PIECES) {
                    const int row = p / (KCW / 4), c4 = p - row * (KCW / 4);
                    *reinterpret_cast<float4 *>(refs + row * KSW + c4 * 4) = pre[i];
                }
            }
            // this chunk of the wave's query rows (B operand): lane (c, hh) holds k = 8g + 4hh + j of the chunk
            float qf[KCW / 2];
#pragma unroll
            for (int g = 0; g < KCW / 8; ++g) { qf[4 * g + 0] = qn[g].x; qf[4 * g + 1] = qn[g].y; qf[4 * g + 2] = qn[g].z; qf[4 * g + 3] = qn[g].w; }
            __syncthreads();
            // the next chunk (of this tile or the first of the next one) is in flight while this one is multiplied
            if (ch + 1 < n_chunks) { CM_WIDE_FETCH(t, ch + 1); }
            else if (t + 1 < t1) { CM_WIDE_FETCH(t + 1, 0); }
#pragma unroll
            for (int st = 0; st < BR / 32; ++st) {
                const float *arow = refs + (st * 32 + c) * KSW;
#pragma unroll
                for (int g = 0; g < KCW / 8; ++g) {
                    const float4 a = *reinterpret_cast<const float4 *>(arow + 8 * g + 4 * hh);
                    acc[st] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, qf[4 * g + 0], acc[st], 0, 0, 0);
                    acc[st] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, qf[4 * g + 1], acc[st], 0, 0, 0);
                    acc[st] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, qf[4 * g + 2], acc[st], 0, 0, 0);
                    acc[st] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, qf[4 * g + 3], acc[st], 0, 0, 0);
                }
            }
        }
        // filter of the tile's four sub-tiles (as in knn_cand_mfma)
#pragma unroll
        for (int st = 0; st < BR / 32; ++st) {
            const f32x16 &a16 = acc[st];
            float m0 = fminf(fminf(a16[0], a16[1]), fminf(a16[2], a16[3]));
            float m1 = fminf(fminf(a16[4], a16[5]), fminf(a16[6], a16[7]));
            float m2 = fminf(fminf(a16[8], a16[9]), fminf(a16[10], a16[11]));
            float m3 = fminf(fminf(a16[12], a16[13]), fminf(a16[14], a16[15]));
            float m = fminf(fminf(m0, m1), fminf(m2, m3));
            if (__ballot(m < thr) != 0ull) {
                const int rbase = (t * BR + st * 32) + 4 * hh;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = a16[r];
                    if (v < thr) {
                        int slot = atomicAdd(&cnt[qloc], 1);
                        mybuf[slot] = make_uint2(__float_as_uint(v), (unsigned)(rbase + (r & 3) + 8 * (r >> 2)));
                    }
                }
                __builtin_amdgcn_wave_barrier();
                int n = cnt[qloc];
                unsigned long long need = __ballot(n > C - 32) & 0xffffffffull;
                while (need) {
                    int ql = __ffsll((long long)need) - 1;
                    need &= need - 1;
                    int qq = wave * 32 + ql;
                    wave_prune(buf + size_t(qq) * C, wtmp, &cnt[qq], &thrs[qq], Kc, lane);
                }
                thr = thrs[qloc];
            }
        }
    }
    for (int ql = 0; ql < 32; ++ql) {
        int qq = wave * 32 + ql;
        wave_prune(buf + size_t(qq) * C, wtmp, &cnt[qq], &thrs[qq], Kc, lane);
        int64_t qg = (int64_t)blockIdx.x * QB + qq;
        int n = cnt[qq];
        int *dst = cand + (qg * S + split) * Kc;
        const uint2 *qb = buf + size_t(qq) * C;
        for (int i = lane; i < Kc; i += 64) dst[i] = (i < n) ? (int)qb[i].y : -1;
        if (lane == 0) cthr[qg * S + split] = thrs[qq];
    }
#undef CM_WIDE_FETCH
}
}  // namespace cmk
