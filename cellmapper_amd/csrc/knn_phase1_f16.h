// Phase 1 (v2) of the exact k-NN: split-fp16 MFMA distance GEMM fused with a per-query top-K' candidate filter.
//
// Same contract as knn_phase1.h (certified superset filter: K' candidate rows per query + threshold tau, consumed by
// knn_rerank), but the contraction runs on v_mfma_f32_32x32x16_f16 -- 16x the per-instruction MAC count of the
// f32-input MFMA -- with every operand split into two fp16 halves:
//     x*sc = hi + lo + r,   hi = fp16(x*sc),  lo = fp16(x*sc - hi),  |r| <= 2^-22 |x*sc|      (sc = power of two)
//     q.y ~= qh.yh + qh.yl + ql.yh            (three MFMA chains; the dropped ql.yl term is <= 2^-22 |q||y|)
// fp16 x fp16 products are exact in the f32 accumulator, so the filter value differs from the float64 value by a bound
// of the same order as the f32 path's; the float64 re-rank + certificate make the result exact.
//
// gfx950 mapping
//   * workgroup = 8 waves (2 per SIMD), one per CU, persistent over (512-query block, reference split) work items;
//   * wave tile = 32 reference rows x 64 queries: two 32x32 accumulators share one A (reference) fragment; the 64
//     queries' hi/lo fragments (B operand) stay in VGPRs for the whole pass;
//   * |y|^2 rides in three spare feature slots as fp16 digits (exact to 2^-33), -2 and the scale are folded into
//     the rows, so the accumulator is the ranking value |y|^2 - 2 q.y with no epilogue arithmetic;
//   * 128-row reference tiles are copied global -> LDS by LDS-DMA (global_load_lds_dwordx4; no VGPR staging, no
//     ds_write pass) into a ring of 8 hi-only / 4 full tiles, one s_barrier per tile; the global tile image is byte-identical to the LDS
//     image, rows XOR-swizzled by 16-byte granule so the fragment ds_read_b128s are bank-conflict free;
//   * queries sit on the accumulator's lane (column) index: each lane owns 2 queries x 16 rows per tile, the filter is
//     a v_min3 tree + one compare against a threshold register; a passing value is appended with a plain 8-byte store
//     to a LANE-PRIVATE global buffer (no atomics; L2-resident); when a lane's buffer could overflow the wave merges
//     it with the query's kept list (rank-by-counting in LDS), keeps the K' smallest and tightens the threshold.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

namespace cmk2 {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef CM2_BR
#define CM2_BR 128
#endif
#ifndef CM2_NW
#define CM2_NW 8
#endif
#ifndef CM2_BT
#define CM2_BT 2   // tiles per workgroup barrier in knn_cand_f16 (8-deep rings)
#endif
#ifndef CM2_SB8
#define CM2_SB8 0   // K = 128 hi*hi kernel: 1 = single fragment set (see knn_cand_f16); measured 0.8 % slower, so off
#endif
#ifndef CM2_PROD_SB
#define CM2_PROD_SB 0   // production K = 128 body: 1 = single fragment set (32 VGPRs less; measured slower in round 2, and the 12 bytes of scratch left are prologue spills)
#endif
#ifndef CM2_EXP
#define CM2_EXP 0  // timing experiments only (scripts/ablate_skeleton.sh): 1 no DMA pieces, 2 no barrier, 4 no fragment reads, 8 no min trees
#endif
constexpr int BR = CM2_BR;  // reference rows per LDS tile
constexpr int NW = CM2_NW;  // waves per workgroup: one 8-wave workgroup per CU (512 queries share every tile fill;
                            // measured 6-10 % faster than two 4-wave workgroups once threshold refreshes were scheduled)
constexpr int NT = NW * 64;
constexpr int QB = NW * 64;  // queries per workgroup pass
constexpr int CL = 128;      // lane-private append buffer capacity (entries)
// LDS tile buffers of knn_cand_f16 (tile t+1 has landed while tile t is computed, see the kernel): 4 full 32-KB tiles
// at level 3, 8 hi-only 16-KB tiles at levels 1/2
#ifndef CM2_RING
#define CM2_RING 8
#endif
__host__ __device__ constexpr int ring_depth(bool hi_only, int KSTEPS = 4) { return (hi_only && KSTEPS < 8) ? CM2_RING : 4; }

__host__ __device__ constexpr int row_bytes(int KSTEPS) { return KSTEPS * 16 * 2 * 2; }  // hi + lo halves
__host__ __device__ constexpr int tile_bytes(int KSTEPS) { return BR * row_bytes(KSTEPS); }
// |y|^2 rides in three spare feature slots as fp16 "digits": sc^2|y|^2 = a1*2^15 + a2*2^4 + a3*2^-7 (query side holds
// the three constants), so it is summed by the hi*hi MFMA chain itself -- no accumulator initialisation, no extra reads.
constexpr int NORM_SLOTS = 3;
// XOR applied to a row's 16-byte granule index (row stride is a power of two: spread rows over the 16 LDS slots)
__host__ __device__ constexpr int swz_b(int row, int rowb) {   // by row size: the index of the row's 256-byte bank row
    return rowb >= 256 ? (row & 15) : (rowb == 128 ? ((row >> 1) & 7) : ((row >> 2) & 3));
}
__host__ __device__ constexpr int swz(int row, int KSTEPS) { return swz_b(row, row_bytes(KSTEPS)); }
// Filter levels 1 and 2 never touch the lo halves of the reference rows: they run on a hi-only tile image (half the
// LDS-DMA traffic and LDS footprint).  Level 3 uses the full [hi | lo] rows.
__host__ __device__ constexpr int img_row_bytes(int KSTEPS, bool hi_only) { return hi_only ? KSTEPS * 32 : row_bytes(KSTEPS); }
// K = 128 (KSTEPS = 8) rows have no spare slots for the norm digits: the tile image ends with a 1-KiB side block of
// sc^2|y|^2 (float32 per row) that initialises the accumulators
__host__ __device__ constexpr int img_tile_bytes(int KSTEPS, bool hi_only) {
    return BR * img_row_bytes(KSTEPS, hi_only) + (KSTEPS >= 8 ? 1024 : 0);
}
__host__ inline size_t lds_bytes(int KSTEPS, int scratch_entries) {
    return 2 * size_t(tile_bytes(KSTEPS)) + size_t(NW) * scratch_entries * 8 + 64;
}

struct P1Args {
    const _Float16 *Q;     // [nq_pad][2*K] prepared query rows (hi | lo), scaled
    const char *R;         // [n_tiles] tile images: BR swizzled rows (hi | lo of [-2*sc*y, norm digits, 0..])
    int n_tiles, tiles_per_split, S;
    int nqb;               // query blocks of QB rows
    int Kc;                // candidates kept per (query, split)
    int *cand;             // [nq_pad][S][Kc]
    float *cthr;           // [nq_pad][S]   (scaled units)
    uint2 *kept;           // [gridDim.x][NW][64][Kc]    per-resident-wave kept lists
    uint2 *lbuf;           // [gridDim.x][NW][64][2][CL] lane-private append buffers
    int nprod;             // split products evaluated: 3 = hi*hi + hi*lo + lo*hi (error ~2^-21 |q||y|), 2 = drop lo(ref)*hi
                           // (~2^-11), 1 = hi*hi only (~2^-10); the certificate in knn_rerank uses the matching bound
    int dbg;               // timing experiments only (CELLMAP_DBG): 1 = never append, 2 = skip the filter, 4 = counters
    unsigned long long *ctr;  // [16] debug counters (dbg & 4): branches, appends, prunes, prune cycles, wave cycles
    const float *side = nullptr;     // centred mode (knn_cand_f16<.,1,true>): [n_buckets][side_stride] float32 -2 sc^2 c_bucket . y'_row
    const int *blk_bucket = nullptr; // centred mode: bucket of every 512-query block
    long long side_stride = 0; // centred mode: floats per bucket row (padded image rows + 256)
    // bucket pruning (knn_kernels.hip, ivf_*): explicit work list instead of (query block, split) pairs.  Item =
    // (query block, first tile, end tile, candidate-list slot); n_wlist_dev (device) or n_wlist items.  thr0_slot0: the
    // thresholds start at the final threshold of the query's slot-0 list (cthr[q * S]) instead of +inf.
    const int4 *wlist = nullptr;
    const int *n_wlist_dev = nullptr;
    int n_wlist = 0;
    int thr0_slot0 = 0;
    // chained runs (round 5; knn_cand_f16 only): with wruns != nullptr an item is (query block, first run, end run, slot) and
    // wruns[r] = (first tile, end tile): the workgroup scans the item's runs one after the other INTO ONE candidate list --
    // thresholds, logs and kept lists carry over -- instead of one list per run.  A cluster that the locality sort split over
    // six buckets then costs the re-rank one list of K' candidates, not six.
    const int2 *wruns = nullptr;
    // sample-seeded thresholds (knn_run_staged): thr0[q] is an initial threshold per query row (any value is valid: only
    // rows below the running threshold are ever appended, so "every non-candidate has a filter value >= the final
    // threshold" holds whatever it started from).  no_sched: no scheduled threshold refreshes, merges only when a log fills.
    const float *thr0 = nullptr;
    int seed_len = 0;      // > 0: threshold-seeding mode of knn_cand_f16 (SEEDL = seed_len; only cthr[] is written)
    int no_sched = 0;      // 1: as described; 2: one scheduled refresh after the first tile, none afterwards (sample scan)
    int *wctr = nullptr;   // work-list mode: items beyond the first gridDim.x are handed out through this counter (zeroed
                           // by the caller): runs differ in length by orders of magnitude, a static assignment does not balance
    unsigned perm_mul = 1; // knn_cand_f16: logical tile t is physical tile (t * perm_mul) % n_tiles (perm_mul coprime with
                           // n_tiles; 0/1 = identity).  A scattered visiting order keeps the streaming thresholds
                           // representative when the reference rows are sorted (by cluster, batch, ...)
};

__device__ __forceinline__ uint2 ld_fresh(const uint2 *p) {
    // L1-bypassing load: the wave reads back entries it stored earlier (stores are write-through to L2)
    unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT);
    return make_uint2((unsigned)(v & 0xffffffffull), (unsigned)(v >> 32));
}

__device__ __forceinline__ float vmin3(float a, float b, float c) {
    float r;  // plain v_min3_f32: no canonicalising v_max pair like fminf() lowers to; NaNs cannot occur here
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// Merge one query's kept list with its two lane buffers and keep the Kc smallest (value, row) pairs (unordered).
// Wave-uniform arguments; all 64 lanes participate; at most 4 entries per lane live in registers (n <= 256).
// Selection is a 32-step radix select on the order-preserving integer image of the float values (ballot + popcount
// per step) instead of a sort; ties at the K'-th value are broken arbitrarily, which keeps the filter invariant
// (every discarded entry is >= the new threshold).  Returns the new kept count; *thr_out is updated when full.
__device__ __forceinline__ unsigned f2key(float v) {
    unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
    unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}
// load phase: entry i of the concatenation (kept list, lane-0 buffer, lane-1 buffer) goes to lane i%64, slot i/64
template <int E>
__device__ __forceinline__ void prune_load(const uint2 *kept_src, int nk, const uint2 *b0, int n0, const uint2 *b1, int n1,
                                           int lane, unsigned (&key)[E], unsigned (&row)[E]) {
    const int n = nk + n0 + n1;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = e * 64 + lane;
        key[e] = 0xffffffffu;
        row[e] = 0;
        if (i < n) {
            const uint2 *src = i < nk ? kept_src + i : (i < nk + n0 ? b0 + (i - nk) : b1 + (i - nk - n0));
            uint2 x = ld_fresh(src);
            key[e] = f2key(__uint_as_float(x.x));
            row[e] = x.y;
        }
    }
}
// select phase: keep the Kc smallest of the n loaded entries, write them (unordered) and refresh the threshold
template <int E>
__device__ __forceinline__ int prune_select(const unsigned (&key)[E], const unsigned (&row)[E], int n, int Kc,
                                            uint2 *dst_pairs, int *dst_rows, float *thr_out, int lane) {
    bool valid[E];
#pragma unroll
    for (int e = 0; e < E; ++e) valid[e] = e * 64 + lane < n;
    unsigned kth = 0xffffffffu;  // n <= Kc: keep everything
    if (n > Kc) {
        // all keys share their leading bits (the values sit in a narrow band below the old threshold): skip them
        unsigned kmin = 0xffffffffu, kmax = 0u;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            kmin = valid[e] ? (key[e] < kmin ? key[e] : kmin) : kmin;
            kmax = valid[e] ? (key[e] > kmax ? key[e] : kmax) : kmax;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            unsigned a = __shfl_xor(kmin, o), b = __shfl_xor(kmax, o);
            kmin = a < kmin ? a : kmin;
            kmax = b > kmax ? b : kmax;
        }
        const unsigned diff = kmin ^ kmax;
        const int hb = diff ? 31 - __clz(diff) : -1;  // highest bit in which the keys differ
        unsigned prefix = hb >= 31 ? 0u : (kmin & ~((2u << (hb < 0 ? 0 : hb)) - 1u));
        if (hb < 0) prefix = kmin;
        // Bit by bit: fewer than Kc keys below prefix | 1 << bit  =>  the Kc-th key has that bit set.  ONE unsigned compare per
        // register and bit (round 3; was mask / compare / shift / test).  No key is below the starting prefix (the common
        // leading bits), and an unused slot holds 0xffffffff, which is never below a pivot.
        for (int bit = hb; bit >= 0; --bit) {
            const unsigned pivot = prefix | (1u << bit);
            int c = 0;
#pragma unroll
            for (int e = 0; e < E; ++e) c += __popcll(__ballot(key[e] < pivot));
            if (Kc > c) prefix = pivot;
        }
        kth = prefix;
    }
    // compaction: all entries below the K'-th key, then as many equal ones as still fit
    int pos = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        bool less = valid[e] && (n <= Kc || key[e] < kth);
        unsigned long long b = __ballot(less);
        if (less) {
            int p = pos + __popcll(b & ((1ull << lane) - 1ull));
            if (dst_pairs) dst_pairs[p] = make_uint2(__float_as_uint(key2f(key[e])), row[e]);
            if (dst_rows) dst_rows[p] = (int)row[e];
        }
        pos += __popcll(b);
    }
    if (n > Kc) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            bool eq = valid[e] && key[e] == kth;
            unsigned long long b = __ballot(eq);
            int p = pos + __popcll(b & ((1ull << lane) - 1ull));
            if (eq && p < Kc) {
                if (dst_pairs) dst_pairs[p] = make_uint2(__float_as_uint(key2f(key[e])), row[e]);
                if (dst_rows) dst_rows[p] = (int)row[e];
            }
            pos += __popcll(b);
        }
        *thr_out = key2f(kth);
        return Kc;
    }
    return n;
}
template <int E>
__device__ __forceinline__ int wave_merge_prune_e(const uint2 *kept_src, int nk, const uint2 *b0, int n0, const uint2 *b1,
                                                  int n1, int Kc, uint2 *dst_pairs, int *dst_rows, float *thr_out, int lane) {
    unsigned key[E], row[E];
    prune_load<E>(kept_src, nk, b0, n0, b1, n1, lane, key, row);
    return prune_select<E>(key, row, nk + n0 + n1, Kc, dst_pairs, dst_rows, thr_out, lane);
}
__device__ __forceinline__ int wave_merge_prune(uint2 * /*unused*/, const uint2 *kept_src, int nk, const uint2 *b0, int n0,
                                                const uint2 *b1, int n1, int Kc, uint2 *dst_pairs, int *dst_rows,
                                                float *thr_out, int lane) {
    const int n = nk + n0 + n1;
    if (n <= 128) return wave_merge_prune_e<2>(kept_src, nk, b0, n0, b1, n1, Kc, dst_pairs, dst_rows, thr_out, lane);
    if (n <= 256) return wave_merge_prune_e<4>(kept_src, nk, b0, n0, b1, n1, Kc, dst_pairs, dst_rows, thr_out, lane);
    return wave_merge_prune_e<8>(kept_src, nk, b0, n0, b1, n1, Kc, dst_pairs, dst_rows, thr_out, lane);
}

// ---- shared by the MFMA and the low-d kernels (they use the same lane <-> (query, row) mapping and variable names) --
#define CM2_MIN16(A, OUT)                                                                                      \
    {                                                                                                          \
        float m0_ = vmin3((A)[0], (A)[1], (A)[2]), m1_ = vmin3((A)[3], (A)[4], (A)[5]);                        \
        float m2_ = vmin3((A)[6], (A)[7], (A)[8]), m3_ = vmin3((A)[9], (A)[10], (A)[11]);                      \
        float m4_ = vmin3((A)[12], (A)[13], (A)[14]);                                                          \
        OUT = vmin3(vmin3(m0_, m1_, m2_), vmin3(m3_, m4_, (A)[15]), m0_);                                      \
    }
#define CM2_APPEND1(V, RR, RB)                                                                                 \
    if ((V) < tj_) {                                                                                           \
        mybuf_[cj_] = make_uint2(__float_as_uint(V), (unsigned)((RB) + ((RR) & 3) + 8 * ((RR) >> 2)));         \
        ++cj_;                                                                                                 \
    }
#define CM2_APPEND3(A, R0, RB)                                                                                 \
    if (vmin3((A)[R0], (A)[(R0) + 1], (A)[(R0) + 2]) < tj_) {                                                  \
        CM2_APPEND1((A)[R0], R0, RB) CM2_APPEND1((A)[(R0) + 1], (R0) + 1, RB) CM2_APPEND1((A)[(R0) + 2], (R0) + 2, RB) \
    }
// rare path: a value of the previous sub-tile beats a threshold.  Tests are hierarchical (accumulator ->
// group of three registers -> register) because typically ONE value in ONE lane passes.
#define CM2_FILTER(P0, P1, M0, M1, RB)                                                                         \
    {                                                                                                          \
        const bool pass0_ = (M0) < thr[0], pass1_ = (M1) < thr[1];                                             \
        if (__ballot(pass0_ || pass1_) != 0ull) {                                                              \
            unsigned long long te0_ = 0;                                                                       \
            if (a.dbg & 4) { ++c_br; te0_ = __builtin_amdgcn_s_memtime(); }                                    \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                    \
                const f32x16 &acc = j ? (P1) : (P0);                                                           \
                if (j ? pass1_ : pass0_) {                                                                     \
                    uint2 *mybuf_ = wlbuf + (size_t)((j * 32 + c) * 2 + hh) * CL;                              \
                    const float tj_ = thr[j];                                                                  \
                    int cj_ = cnt[j];                                                                          \
                    CM2_APPEND3(acc, 0, RB) CM2_APPEND3(acc, 3, RB) CM2_APPEND3(acc, 6, RB)                    \
                    CM2_APPEND3(acc, 9, RB) CM2_APPEND3(acc, 12, RB) CM2_APPEND1(acc[15], 15, RB)              \
                    cnt[j] = cj_;                                                                              \
                }                                                                                              \
            }                                                                                                  \
            const bool any_full = (cnt[0] > CL - 16) || (cnt[1] > CL - 16);                                    \
            if (__ballot(any_full) != 0ull) {                                                                  \
                unsigned long long tp0_ = (a.dbg & 4) ? __builtin_amdgcn_s_memtime() : 0ull;                   \
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* own appends visible to own L2 reads */     \
                _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                \
                    unsigned long long need = __ballot(cnt[j] > CL - 16);                                      \
                    need = (need | (need >> 32)) & 0xffffffffull;                                              \
                    while (need) {                                                                             \
                        const int ql_ = __ffsll((long long)need) - 1;                                          \
                        need &= need - 1;                                                                      \
                        const int wq = j * 32 + ql_;                                                           \
                        const int n0 = __shfl(cnt[j], ql_), n1 = __shfl(cnt[j], ql_ + 32), nk = __shfl(keptn[j], ql_); \
                        float tq = __shfl(thr[j], ql_);                                                        \
                        uint2 *kq = wkept + (size_t)wq * a.Kc;                                                 \
                        int nn = wave_merge_prune(wsc, kq, nk, wlbuf + (size_t)(wq * 2) * CL, n0,              \
                                                  wlbuf + (size_t)(wq * 2 + 1) * CL, n1, a.Kc, kq, nullptr, &tq, lane); \
                        if (c == ql_) { cnt[j] = 0; keptn[j] = nn; thr[j] = tq; }                              \
                        if (a.dbg & 4) { ++c_pr; c_app += n0 + n1; }                                           \
                    }                                                                                          \
                }                                                                                              \
                if (a.dbg & 4) c_prcyc += __builtin_amdgcn_s_memtime() - tp0_;                                 \
            }                                                                                                  \
            if (a.dbg & 4) c_evcyc += __builtin_amdgcn_s_memtime() - te0_;                                     \
        }                                                                                                      \
    }
// Append-only filter of the MFMA kernels: the logs are merged at ONE place per tile (CM2_REFRESH_T), so this path is a
// handful of instructions and the (large) merge code exists once in the kernel instead of once per unrolled sub-tile.
// A lane appends at most 16 entries per sub-tile and log; CM2_REFRESH_T keeps every log at most CL - 16*NST full.
#define CM2_FILTER_A(P0, P1, M0, M1, RB)                                                                       \
    {                                                                                                          \
        const bool pass0_ = (M0) < thr[0], pass1_ = (M1) < thr[1];                                             \
        if (__ballot(pass0_ || pass1_) != 0ull) {                                                              \
            unsigned long long te0_ = 0;                                                                       \
            if (a.dbg & 4) { ++c_br; te0_ = __builtin_amdgcn_s_memtime(); }                                    \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                    \
                const f32x16 &acc = j ? (P1) : (P0);                                                           \
                if (__ballot(j ? pass1_ : pass0_) != 0ull) {                                                   \
                    /* Typically ONE value in ONE lane passes.  All 16 compares are issued back to back into    \
                       wave masks (SGPR pairs); the search for the non-empty ones is scalar code, so the common \
                       case costs no VALU->SALU->branch round trip per register. */                             \
                    uint2 *mybuf_ = wlbuf + (size_t)((j * 32 + c) * 2 + hh) * CL;                              \
                    const float tj_ = thr[j];                                                                  \
                    int cj_ = cnt[j];                                                                          \
                    unsigned long long bm_[16];                                                                \
                    _Pragma("unroll") for (int e = 0; e < 16; ++e) bm_[e] = __ballot(acc[e] < tj_);            \
                    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                            \
                        if ((bm_[4 * g] | bm_[4 * g + 1] | bm_[4 * g + 2] | bm_[4 * g + 3]) != 0ull) {         \
                            _Pragma("unroll") for (int e = 4 * g; e < 4 * g + 4; ++e) {                        \
                                if (bm_[e] != 0ull) {                                                          \
                                    if (acc[e] < tj_) {                                                        \
                                        mybuf_[cj_] = make_uint2(__float_as_uint(acc[e]),                      \
                                                                 (unsigned)((RB) + (e & 3) + 8 * (e >> 2)));   \
                                        ++cj_;                                                                 \
                                    }                                                                          \
                                }                                                                              \
                            }                                                                                  \
                        }                                                                                      \
                    }                                                                                          \
                    cnt[j] = cj_;                                                                              \
                }                                                                                              \
            }                                                                                                  \
            if (a.dbg & 4) c_evcyc += __builtin_amdgcn_s_memtime() - te0_;                                     \
        }                                                                                                      \
    }
// Event path of knn_cand_f16 (round 2).  Typically ONE value in ONE lane passes.  The partial minima of the min tree
// (MM[0..4]: registers 3g..3g+2 of P0, MM[5..9]: of P1; register 15 stands alone) are still live when the filter runs, so
// the search descends through them: 6 compares find the group, 3 more the register -- no 16-mask SGPR array (which
// spilled), and the append address is a uniform base + a 32-bit lane offset (saddr form: no 64-bit VALU arithmetic).
#define CM2_APPEND_B(V, E, RB, LOFF)                                                                           \
    if ((V) < tj_) {                                                                                           \
        *reinterpret_cast<uint2 *>(reinterpret_cast<char *>(wlbuf) + (size_t)((LOFF) + 8u * (unsigned)cj_)) =  \
            make_uint2(__float_as_uint(V), (unsigned)((RB) + ((E) & 3) + 8 * ((E) >> 2)));                     \
        ++cj_;                                                                                                 \
    }
#define CM2_GROUP_B(ACC, MMG, G, RB, LOFF)                                                                     \
    if (__ballot((MMG) < tj_) != 0ull) {                                                                       \
        CM2_APPEND_B((ACC)[3 * (G)], 3 * (G), RB, LOFF)                                                        \
        CM2_APPEND_B((ACC)[3 * (G) + 1], 3 * (G) + 1, RB, LOFF)                                                \
        CM2_APPEND_B((ACC)[3 * (G) + 2], 3 * (G) + 2, RB, LOFF)                                                \
    }
#define CM2_FILTER_B(P0, P1, MM, RB)                                                                           \
    {                                                                                                          \
        const bool pass0_ = (MM)[14] < thr[0], pass1_ = (MM)[15] < thr[1];                                     \
        if (__ballot(pass0_ || pass1_) != 0ull) {                                                              \
            unsigned long long te0_ = 0;                                                                       \
            if (a.dbg & 4) { ++c_br; te0_ = __builtin_amdgcn_s_memtime(); }                                    \
            if (__ballot(pass0_) != 0ull) {                                                                    \
                const float tj_ = thr[0];                                                                      \
                int cj_ = cnt[0];                                                                              \
                CM2_GROUP_B(P0, (MM)[0], 0, RB, lboff0) CM2_GROUP_B(P0, (MM)[1], 1, RB, lboff0)                \
                CM2_GROUP_B(P0, (MM)[2], 2, RB, lboff0) CM2_GROUP_B(P0, (MM)[3], 3, RB, lboff0)                \
                CM2_GROUP_B(P0, (MM)[4], 4, RB, lboff0)                                                        \
                if (__ballot((P0)[15] < tj_) != 0ull) CM2_APPEND_B((P0)[15], 15, RB, lboff0)                   \
                cnt[0] = cj_;                                                                                  \
            }                                                                                                  \
            if (__ballot(pass1_) != 0ull) {                                                                    \
                const float tj_ = thr[1];                                                                      \
                int cj_ = cnt[1];                                                                              \
                CM2_GROUP_B(P1, (MM)[5], 0, RB, lboff1) CM2_GROUP_B(P1, (MM)[6], 1, RB, lboff1)                \
                CM2_GROUP_B(P1, (MM)[7], 2, RB, lboff1) CM2_GROUP_B(P1, (MM)[8], 3, RB, lboff1)                \
                CM2_GROUP_B(P1, (MM)[9], 4, RB, lboff1)                                                        \
                if (__ballot((P1)[15] < tj_) != 0ull) CM2_APPEND_B((P1)[15], 15, RB, lboff1)                   \
                cnt[1] = cj_;                                                                                  \
            }                                                                                                  \
            if (a.dbg & 4) c_evcyc += __builtin_amdgcn_s_memtime() - te0_;                                     \
        }                                                                                                      \
    }
// Merge the logs of every query whose lanes hold more than LIM_ entries (runtime loop over the accumulators: one copy
// of the merge code).
#define CM2_MERGE_PASS(LIM_)                                                                                   \
    {                                                                                                          \
        unsigned long long tp0_ = (a.dbg & 4) ? __builtin_amdgcn_s_memtime() : 0ull;                           \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* own appends visible to own L2 reads */             \
        _Pragma("unroll 1") for (int j = 0; j < NJ_; ++j) {                                                    \
            int cj_ = j ? cnt[1] : cnt[0], kj_ = j ? keptn[1] : keptn[0];                                      \
            float tj_ = j ? thr[1] : thr[0];                                                                   \
            unsigned long long need = __ballot(cj_ > (LIM_));                                                  \
            need = (need | (need >> 32)) & 0xffffffffull;                                                      \
            while (need) {                                                                                     \
                const int ql_ = __ffsll((long long)need) - 1;                                                  \
                need &= need - 1;                                                                              \
                const int wq = j * 32 + ql_;                                                                   \
                const int n0 = __shfl(cj_, ql_), n1 = __shfl(cj_, ql_ + 32), nk = __shfl(kj_, ql_);            \
                float tq = __shfl(tj_, ql_);                                                                   \
                uint2 *kq = wkept + (size_t)wq * a.Kc;                                                         \
                int nn = wave_merge_prune(nullptr, kq, nk, wlbuf + (size_t)(wq * 2) * CL, n0,                  \
                                          wlbuf + (size_t)(wq * 2 + 1) * CL, n1, a.Kc, kq, nullptr, &tq, lane); \
                if (c == ql_) { cj_ = 0; kj_ = nn; tj_ = tq; }                                                 \
                if (a.dbg & 4) { ++c_pr; c_app += n0 + n1; }                                                   \
            }                                                                                                  \
            if (j) { cnt[1] = cj_; keptn[1] = kj_; thr[1] = tj_; }                                             \
            else { cnt[0] = cj_; keptn[0] = kj_; thr[0] = tj_; }                                               \
        }                                                                                                      \
        if (a.dbg & 4) c_prcyc += __builtin_amdgcn_s_memtime() - tp0_;                                         \
    }
// End of tile T.  Scheduled threshold refresh: all waves of the workgroup merge at the same tile, so nobody waits at
// the next barrier for a straggler; with refreshes after 2, 4, 8, ... tiles a query appends ~K' entries per doubling
// of scanned rows.  In between, a log that could overflow during the next tile (more than FULL_ entries) is merged
// on the spot (rare).
#define CM2_REFRESH_T(T, T0, FULL_)                                                                            \
    if (!(a.dbg & 3)) {                                                                                        \
        const bool sched_ = (T) == next_refresh;                                                               \
        if (sched_ || __ballot(cnt[0] > (FULL_) || cnt[1] > (FULL_)) != 0ull) {                                \
            if (sched_) next_refresh = (T0) + 2 * ((T) - (T0) + 1) - 1;                                        \
            const int lim_ = sched_ ? sched_lim : (FULL_);                                                     \
            CM2_MERGE_PASS(lim_)                                                                               \
        }                                                                                                      \
    }
// low-d kernel: emergency merges happen inside CM2_FILTER (a 512-row tile can append 256 entries per lane)
#define CM2_REFRESH(T, T0) CM2_REFRESH_T(T, T0, CL)
// Final merge of a work item: emit K' candidate rows (-1 padded) and the threshold of every query of the wave.
#define CM2_FINAL()                                                                                            \
    {                                                                                                          \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                       \
        _Pragma("unroll") for (int j = 0; j < NJ_; ++j) {                                                      \
            for (int ql_ = 0; ql_ < 32; ++ql_) {                                                               \
                const int wq = j * 32 + ql_;                                                                   \
                const int n0 = __shfl(cnt[j], ql_), n1 = __shfl(cnt[j], ql_ + 32), nk = __shfl(keptn[j], ql_); \
                float tq = __shfl(thr[j], ql_);                                                                \
                const int64_t qg = qbase + wq;                                                                 \
                int *dst = a.cand + (qg * a.S + split) * a.Kc;                                                 \
                int nn = wave_merge_prune(nullptr, wkept + (size_t)wq * a.Kc, nk, wlbuf + (size_t)(wq * 2) * CL, n0, \
                                          wlbuf + (size_t)(wq * 2 + 1) * CL, n1, a.Kc, nullptr, dst, &tq, lane); \
                for (int i = nn + lane; i < a.Kc; i += 64) dst[i] = -1;                                        \
                if (lane == 0) a.cthr[qg * a.S + split] = tq;                                                  \
            }                                                                                                  \
        }                                                                                                      \
    }

#define CM2_AS1 __attribute__((address_space(1)))
#define CM2_AS3 __attribute__((address_space(3)))
// 16-byte LDS read the compiler does not see as an LDS access (see knn_cand_f16), and the matching explicit wait: the
// fragments are passed through the wait as in/out operands so that no use can be scheduled above it.
#define CM2_DSR128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off) : "memory")
// Same, reading from ((a ^ x) + tb) + off (x uniform in an SGPR, tb in a VGPR: one constant-bus operand).  The swizzled granule addresses of one row differ by an XOR, so a lane keeps
// ONE row address and forms each fragment address inside the asm (v_xad_u32), where the compiler cannot hoist it into a
// long-lived register per fragment.
#define CM2_DSR128X(dst, a, x, tb, off)                                                                        \
    {                                                                                                          \
        unsigned tmp_;                                                                                         \
        asm volatile("v_xad_u32 %1, %2, %3, %4\n\tds_read_b128 %0, %1 offset:%5"                              \
                     : "=&v"(dst), "=&v"(tmp_)                                                                 \
                     : "v"(a), "s"(x), "v"(tb), "i"(off)                                                       \
                     : "memory");                                                                              \
    }
// Wait until at most CNT LDS reads issued AFTER the fragments f are still outstanding (LDS returns in order; other
// lgkm traffic outstanding at the same time only makes the wait more conservative).
template <int CNT, int N>
__device__ __forceinline__ void wait_lds(half8 (&f)[N]) {
    static_assert(N == 2 || N == 4 || N == 8, "fragment count");
    static_assert(CNT >= 0 && CNT <= 15, "lgkmcnt is a 4-bit counter");
    if constexpr (N == 2) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f[0]), "+v"(f[1]) : "i"(CNT));
    if constexpr (N == 4)
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : "i"(CNT));
    if constexpr (N == 8)
        asm volatile("s_waitcnt lgkmcnt(%8)"
                     : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7])
                     : "i"(CNT));
}

// Step I (0..15) of the two 16-value min trees of CM2_MIN16 (P0 -> m[14], P1 -> m[15]), so that the tree can be dealt
// out over the gaps of an MFMA chain.  I is a compile-time constant after unrolling.
__device__ __forceinline__ void min_step(int I, const f32x16 &P0, const f32x16 &P1, float (&m)[16]) {
    if (I < 5) m[I] = vmin3(P0[3 * I], P0[3 * I + 1], P0[3 * I + 2]);
    else if (I < 10) m[I] = vmin3(P1[3 * (I - 5)], P1[3 * (I - 5) + 1], P1[3 * (I - 5) + 2]);
    else if (I == 10) m[10] = vmin3(m[0], m[1], m[2]);
    else if (I == 11) m[11] = vmin3(m[3], m[4], P0[15]);
    else if (I == 12) m[12] = vmin3(m[5], m[6], m[7]);
    else if (I == 13) m[13] = vmin3(m[8], m[9], P1[15]);
    else if (I == 14) m[14] = vmin3(m[10], m[11], m[0]);
    else m[15] = vmin3(m[12], m[13], m[5]);
}

// NP = number of split products evaluated: 3 = hi*hi + hi*lo(query) + lo(ref)*hi (needs the lo halves of the reference
// rows), 2 = hi*hi + hi*lo(query), 1 = hi*hi.  For NP < 3 the lo fragments' registers double-buffer the hi fragments.
// CEN (hi*hi level, K <= 64): bucket-centred queries.  The B operand holds q'' = q' - c (c = centroid of the bucket all 512
// queries of the work item belong to); the missing term -2 sc^2 c.y' is a float32 side value per (bucket, reference row),
// computed in float64 beforehand (knn_kernels.hip: cen_side) and copied per tile next to the rows like the |y|^2 side
// block of K = 128: it initialises the accumulators.  Every error term of the filter then scales with the radius of the
// bucket instead of the distance of the query from the global mean.
// SEEDL > 0: threshold-seeding mode (hi*hi level over a 1/R row sample; no candidate lists at all).  Every lane keeps the
// SEEDL smallest SUB-TILE MINIMA of each of its two queries in registers (sorted insertion of the min tree's result: 2
// VALU per list entry and sub-tile, no branch, no log, no merge).  A sub-tile minimum is the value of one row and
// different sub-tiles hold different rows, so at least SEEDL rows of the lane's half of the sample lie at or below its
// SEEDL-th entry; the two lanes of a query own disjoint halves, hence at least 2*SEEDL sample rows lie at or below the
// larger of their SEEDL-th entries -- that value is written to cthr[] as the query's seed (see knn_run_staged).
// PROD = true (round 3, instantiated for the K = 128 hi*hi level = config 4's kernel): the production body.
//   * the CELLMAP_DBG modes are compiled out (a.dbg is a constant 0 in a local copy of the arguments: every runtime test of
//     it folds away -- two vector and a dozen scalar instructions per accumulator and sub-tile, and the counters' SGPRs);
//   * the 64 registers of query fragments live in the ACCUMULATION half of the register file and feed the matrix core from
//     there (B operand of `v_mfma`; loaded straight into AGPRs, the MFMAs written as inline asm -- left to the compiler the
//     fragments are parked there and copied back in front of every MFMA): round 2's body needed 256 VGPRs + 44 bytes of
//     scratch, this one fits the 256-register budget of two waves per SIMD without scratch.
template <int KSTEPS, int NP, bool CEN, int SEEDL, bool PROD>
__device__ __forceinline__ void knn_cand_f16_body(P1Args a_in) {
    P1Args a = a_in;
    if constexpr (PROD) a.dbg = 0;
    constexpr bool AQ = false && PROD && NP == 1;   // query fragments in AGPRs (off: see below)
    constexpr bool LO3 = NP == 3;
    constexpr int K = KSTEPS * 16;
    constexpr int ROWB = img_row_bytes(KSTEPS, !LO3);
    constexpr int GRAN = row_bytes(KSTEPS) / 16;   // granules of a full row; hi = [0, GRAN/2), lo = [GRAN/2, GRAN)
    constexpr bool SIDE = KSTEPS >= 8 || CEN;   // a float32 per row initialises the accumulators (|y|^2, or the centring term)
    static_assert(!(SIDE && LO3), "K = 128 runs level 3 on knn_cand_f16w; centring is implemented for the hi*hi level");
    static_assert(!(CEN && (KSTEPS >= 8 || NP != 1)), "centred mode: K <= 64, hi*hi level");
    constexpr int GTILEB = img_tile_bytes(KSTEPS, !LO3);      // stride of the tile images in global memory
    constexpr int TILEB = GTILEB + (CEN ? 1024 : 0);          // stride of the LDS ring (CEN: + the side block)
    constexpr int ROWS_B = BR * ROWB;      // bytes of the rows of a tile; SIDE: followed by BR floats sc^2|y|^2
    constexpr int RING = ring_depth(!LO3, KSTEPS);
    // Workgroup barrier every BT tiles (8-deep ring: BT = CM2_BT, default 2): between barriers the waves drift freely, so
    // the time a wave loses to append events / merges is averaged over BT tiles before the slowest wave is waited for.
    // The barrier at tile t (t - t0 a multiple of BT) makes tiles t+1 .. t+BT complete for everyone and tells that
    // everyone is done with tiles < t; the buffer of tile t - BT is therefore free during tile t.
    constexpr int BT = RING >= 8 ? CM2_BT : 1;
    constexpr int AHEAD = RING - BT;   // the DMA pieces of tile t + AHEAD are issued during tile t
    static_assert(AHEAD - 1 - BT >= 0, "ring too shallow for this barrier interval");
    constexpr int NCHUNK = ROWS_B / 1024;  // 1-KiB pieces of the rows = one LDS-DMA wave-instruction each (+1 side piece)
    constexpr int NST = BR / 32;           // 32-row sub-tiles per tile
    constexpr int NJ_ = 2;                 // 32-query accumulators per wave

    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char *ring = smem;                                            // [RING][TILEB]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, hh = lane >> 5;
    uint2 *wsc = nullptr;
    (void)wsc;
    uint2 *wkept = a.kept + (size_t(blockIdx.x) * NW + wave) * 64 * a.Kc;
    uint2 *wlbuf = a.lbuf + (size_t(blockIdx.x) * NW + wave) * 64 * 2 * CL;
    const unsigned ldsbase = (unsigned)(uintptr_t)(CM2_AS3 unsigned char *)ring;

    // Asynchronous tile staging: global -> LDS by LDS-DMA (global_load_lds_dwordx4), no VGPRs, no ds_write pass.
    // The tile image in global memory is byte-for-byte the LDS image (rows pre-swizzled by prep_rows_f16), so every
    // wave-instruction copies one contiguous KiB.  Wave w moves pieces w, w+NW, ...
#define CM2_DMA_TILE(t_, buf_)                                                                                  \
    {                                                                                                           \
        const char *gsrc_ = a.R + (size_t)(t_) * GTILEB;              /* uniform: scalar base + one 32-bit VGPR */ \
        unsigned char *ldst_ = ring + (buf_) * TILEB + wave * 1024;                                             \
        _Pragma("unroll") for (int q_ = 0; q_ < NCHUNK / NW; ++q_)                                              \
            __builtin_amdgcn_global_load_lds((const CM2_AS1 void *)(gsrc_ + q_ * (NW * 1024) + dma_voff),       \
                                             (CM2_AS3 void *)(ldst_ + q_ * (NW * 1024)), 16, 0, 0);             \
        if (SIDE && wave0) CM2_DMA_SIDE(t_, buf_);                                                              \
    }
    // the side block (1 KiB) is one more piece, copied by wave 0
#define CM2_DMA_SIDE(t_, buf_)                                                                                  \
    __builtin_amdgcn_global_load_lds(                                                                           \
        (const CM2_AS1 void *)(CEN ? reinterpret_cast<const char *>(side_row + (size_t)(t_) * BR) + lane * 16   \
                                   : a.R + (size_t)(t_) * GTILEB + ROWS_B + lane * 16),                         \
        (CM2_AS3 void *)(ring + (buf_) * TILEB + ROWS_B), 16, 0, 0)
#define CM2_DMA_PIECE(t_, buf_, q_)                                                                             \
    __builtin_amdgcn_global_load_lds(                                                                           \
        (const CM2_AS1 void *)(a.R + (size_t)(t_) * GTILEB + (q_) * (NW * 1024) + dma_voff),                    \
        (CM2_AS3 void *)(ring + (buf_) * TILEB + wave * 1024 + (q_) * (NW * 1024)), 16, 0, 0)
    static_assert(NCHUNK % NW == 0, "every wave copies the same number of 1-KiB pieces");
    constexpr int NPIECE = NCHUNK / NW;    // DMA pieces per wave and tile
    static_assert(NPIECE <= BR / 32, "one DMA piece per sub-tile");
    const unsigned dma_voff = wave * 1024 + lane * 16;
    const bool wave0 = __builtin_amdgcn_readfirstlane(wave) == 0;   // scalar: issues (and counts) the side piece
    // physical tile of logical tile t_ (uniform; a 64-bit modulo: used at the start of a work item only, the tile loop
    // advances the physical index by perm_mul modulo n_tiles)
#define CM2_PT(t_) (a.perm_mul > 1u ? (int)(((unsigned long long)(unsigned)(t_) * a.perm_mul) % (unsigned)a.n_tiles) : (int)(t_))

    unsigned long long c_br = 0, c_app = 0, c_pr = 0, c_prcyc = 0, c_evcyc = 0, c_barcyc = 0, c_dmacyc = 0, c_clus = 0;
    unsigned long long c_procyc = 0, c_loopcyc = 0, c_fincyc = 0;
    const unsigned long long c_t0 = __builtin_amdgcn_s_memtime();
    const int n_work = a.wlist ? (a.n_wlist_dev ? *a.n_wlist_dev : a.n_wlist) : a.nqb * a.S;
    auto next_work = [&](int w) -> int {
        if (!a.wctr) return w + (int)gridDim.x;
        volatile int *slot = reinterpret_cast<volatile int *>(smem + RING * TILEB);   // spare bytes behind the ring
        __syncthreads();
        if (tid == 0) *slot = (int)gridDim.x + atomicAdd(a.wctr, 1);
        __syncthreads();
        return __builtin_amdgcn_readfirstlane(*slot);
    };
    for (int work = blockIdx.x; work < n_work; work = next_work(work)) {
        int qb, split, t0, t1;
        int run_b = 0, run_e = 1;   // chained runs of this item (one run unless a.wruns)
        if (a.wlist) {
            const int4 w_ = a.wlist[work];
            qb = __builtin_amdgcn_readfirstlane(w_.x); t0 = __builtin_amdgcn_readfirstlane(w_.y);
            t1 = __builtin_amdgcn_readfirstlane(w_.z); split = __builtin_amdgcn_readfirstlane(w_.w);
            if (a.wruns) { run_b = t0; run_e = t1; }
        } else {
            qb = work / a.S; split = work - qb * a.S;
            t0 = split * a.tiles_per_split;
            t1 = (t0 + a.tiles_per_split < a.n_tiles) ? t0 + a.tiles_per_split : a.n_tiles;
        }
        const int64_t qbase = (int64_t)qb * QB + wave * 64;
        const unsigned long long tw0_ = (a.dbg & 4) ? __builtin_amdgcn_s_memtime() : 0ull;
        // CEN: side values of this work item's bucket, one float per row of the (padded) reference image
        const float *side_row = CEN ? a.side + (size_t)a.blk_bucket[qb] * (size_t)a.side_stride : nullptr;
        (void)side_row;

        // ---- query fragments (B operand): lane (c,hh), k-step s -> granule 2s+hh of the hi / lo half
        constexpr int KL = NP >= 2 ? KSTEPS : 1;   // the lo fragments of the queries are only needed from two products on
        half8 qh[2][KSTEPS], ql[2][KL];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const _Float16 *qrow = a.Q + (qbase + j * 32 + c) * (2 * K);
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) {
                if constexpr (AQ) asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(qh[j][s]) : "v"(qrow + (2 * s + hh) * 8) : "memory");
                else qh[j][s] = *reinterpret_cast<const half8 *>(qrow + (2 * s + hh) * 8);
            }
#pragma unroll
            for (int s = 0; s < KL; ++s) ql[j][s] = *reinterpret_cast<const half8 *>(qrow + K + (2 * s + hh) * 8);
        }
        if constexpr (AQ) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // Touch the query fragments here: the compiler places its vmcnt wait for the loads above at this first use
        // instead of in front of the first MFMA of the tile loop, where it would also wait for the tile prefetch.
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            if constexpr (AQ) asm volatile("" ::"a"(qh[0][s]), "a"(qh[1][s]));
            else asm volatile("" ::"v"(qh[0][s]), "v"(qh[1][s]));
        }
#pragma unroll
        for (int s = 0; s < KL; ++s) asm volatile("" ::"v"(ql[0][s]), "v"(ql[1][s]));
        const float thr_init = (a.dbg & 1) ? -INFINITY : INFINITY;
        float thr[2] = {thr_init, thr_init};
        if (a.thr0_slot0) {
            thr[0] = a.cthr[(qbase + c) * a.S];
            thr[1] = a.cthr[(qbase + 32 + c) * a.S];
            asm volatile("" : "+v"(thr[0]), "+v"(thr[1]));   // (the wait for these loads belongs here, not in the tile loop)
        } else if (a.thr0) {
            thr[0] = a.thr0[qbase + c];
            thr[1] = a.thr0[qbase + 32 + c];
            asm volatile("" : "+v"(thr[0]), "+v"(thr[1]));
        }
        int cnt[2] = {0, 0};     // entries in this lane's append buffer for query j*32+c
        int keptn[2] = {0, 0};   // entries in the query's kept list (same value in both hh lanes)
        float sl0[SEEDL > 0 ? SEEDL : 1], sl1[SEEDL > 0 ? SEEDL : 1];   // SEEDL: smallest sub-tile minima, ascending
#pragma unroll
        for (int i = 0; i < (SEEDL > 0 ? SEEDL : 1); ++i) { sl0[i] = INFINITY; sl1[i] = INFINITY; }
        (void)sl0; (void)sl1;
        // byte offsets of this lane's two append logs from the wave's (uniform) log base
        const unsigned lboff0 = (unsigned)((c * 2 + hh) * CL * 8), lboff1 = (unsigned)(((32 + c) * 2 + hh) * CL * 8);

        // Ring protocol (tile t lives in buffer t % RING).  At the top of tile t every wave waits for its own DMA pieces
        // of tile t+1, then the workgroup barrier makes tile t+1 complete for everyone and tells that everyone is done
        // with tile t-1.  Tile t itself was complete one barrier earlier, so the fragments of its first sub-tile are
        // requested BEFORE the barrier (end of tile t-1): the MFMA chain runs through the barrier without an
        // LDS-latency bubble.  The DMA pieces of tile t+AHEAD (buffer of tile t-1) are spread over the sub-tiles of
        // tile t, so the wait at a tile boundary is a counted vmcnt that leaves the pieces of the AHEAD-2 youngest
        // tiles in flight.
        unsigned long long tw1_ = 0ull;
        for (int run = run_b; run < run_e; ++run) {   // (chained runs: the candidate state above carries from run to run)
        if (a.wruns) {
            const int2 r_ = a.wruns[run];
            t0 = __builtin_amdgcn_readfirstlane(r_.x); t1 = __builtin_amdgcn_readfirstlane(r_.y);
        }
        __builtin_amdgcn_s_barrier();  // previous work item / run: every wave is done reading the ring
#pragma unroll
        for (int u = 0; u < AHEAD; ++u)
            if (t0 + u < t1) CM2_DMA_TILE(CM2_PT(t0 + u), (t0 + u) % RING);

        // Software pipeline: the filter of sub-tile i-1 (VALU) is issued in the shadow of sub-tile i's MFMA chain.
        f32x16 p0, p1;          // accumulators of the previous sub-tile, waiting to be filtered (+inf = nothing pending)
#pragma unroll
        for (int e = 0; e < 16; ++e) { p0[e] = INFINITY; p1[e] = INFINITY; }
        int prbase = 0;         // first reference row (+4*hh) of the previous sub-tile

        // A-operand fragments of one 32-row sub-tile: conflict-free swizzled ds_read_b128, written as inline asm.
        // The compiler cannot tell the in-flight LDS-DMA of the NEXT tile from reads of the CURRENT one and puts
        // s_waitcnt vmcnt(0) in front of the first MFMA if it sees the reads, which serialises the prefetch; the asm
        // reads are invisible to that analysis and are waited for explicitly (CM2_WAIT_LDS) before their first use.
#define CM2_LOAD_AH(F, TB, ST)                                                                                 \
    {                                                                                                          \
        _Pragma("unroll") for (int s = 0; s < KSTEPS; ++s)                                                     \
            CM2_DSR128X(F[s], arow, 32 * s, TB, (ST) * 32 * ROWB);                                             \
    }
        // granule (2s+hh) of the hi half sits at (2s+hh)^f = (hh^f) ^ 2s, the lo half at ... ^ GRAN/2 (f < GRAN, the
        // swizzle repeats every 32 rows): one row address per lane, fragment = address ^ constant
        const unsigned arow = c * ROWB + ((hh ^ swz_b(c, ROWB)) * 16);

        // SIDE: sc^2|y|^2 of the lane's 16 rows of a sub-tile (rows 4*hh + {0..3} + 8*i): half-wave broadcast reads
        f32x4 y0 = {0.f, 0.f, 0.f, 0.f}, y1 = y0, y2 = y0, y3 = y0;
        const unsigned yrow = ROWS_B + 16 * hh;
#define CM2_LOAD_SIDE(TB, ST)                                                                                  \
    {                                                                                                          \
        CM2_DSR128X(y0, yrow, 0, TB, (ST) * 128);                                                              \
        CM2_DSR128X(y1, yrow, 0, TB, (ST) * 128 + 32);                                                         \
        CM2_DSR128X(y2, yrow, 0, TB, (ST) * 128 + 64);                                                         \
        CM2_DSR128X(y3, yrow, 0, TB, (ST) * 128 + 96);                                                         \
    }
        const int sched_lim = a.no_sched ? 8 : 0;   // logs shorter than this are left alone by a scheduled refresh
        int next_refresh = a.no_sched ? (a.no_sched == 2 ? t0 : 0x7fffffff) : t0 + 1;  // tile index at whose end every query's threshold is refreshed (doubling schedule)
        // K = 128 (16-MFMA chains), CM2_SB8: ONE fragment set -- fragment s is last read by MFMA 2s+1 and reloaded right
        // behind it for the next sub-tile, 14 MFMAs (~450 cycles) before its next use; saves the 32 VGPRs of the second set
        // (scratch 44 -> 12 bytes per lane).  Measured at 1.5 M x 1.5 M x d=128 on one box, alternating: 520.3 ms with two
        // sets, 524.0 ms with one -- the spilled values are not touched in the tile loop, so the default keeps two sets.
        constexpr bool SB = !LO3 && KSTEPS >= 8 && (CM2_SB8 || (PROD && CM2_PROD_SB));
        half8 ah[KSTEPS], al[SB ? 1 : KSTEPS];   // LO3: hi / lo fragments; otherwise the two hi fragment buffers
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) ah[s] = half8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < (SB ? 1 : KSTEPS); ++s) al[s] = half8{0, 0, 0, 0, 0, 0, 0, 0};
        if (t0 < t1) {   // first tile: nothing to overlap with
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // both prologue tiles (simplest; once per work item)
            __builtin_amdgcn_s_barrier();
            const unsigned tile0 = ldsbase + (t0 % RING) * TILEB;
            CM2_LOAD_AH(ah, tile0, 0);
            if (SIDE) CM2_LOAD_SIDE(tile0, 0);
        }
        int ptile = CM2_PT(t0), ptile_a = CM2_PT(t0 + AHEAD);
        const int pstep = a.perm_mul > 1u ? (int)(a.perm_mul % (unsigned)a.n_tiles) : 1;
        if (run == run_b) tw1_ = (a.dbg & 4) ? __builtin_amdgcn_s_memtime() : 0ull;
        for (int t = t0; t < t1; ++t) {
            unsigned long long tb0_ = (a.dbg & 4) ? __builtin_amdgcn_s_memtime() : 0ull;
            if (t > t0 && (BT == 1 || (t - t0) % BT == 0)) {   // (the first tile synchronised above)
                // own pieces of tiles t+1 .. t+BT have landed (in flight before this point: tiles t+1 .. t+AHEAD-1; those
                // of the AHEAD-1-BT youngest may stay in flight -- loads complete in order, stores in between only make
                // the wait more conservative)
                if (t + AHEAD - 1 >= t1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (conservative in the tail of a work item)
                else if (SIDE && wave0) asm volatile("s_waitcnt vmcnt(%0)" ::"i"((AHEAD - 1 - BT) * (NPIECE + 1)) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"i"((AHEAD - 1 - BT) * NPIECE) : "memory");
                if (a.dbg & 4) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); c_dmacyc += t_ - tb0_; tb0_ = t_; }
                if (!(CM2_EXP & 2)) __builtin_amdgcn_s_barrier();
                if (a.dbg & 4) c_barcyc += __builtin_amdgcn_s_memtime() - tb0_;
            }
            const unsigned tile = ldsbase + (t % RING) * TILEB;
            const unsigned tile_n = ldsbase + ((t + 1) % RING) * TILEB;
            // physical tiles of t and of t + AHEAD (advanced by perm_mul modulo n_tiles at the end of the iteration)
            const bool more = t + 1 < t1;   // uniform: prefetch the next tile's first fragments at the last sub-tile
#pragma unroll
            for (int st = 0; st < NST; ++st) {
                // accumulator registers map to rows (r&3) + 8*(r>>2) + 4*hh of the sub-tile, one query per lane column
                f32x16 acc0, acc1;
#pragma unroll
                for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
                // fragments of the next sub-tile: st+1 of this tile, or sub-tile 0 of the next one
                const unsigned ntile = (st + 1 < NST) ? tile : tile_n;
                const int nst = (st + 1 < NST) ? st + 1 : 0;
                const bool pre = (st + 1 < NST) || more;
                float mm[16];
                if (CM2_EXP & 8) { _Pragma("unroll") for (int e = 0; e < 16; ++e) mm[e] = p0[e]; }
                (void)pre;
                // Hand-placed stream: after every MFMA of the chain one gap's worth of the other work of this sub-tile --
                // a fragment read, one or more steps of the PREVIOUS sub-tile's min trees, one DMA piece of tile
                // t+AHEAD -- so that a wave paces the matrix pipe on its own and the two waves of a SIMD mesh MFMA by
                // MFMA, whatever their relative phase.  Chain order: hi*hi, hi(ref)*lo(query), lo(ref)*hi(query).
                //   NP < 3: the hi fragments are double-buffered (ah / al registers): those of the next sub-tile are read
                //           in the first KSTEPS gaps.
                //   NP = 3: the lo fragments of THIS sub-tile are read in the first KSTEPS gaps (they are needed from MFMA
                //           4*KSTEPS on), the hi fragments of the NEXT sub-tile in the gaps after the last MFMA that reads
                //           the current ones.
                // (At the last sub-tile of the last tile the next-sub-tile reads hit a stale buffer; never used.)
                {
                    constexpr int NM1 = 2 * KSTEPS;            // MFMAs per split product
                    constexpr int NMT = NM1 * NP;              // MFMAs of the whole chain
                    constexpr int PER = (16 + (NMT < 16 ? NMT : 16) - 1) / (NMT < 16 ? NMT : 16);   // min-tree steps per gap
                    half8(&cur)[KSTEPS] = (!LO3 && !SB && (st & 1)) ? reinterpret_cast<half8(&)[KSTEPS]>(al) : ah;
                    half8(&nxt)[KSTEPS] = (!LO3 && !SB && (st & 1)) ? ah : (SB ? ah : reinterpret_cast<half8(&)[KSTEPS]>(al));
                    wait_lds<0>(cur);
                    if constexpr (SIDE) {   // both accumulators start from the rows' sc^2|y|^2 (C operand of the first MFMAs)
                        asm volatile("" : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3));   // (ordered after the wait above)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            acc0[e] = y0[e]; acc0[4 + e] = y1[e]; acc0[8 + e] = y2[e]; acc0[12 + e] = y3[e];
                        }
                        acc1 = acc0;
                    }
#pragma unroll
                    for (int i = 0; i < NMT; ++i) {
                        const int pr = i / NM1, s = (i % NM1) >> 1, jq = i & 1;
                        if constexpr (LO3) { if (i == 2 * NM1) wait_lds<0>(al); }
                        if (pr == 2) {
                            if (jq) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], qh[1][s], acc1, 0, 0, 0);
                            else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], qh[0][s], acc0, 0, 0, 0);
                        } else if (pr == 1) {
                            if (jq) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[s], ql[1][s % KL], acc1, 0, 0, 0);
                            else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[s], ql[0][s % KL], acc0, 0, 0, 0);
                        } else if constexpr (AQ) {
                            if (jq) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc1) : "v"(cur[s]), "a"(qh[1][s]));
                            else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc0) : "v"(cur[s]), "a"(qh[0][s]));
                        } else {
                            if (jq) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[s], qh[1][s], acc1, 0, 0, 0);
                            else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[s], qh[0][s], acc0, 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if (!(CM2_EXP & 4)) {
                            if (LO3) {
                                if (i < KSTEPS) CM2_DSR128X(al[i], arow, (GRAN / 2) * 16 + 32 * i, tile, st * 32 * ROWB);
                                if (i >= 2 * NM1 && i < 2 * NM1 + KSTEPS)
                                    CM2_DSR128X(ah[i - 2 * NM1], arow, 32 * (i - 2 * NM1), ntile, nst * 32 * ROWB);
                            } else {
                                if (SB) { if (i & 1) CM2_DSR128X(ah[i >> 1], arow, 32 * (i >> 1), ntile, nst * 32 * ROWB); }
                                else if (i < KSTEPS) CM2_DSR128X(nxt[i], arow, 32 * i, ntile, nst * 32 * ROWB);
                                if (SIDE && i == KSTEPS) CM2_LOAD_SIDE(ntile, nst);   // (y0..y3 were consumed by MFMA 0 and 1)
                            }
                        }
#pragma unroll
                        for (int jj = 0; jj < PER; ++jj)
                            if (i * PER + jj < 16 && (!(CM2_EXP & 8) || i * PER + jj >= 14)) min_step(i * PER + jj, p0, p1, mm);
                        if (i == NMT - 1 && st < NPIECE && !(CM2_EXP & 1)) {
                            if (t + AHEAD < t1) {
                                CM2_DMA_PIECE(ptile_a, (t + AHEAD) % RING, st);
                                if (SIDE && st == 0 && wave0) CM2_DMA_SIDE(ptile_a, (t + AHEAD) % RING);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if constexpr (SEEDL > 0) {
                    float v0_ = mm[14], v1_ = mm[15];
#pragma unroll
                    for (int i = 0; i < SEEDL; ++i) {   // sorted insertion (ascending lists)
                        const float l0_ = fminf(sl0[i], v0_), l1_ = fminf(sl1[i], v1_);
                        v0_ = fmaxf(sl0[i], v0_); v1_ = fmaxf(sl1[i], v1_);
                        sl0[i] = l0_; sl1[i] = l1_;
                    }
                } else if (CM2_EXP & 8) {
                    float mp0 = mm[14], mp1 = mm[15];
                    if (!(a.dbg & 2)) CM2_FILTER_A(p0, p1, mp0, mp1, prbase);
                } else if (!(a.dbg & 2)) CM2_FILTER_B(p0, p1, mm, prbase);
                p0 = acc0;
                p1 = acc1;
                prbase = ptile * BR + st * 32 + 4 * hh;
            }
            if constexpr (SEEDL == 0) {
            if (a.no_sched) {
                // Seeded scan: no refresh schedule, and merges that each wave starts on its own leave the other waves
                // waiting at the next tile barrier.  So a wave with a log more than 48 entries long announces it (tile index in an
                // LDS word behind the ring, visible after the barrier at the top of the next tile) and at the end of
                // that next tile EVERY wave of the workgroup merges its longer logs together.
                const unsigned fl_ = ldsbase + RING * TILEB + 16;
                if (__ballot(cnt[0] > 48 || cnt[1] > 48) != 0ull) {   // (48: measured 32 / 48 / 56 -- flat above 48; a tile adds <= 64 per log)
                    const int tn_ = t + 1;
                    asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(fl_ + 4 * (tn_ & 1)), "v"(tn_) : "memory");
                }
                int f_;
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(f_) : "v"(fl_ + 4 * (t & 1)) : "memory");
                if (__builtin_amdgcn_readfirstlane(f_) == t) next_refresh = t;
            }
            CM2_REFRESH_T(t, t0, CL - 16 * NST);
            if (a.no_sched) next_refresh = 0x7fffffff;
            }
            ptile += pstep; if (ptile >= a.n_tiles) ptile -= a.n_tiles;
            ptile_a += pstep; if (ptile_a >= a.n_tiles) ptile_a -= a.n_tiles;
        }
        if constexpr (SEEDL > 0) {   // last sub-tile, then the seed of both queries of the lane
            float v0_, v1_;
            CM2_MIN16(p0, v0_);
            CM2_MIN16(p1, v1_);
#pragma unroll
            for (int i = 0; i < SEEDL; ++i) {
                const float l0_ = fminf(sl0[i], v0_), l1_ = fminf(sl1[i], v1_);
                v0_ = fmaxf(sl0[i], v0_); v1_ = fmaxf(sl1[i], v1_);
                sl0[i] = l0_; sl1[i] = l1_;
            }
            const float s0_ = fmaxf(sl0[SEEDL - 1], __shfl_xor(sl0[SEEDL - 1], 32));
            const float s1_ = fmaxf(sl1[SEEDL - 1], __shfl_xor(sl1[SEEDL - 1], 32));
            if (hh == 0) {
                a.cthr[(qbase + c) * a.S + split] = s0_;
                a.cthr[(qbase + 32 + c) * a.S + split] = s1_;
            }
        } else if (!(a.dbg & 2)) {  // drain the pipeline: filter the last sub-tile
            float mp0, mp1;
            CM2_MIN16(p0, mp0);
            CM2_MIN16(p1, mp1);
            CM2_FILTER_A(p0, p1, mp0, mp1, prbase);
        }
        if (a.dbg & 2) asm volatile("" ::"v"(p0[0]), "v"(p1[0]), "v"(p0[15]), "v"(p1[15]));
#undef CM2_LOAD_AH
#undef CM2_LOAD_SIDE
        }   // (runs of the item)

        const unsigned long long tw2_ = (a.dbg & 4) ? __builtin_amdgcn_s_memtime() : 0ull;
        if constexpr (SEEDL == 0) CM2_FINAL();
        if (a.dbg & 4) {
            const unsigned long long tw3_ = __builtin_amdgcn_s_memtime();
            c_procyc += tw1_ - tw0_; c_loopcyc += tw2_ - tw1_; c_fincyc += tw3_ - tw2_;
        }
    }
#undef CM2_DMA_TILE
#undef CM2_DMA_PIECE
#undef CM2_DMA_SIDE
#undef CM2_PT
    if ((a.dbg & 4) && lane == 0) {
        atomicAdd(&a.ctr[0], c_br);
        atomicAdd(&a.ctr[1], c_app);
        atomicAdd(&a.ctr[2], c_pr);
        atomicAdd(&a.ctr[3], c_prcyc);
        atomicAdd(&a.ctr[4], __builtin_amdgcn_s_memtime() - c_t0);
        atomicAdd(&a.ctr[5], 1ull);
        atomicAdd(&a.ctr[6], c_evcyc);
        atomicAdd(&a.ctr[7], c_barcyc);
        atomicAdd(&a.ctr[8], c_dmacyc);
        atomicAdd(&a.ctr[9], c_clus);
        atomicAdd(&a.ctr[10], c_procyc);
        atomicAdd(&a.ctr[11], c_loopcyc);
        atomicAdd(&a.ctr[12], c_fincyc);
    }
}

// the kernels proper: the generic family, and the production form of the K = 128 hi*hi level (config 4), whose register budget
// is split 192 VGPRs + 64 accumulation registers (the query fragments) by the attribute -- left alone the compiler halves
// the 256 registers of a two-waves-per-SIMD kernel between the two files and spills
template <int KSTEPS, int NP, bool CEN = false, int SEEDL = 0>
__global__ __launch_bounds__(NT, 2) void knn_cand_f16(P1Args a) {
    knn_cand_f16_body<KSTEPS, NP, CEN, SEEDL, false>(a);
}
template <int KSTEPS>
__global__ __launch_bounds__(NT, 2) void knn_cand_f16p(P1Args a) {
    knn_cand_f16_body<KSTEPS, 1, false, 0, true>(a);
}


// ---------------------------------------------------------------------------------------------------------------
// Low-dimensional variant (d <= 15, e.g. spatial coordinates): the candidate filter is computed on the VALU as the
// DIRECT squared difference sum_c (q_c - y_c)^2 in fp32.  For such inputs the |y|^2 - 2q.y form loses everything to
// cancellation (coordinates ~1e3, neighbour distances ~1), while the direct form has a purely RELATIVE error
// (~(D+2)*2^-24), so the certificate in knn_rerank stays tight.  Same workgroup geometry, lane <-> (query, row)
// mapping, candidate logs, merges and refresh schedule as knn_cand_f16; reference rows are plain [row][D] floats in
// 512-row tiles copied by LDS-DMA, read back as half-wave broadcasts.
constexpr int BRL = 512;  // reference rows per tile of the low-d kernel (d <= 15); wider rows use 128-row tiles
__host__ __device__ constexpr int lowd_rows(int D) { return D <= 15 ? BRL : 128; }
__host__ __device__ constexpr int lowd_tile_bytes(int D) { return lowd_rows(D) * D * 4; }

// OP: 0 = sum (q - y)^2 (Euclidean), 1 = sum |q - y| (Manhattan), 2 = max |q - y| (Chebyshev: exact in fp32 for fp32 rows),
// 3 = Canberra, 4 = Bray-Curtis (round 4).
// Rows wider than 64 floats (65 <= d <= 128: the direct level of the K = 128 family, and the other metrics at those widths):
// ONE query per lane instead of two (128 VGPRs of coordinates), 32 queries per wave, 256 per workgroup (LOWD_QB).
__host__ __device__ constexpr int lowd_nj(int D) { return D <= 64 ? 2 : 1; }
__host__ __device__ constexpr int lowd_qb(int D) { return NW * 32 * lowd_nj(D); }   // queries per workgroup pass
template <int D, int OP = 0>
__global__ __launch_bounds__(NT, 2) void knn_cand_lowd(P1Args a) {
    constexpr int TILEB = lowd_tile_bytes(D);
    constexpr int NCHUNK = TILEB / 1024;
    constexpr int BRLD = lowd_rows(D);   // rows per tile
    constexpr int NST = BRLD / 32;
    constexpr int NJ_ = lowd_nj(D);
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char *ring = smem;  // [2][TILEB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, hh = lane >> 5;
    uint2 *wsc = nullptr;
    uint2 *wkept = a.kept + (size_t(blockIdx.x) * NW + wave) * 64 * a.Kc;
    uint2 *wlbuf = a.lbuf + (size_t(blockIdx.x) * NW + wave) * 64 * 2 * CL;
    const float *Qf = reinterpret_cast<const float *>(a.Q);
    unsigned long long c_br = 0, c_app = 0, c_pr = 0, c_prcyc = 0, c_evcyc = 0, c_barcyc = 0, c_dmacyc = 0, c_clus = 0;
    (void)wsc;

    const int n_work = a.wlist ? (a.n_wlist_dev ? *a.n_wlist_dev : a.n_wlist) : a.nqb * a.S;
    auto next_work = [&](int w) -> int {
        if (!a.wctr) return w + (int)gridDim.x;
        volatile int *slot = reinterpret_cast<volatile int *>(smem + 2 * TILEB);   // spare bytes behind the ring
        __syncthreads();
        if (tid == 0) *slot = (int)gridDim.x + atomicAdd(a.wctr, 1);
        __syncthreads();
        return __builtin_amdgcn_readfirstlane(*slot);
    };
    for (int work = blockIdx.x; work < n_work; work = next_work(work)) {
        int qb, split, t0, t1;
        if (a.wlist) {   // bucket pruning: explicit (query block, tile run, candidate-list slot) items
            const int4 w_ = a.wlist[work];
            qb = __builtin_amdgcn_readfirstlane(w_.x); t0 = __builtin_amdgcn_readfirstlane(w_.y);
            t1 = __builtin_amdgcn_readfirstlane(w_.z); split = __builtin_amdgcn_readfirstlane(w_.w);
        } else {
            qb = work / a.S; split = work - qb * a.S;
            t0 = split * a.tiles_per_split;
            t1 = (t0 + a.tiles_per_split < a.n_tiles) ? t0 + a.tiles_per_split : a.n_tiles;
        }
        const int64_t qbase = (int64_t)qb * lowd_qb(D) + wave * (32 * NJ_);
        float q0[D], q1[NJ_ == 2 ? D : 1];
#pragma unroll
        for (int e = 0; e < D; ++e) {
            q0[e] = Qf[(qbase + c) * D + e];
            if (NJ_ == 2) q1[e] = Qf[(qbase + 32 + c) * D + e];
        }
        float thr[2] = {INFINITY, INFINITY};
        if (a.thr0_slot0) { thr[0] = a.cthr[(qbase + c) * a.S]; if (NJ_ == 2) thr[1] = a.cthr[(qbase + 32 + c) * a.S]; }
        int cnt[2] = {0, 0};
        int keptn[2] = {0, 0};

#define CM2_DMA_TILE_L(t_, buf_)                                                                                \
    {                                                                                                           \
        const char *gsrc_ = a.R + (size_t)(t_) * TILEB + lane * 16;                                             \
        unsigned char *ldst_ = ring + (buf_) * TILEB;                                                           \
        _Pragma("unroll") for (int q_ = 0; q_ < (NCHUNK + NW - 1) / NW; ++q_) {                                 \
            const int piece_ = wave + q_ * NW;                                                                  \
            if (piece_ < NCHUNK)                                                                                \
                __builtin_amdgcn_global_load_lds((const CM2_AS1 void *)(gsrc_ + piece_ * 1024),                 \
                                                 (CM2_AS3 void *)(ldst_ + piece_ * 1024), 16, 0, 0);            \
        }                                                                                                       \
    }
        __builtin_amdgcn_s_barrier();
        if (t0 < t1) CM2_DMA_TILE_L(t0, t0 & 1);
        const int sched_lim = 0;
        int next_refresh = t0;  // tiles are 512 rows: refresh after 1, 2, 4, ... tiles
        for (int t = t0; t < t1; ++t) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (t + 1 < t1) CM2_DMA_TILE_L(t + 1, (t + 1) & 1);
            const float *tile = reinterpret_cast<const float *>(ring + (t & 1) * TILEB);
#pragma unroll 1
            for (int st = 0; st < NST; ++st) {
                f32x16 acc0, acc1;
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) {
                    const float *y = tile + (st * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * hh) * D;  // half-wave broadcast
                    float s0 = 0.f, s1 = 0.f, e0 = 0.f, e1 = 0.f;   // (e0 / e1: Bray-Curtis denominators)
#pragma unroll
                    for (int e = 0; e < D; ++e) {
                        const float ye = y[e];
                        const float d0 = q0[e] - ye, d1 = (NJ_ == 2 ? q1[e] : 0.f) - ye;
                        if (OP == 0) { s0 = fmaf(d0, d0, s0); if (NJ_ == 2) s1 = fmaf(d1, d1, s1); }
                        else if (OP == 1) { s0 += fabsf(d0); if (NJ_ == 2) s1 += fabsf(d1); }
                        else if (OP == 2) { s0 = fmaxf(s0, fabsf(d0)); if (NJ_ == 2) s1 = fmaxf(s1, fabsf(d1)); }
                        else if (OP == 3) {   // Canberra: |q - y| / (|q| + |y|), a 0 / 0 term counts 0 (padded columns are 0 on both sides)
                            const float n0 = fabsf(q0[e]) + fabsf(ye);
                            s0 += n0 > 0.f ? __fdividef(fabsf(d0), n0) : 0.f;
                            if (NJ_ == 2) { const float n1 = fabsf(q1[e]) + fabsf(ye); s1 += n1 > 0.f ? __fdividef(fabsf(d1), n1) : 0.f; }
                        } else {              // Bray-Curtis: numerator and denominator summed separately
                            s0 += fabsf(d0); e0 += fabsf(q0[e]) + fabsf(ye);
                            if (NJ_ == 2) { s1 += fabsf(d1); e1 += fabsf(q1[e]) + fabsf(ye); }
                        }
                    }
                    // (a padded reference row is +inf in its real columns: inf / inf = NaN, which no comparison passes)
                    if (OP == 4) { s0 = e0 > 0.f ? __fdividef(s0, e0) : 0.f; if (NJ_ == 2) s1 = e1 > 0.f ? __fdividef(s1, e1) : 0.f; }
                    acc0[rr] = s0;
                    acc1[rr] = NJ_ == 2 ? s1 : INFINITY;   // (one query per lane: the second accumulator never passes)
                }
                float mp0, mp1;
                CM2_MIN16(acc0, mp0);
                CM2_MIN16(acc1, mp1);
                const int rbase = t * BRLD + st * 32 + 4 * hh;
                CM2_FILTER(acc0, acc1, mp0, mp1, rbase);
            }
            CM2_REFRESH(t, t0);
        }
#undef CM2_DMA_TILE_L
        CM2_FINAL();
    }
    (void)c_br; (void)c_app; (void)c_pr; (void)c_prcyc; (void)c_evcyc; (void)c_barcyc; (void)c_dmacyc; (void)c_clus;
}

// ---------------------------------------------------------------------------------------------------------------
// Wide variant (62 <= d <= 128): K = 128 feature slots.  The 64-query fragment set of knn_cand_f16 would need 128
// VGPRs here, so a wave owns 32 queries (one 32x32 accumulator; hi/lo fragments = 64 VGPRs) and a workgroup has 8
// waves (one workgroup per CU, 256 queries) so that a 64 KB tile still feeds 256 queries.  Rows are 512 B (hi | lo) and
// use all 128 slots, so |y|^2 cannot ride in spare slots: each tile image carries a block of 128 floats sc^2|y|^2 that
// is added to the accumulator when the sub-tile is filtered (in the shadow of the next MFMA chain).  Everything else --
// LDS-DMA ring, swizzle, pipelined filter, lane-private logs, refresh schedule -- is shared with knn_cand_f16.
constexpr int NWW = 8;            // waves per workgroup of the wide kernel
constexpr int NTW = NWW * 64;
constexpr int QBW = NWW * 32;     // queries per workgroup pass
constexpr int KSTEPS_W = 8;
__host__ __device__ constexpr int wide_tile_bytes() { return BR * row_bytes(KSTEPS_W) + 1024; }

template <int KSTEPS, bool LO3>
__global__ __launch_bounds__(NTW, 2) void knn_cand_f16w(P1Args a) {
    constexpr int K = KSTEPS * 16;
    constexpr int ROWB = row_bytes(KSTEPS);
    constexpr int GRAN = ROWB / 16;
    constexpr int TILEB = BR * ROWB + 1024;
    constexpr int NCHUNK = TILEB / 1024;
    constexpr int NST = BR / 32;
    constexpr int NJ_ = 1;

    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char *ring = smem;  // [2][TILEB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, hh = lane >> 5;
    uint2 *wsc = nullptr;
    uint2 *wkept = a.kept + (size_t(blockIdx.x) * NWW + wave) * 64 * a.Kc;
    uint2 *wlbuf = a.lbuf + (size_t(blockIdx.x) * NWW + wave) * 64 * 2 * CL;
    unsigned long long c_br = 0, c_app = 0, c_pr = 0, c_prcyc = 0, c_evcyc = 0, c_barcyc = 0, c_dmacyc = 0, c_clus = 0;
    (void)wsc;

#define CM2_DMA_TILE_W(t_, buf_)                                                                                \
    {                                                                                                           \
        const char *gsrc_ = a.R + (size_t)(t_) * TILEB + lane * 16;                                             \
        unsigned char *ldst_ = ring + (buf_) * TILEB;                                                           \
        _Pragma("unroll") for (int q_ = 0; q_ < (NCHUNK + NWW - 1) / NWW; ++q_) {                               \
            const int piece_ = wave + q_ * NWW;                                                                 \
            if (piece_ < NCHUNK)                                                                                \
                __builtin_amdgcn_global_load_lds((const CM2_AS1 void *)(gsrc_ + piece_ * 1024),                 \
                                                 (CM2_AS3 void *)(ldst_ + piece_ * 1024), 16, 0, 0);            \
        }                                                                                                       \
    }
    // LDS reads as inline asm + explicit waits, see knn_cand_f16
#define CM2_LOAD_AH_W(F, TB, ST)                                                                               \
    {                                                                                                          \
        _Pragma("unroll") for (int s = 0; s < KSTEPS; ++s)                                                     \
            CM2_DSR128X(F[s], arow, 32 * s, TB, (ST) * 32 * ROWB);                                             \
    }
#define CM2_LOAD_AL_W(F, TB, ST)                                                                               \
    {                                                                                                          \
        _Pragma("unroll") for (int s = 0; s < KSTEPS; ++s)                                                     \
            CM2_DSR128X(F[s], arow, (GRAN / 2) * 16 + 32 * s, TB, (ST) * 32 * ROWB);                           \
    }
#define CM2_LOAD_YN_W(TB, ST)                                                                                  \
    {                                                                                                          \
        CM2_DSR128X(y0, yrow, 0, TB, (ST) * 128);                                                              \
        CM2_DSR128X(y1, yrow, 0, TB, (ST) * 128 + 32);                                                         \
        CM2_DSR128X(y2, yrow, 0, TB, (ST) * 128 + 64);                                                         \
        CM2_DSR128X(y3, yrow, 0, TB, (ST) * 128 + 96);                                                         \
    }
    const unsigned ldsbase = (unsigned)(uintptr_t)(CM2_AS3 unsigned char *)ring;
    const unsigned arow = c * ROWB + ((hh ^ swz(c, KSTEPS)) * 16);   // see knn_cand_f16
    const unsigned yrow = BR * ROWB + 16 * hh;   // |y|^2 side block: rows 4*hh + {0..3} + 8*i of a sub-tile

    // work items: (query block, reference split) pairs, or -- bucket pruning (round 5) -- the explicit (query block, tile run,
    // candidate-list slot) items of a work list, as in knn_cand_f16 (static hand-out: the items of a pruned stage are many and short)
    const int n_work = a.wlist ? (a.n_wlist_dev ? *a.n_wlist_dev : a.n_wlist) : a.nqb * a.S;
    for (int work = blockIdx.x; work < n_work; work += gridDim.x) {
        int qb, split, t0, t1;
        if (a.wlist) {
            const int4 w_ = a.wlist[work];
            qb = __builtin_amdgcn_readfirstlane(w_.x); t0 = __builtin_amdgcn_readfirstlane(w_.y);
            t1 = __builtin_amdgcn_readfirstlane(w_.z); split = __builtin_amdgcn_readfirstlane(w_.w);
        } else {
            qb = work / a.S; split = work - qb * a.S;
            t0 = split * a.tiles_per_split;
            t1 = (t0 + a.tiles_per_split < a.n_tiles) ? t0 + a.tiles_per_split : a.n_tiles;
        }
        const int64_t qbase = (int64_t)qb * QBW + wave * 32;
        half8 qh[KSTEPS], ql[KSTEPS];
        {
            const _Float16 *qrow = a.Q + (qbase + c) * (2 * K);
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) {
                qh[s] = *reinterpret_cast<const half8 *>(qrow + (2 * s + hh) * 8);
                ql[s] = *reinterpret_cast<const half8 *>(qrow + K + (2 * s + hh) * 8);
            }
        }
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) asm volatile("" ::"v"(qh[s]), "v"(ql[s]));  // see knn_cand_f16
        // slot 1 of the per-lane state is inert (threshold -inf): the shared macros iterate over two accumulators
        float thr[2] = {INFINITY, -INFINITY};
        if (a.thr0_slot0) {   // pass B of a pruned stage starts from the pass-A threshold of the query (list slot 0)
            thr[0] = a.cthr[(qbase + c) * a.S];
            asm volatile("" : "+v"(thr[0]));
        }
        int cnt[2] = {0, 0};
        int keptn[2] = {0, 0};

        __builtin_amdgcn_s_barrier();
        if (t0 < t1) CM2_DMA_TILE_W(t0, t0 & 1);
        f32x16 p0, py;   // previous sub-tile: raw accumulator (-2 q.y) and its rows' |y|^2
#pragma unroll
        for (int e = 0; e < 16; ++e) { p0[e] = INFINITY; py[e] = 0.f; }
        int prbase = 0;
        const int sched_lim = 0;
        int next_refresh = t0 + 1;
        for (int t = t0; t < t1; ++t) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (t + 1 < t1) CM2_DMA_TILE_W(t + 1, (t + 1) & 1);
            const unsigned tile = ldsbase + (t & 1) * TILEB;
            half8 ah[KSTEPS], al[KSTEPS];   // LO3: hi / lo fragments; otherwise the two hi fragment buffers
            f32x4 y0, y1, y2, y3;
            CM2_LOAD_AH_W(ah, tile, 0);
            if (LO3) CM2_LOAD_AL_W(al, tile, 0);
            CM2_LOAD_YN_W(tile, 0);
#pragma unroll
            for (int st = 0; st < NST; ++st) {
                f32x16 acc0, cy;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc0[e] = 0.f;
                if constexpr (LO3) {
                    wait_lds<KSTEPS + 4>(ah);   // behind them: this sub-tile's lo fragments and |y|^2
#pragma unroll
                    for (int s = 0; s < KSTEPS; ++s) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], qh[s], acc0, 0, 0, 0);
#pragma unroll
                    for (int s = 0; s < KSTEPS; ++s) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], ql[s], acc0, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (st + 1 < NST) {
                        CM2_LOAD_AH_W(ah, tile, st + 1);
                        wait_lds<KSTEPS>(al);
                    } else {
                        wait_lds<0>(al);
                    }
                    asm volatile("" : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3));
#pragma unroll
                    for (int e = 0; e < 4; ++e) { cy[e] = y0[e]; cy[4 + e] = y1[e]; cy[8 + e] = y2[e]; cy[12 + e] = y3[e]; }
#pragma unroll
                    for (int s = 0; s < KSTEPS; ++s) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], qh[s], acc0, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (st + 1 < NST) {
                        CM2_LOAD_AL_W(al, tile, st + 1);
                        CM2_LOAD_YN_W(tile, st + 1);
                    }
                } else {
                    half8(&cur)[KSTEPS] = (st & 1) ? al : ah;
                    half8(&nxt)[KSTEPS] = (st & 1) ? ah : al;
                    wait_lds<0>(cur);
                    asm volatile("" : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3));
#pragma unroll
                    for (int e = 0; e < 4; ++e) { cy[e] = y0[e]; cy[4 + e] = y1[e]; cy[8 + e] = y2[e]; cy[12 + e] = y3[e]; }
                    if (st + 1 < NST) {   // requested before this sub-tile's chain is issued
                        CM2_LOAD_AH_W(nxt, tile, st + 1);
                        CM2_LOAD_YN_W(tile, st + 1);
                    }
#pragma unroll
                    for (int s = 0; s < KSTEPS; ++s) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[s], qh[s], acc0, 0, 0, 0);
                    if (a.nprod >= 2) {
#pragma unroll
                        for (int s = 0; s < KSTEPS; ++s) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[s], ql[s], acc0, 0, 0, 0);
                    }
                }
                // filter the PREVIOUS sub-tile in the shadow of the chain above: value = |y|^2 - 2 q.y
                f32x16 pv;
#pragma unroll
                for (int e = 0; e < 16; ++e) pv[e] = p0[e] + py[e];
                float mp0;
                CM2_MIN16(pv, mp0);
                const float mp1 = INFINITY;
                CM2_FILTER_A(pv, pv, mp0, mp1, prbase);
                p0 = acc0;
                py = cy;
                prbase = t * BR + st * 32 + 4 * hh;
            }
            CM2_REFRESH_T(t, t0, CL - 16 * NST);
        }
        {
            f32x16 pv;
#pragma unroll
            for (int e = 0; e < 16; ++e) pv[e] = p0[e] + py[e];
            float mp0;
            CM2_MIN16(pv, mp0);
            const float mp1 = INFINITY;
            CM2_FILTER_A(pv, pv, mp0, mp1, prbase);
        }
        CM2_FINAL();
    }
#undef CM2_DMA_TILE_W
#undef CM2_LOAD_AH_W
#undef CM2_LOAD_AL_W
#undef CM2_LOAD_YN_W
    (void)c_br; (void)c_app; (void)c_pr; (void)c_prcyc; (void)c_evcyc; (void)c_barcyc; (void)c_dmacyc; (void)c_clus;
}

}  // namespace cmk2
