// Phase 1, hi*hi level at K = 64 on v_mfma_f32_16x16x32_f16 (round 4): the full-set scan of large query sets, re-tiled
// from the 32x32x16 shape of knn_cand_f16q (knn_phase1_q4.h) to the 16x16x32 shape.
//
// Why it was built: the round-3 review asked for it (MI355X_MICROARCH.md, "DVFS give-back" item 7: 12-15 % more FLOP/s per
// watt on this shape), and the in-place probe -- every 32x32x16 of knn_cand_f16q replaced by two independent 16x16x32 of the
// same flops, events off -- ran 199 -> 187 ms at config 2.  What it measures as a real kernel
// (profiles/r04_mfma_shape_probe.md, same box, alternating): MFMA + fragment reads + DMA 172 against 185 ms (-7 %), with the
// min trees 185 against ~197, with the threshold tests 197-201 against 199-200, in production form 232-236 against
// 231-236: LEVEL.  The shape doubles the matrix instructions, each of which holds the SIMD's vector issue port for 8
// cycles (8 of 16 instead of 8 of 32): with two waves per SIMD the hot loop's ~46 vector instructions per sub-tile and wave
// need 84 % of the issue slots that are left (53 % on the 32x32x16 shape), and what the matrix pipe gains in clock the
// filter loses in issue stalls.  The kernel is exact and tested and stays selectable (CELLMAP_R16=1); production keeps
// knn_cand_f16q.
//
// Same contract, tile image, LDS ring, DMA, seeds, scattered tile order, kept lists and outputs as knn_cand_f16q; what changes
// is the lane <-> (query, row) mapping and everything that hangs on it:
//   * D = A(16 rows x 32 k) . B(32 k x 16 queries): lane l holds query c = l & 15 of a 16-query tile and rows 4g .. 4g+3
//     (g = l >> 4) of a 16-row tile; as an operand it supplies k-chunk g (8 halves) of row / query c.  A wave still owns
//     128 queries = 8 query tiles (64 VGPRs of hi fragments: 8 tiles x 2 k-steps) and a sub-tile is still 32 reference
//     rows = 2 row halves (16 VGPRs per fragment set), so one sub-tile is 16 output tiles x 2 k-steps = 32 MFMAs of 16
//     cycles -- the pipe time of the 16 MFMAs of 32 cycles it replaces.  A lane's fragment reads hit row c (+16) at granule
//     (4 s2 + g) ^ swizzle: conflict-free on the existing hi-only tile image (checked by enumeration over the ds_read_b128
//     lane groups; the K = 32 image would be 2-way conflicted, so K = 32 stays on knn_cand_f16q).
//   * FOUR lanes share a query (two did): a query has four lane-private logs of CL / 2 entries (same memory), the merge
//     gathers from four logs, thresholds / counts are replicated over the four lanes.
//   * the chain is pair-major: the query tiles are taken two at a time (8 MFMAs = the pipe time of one accumulator chain
//     of knn_cand_f16q); the two min trees of pair (jj + 2) & 3 -- complete at least 8 MFMAs earlier -- are dealt two
//     steps per gap over them (8 values per lane and tile: 4 v_min3), ONE threshold test per pair and the (rare) events
//     follow; pairs 2 and 3 of a sub-tile are filtered during the first half of the next one.
//   * an event covers 512 values instead of 1 024 and descends through 2 partial minima + 2 registers (4 group tests).
#pragma once
#include "knn_phase1_f16.h"

namespace cmk2 {

constexpr int CLR = CL / 2;   // entries of one lane-private log of knn_cand_f16r (four logs per query)
#ifndef CM2R_TEST
#define CM2R_TEST 1   // 1: pair test through one VCC compare (v_cndmask fold), 0: two compares + s_or
#endif
#ifndef CM2R_EXP
#define CM2R_EXP 0   // timing experiments only: 1 no DMA pieces, 2 no barrier, 4 no fragment reads, 8 no min trees, 16 no threshold
                     // test, 64 thresholds at -inf (never append)
#endif

// entry i of the concatenation (kept list, log 0, log 1, log 2, log 3) goes to lane i % 64, slot i / 64
template <int E>
__device__ __forceinline__ void prune_load4(const uint2 *kept_src, int nk, const uint2 *lb, int n0, int n1, int n2, int n3, int lane,
                                            unsigned (&key)[E], unsigned (&row)[E]) {
    const int n = nk + n0 + n1 + n2 + n3;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = e * 64 + lane;
        key[e] = 0xffffffffu;
        row[e] = 0;
        if (i < n) {
            int r = i - nk;
            const uint2 *src;
            if (r < 0) src = kept_src + i;
            else if (r < n0) src = lb + r;
            else if ((r -= n0) < n1) src = lb + CLR + r;
            else if ((r -= n1) < n2) src = lb + 2 * CLR + r;
            else src = lb + 3 * CLR + (r - n2);
            uint2 x = ld_fresh(src);
            key[e] = f2key(__uint_as_float(x.x));
            row[e] = x.y;
        }
    }
}
template <int E>
__device__ __forceinline__ int wave_merge_prune4_e(const uint2 *kept_src, int nk, const uint2 *lb, int n0, int n1, int n2, int n3, int Kc,
                                                   uint2 *dst_pairs, int *dst_rows, float *thr_out, int lane) {
    unsigned key[E], row[E];
    prune_load4<E>(kept_src, nk, lb, n0, n1, n2, n3, lane, key, row);
    return prune_select<E>(key, row, nk + n0 + n1 + n2 + n3, Kc, dst_pairs, dst_rows, thr_out, lane);
}
__device__ __forceinline__ int wave_merge_prune4(const uint2 *kept_src, int nk, const uint2 *lb, int n0, int n1, int n2, int n3, int Kc,
                                                 uint2 *dst_pairs, int *dst_rows, float *thr_out, int lane) {
    const int n = nk + n0 + n1 + n2 + n3;
    if (n <= 128) return wave_merge_prune4_e<2>(kept_src, nk, lb, n0, n1, n2, n3, Kc, dst_pairs, dst_rows, thr_out, lane);
    if (n <= 256) return wave_merge_prune4_e<4>(kept_src, nk, lb, n0, n1, n2, n3, Kc, dst_pairs, dst_rows, thr_out, lane);
    return wave_merge_prune4_e<8>(kept_src, nk, lb, n0, n1, n2, n3, Kc, dst_pairs, dst_rows, thr_out, lane);
}

#define CM2R_APPEND(V, ROFF, RB, LOFF)                                                                        \
    if ((V) < tj_) {                                                                                           \
        *reinterpret_cast<uint2 *>(reinterpret_cast<char *>(wlbuf) + (size_t)((LOFF) + 8u * (unsigned)cj_)) =  \
            make_uint2(__float_as_uint(V), (unsigned)((RB) + (ROFF)));                                         \
        ++cj_;                                                                                                 \
    }
// One query tile's event: A0 / A1 = its two accumulators (row halves), U0 = min(A0[0..2]), U1 = min(A1[0..2]).
#define CM2R_EVENT(A0, A1, U0, U1, J, RB)                                                                      \
    {                                                                                                          \
        unsigned long long te0_ = 0;                                                                           \
        if (adbg & 4) { ++c_br; te0_ = __builtin_amdgcn_s_memtime(); }                                        \
        const float tj_ = thr[J];                                                                              \
        int cj_ = cnt[J];                                                                                      \
        const unsigned loff_ = (unsigned)((((J) * 16 + c) * 4 + g) * CLR * 8);                                 \
        const unsigned long long g0_ = __ballot((U0) < tj_), g1_ = __ballot((U1) < tj_);                       \
        const unsigned long long g2_ = __ballot((A0)[3] < tj_), g3_ = __ballot((A1)[3] < tj_);                 \
        if (g0_) { CM2R_APPEND((A0)[0], 0, RB, loff_) CM2R_APPEND((A0)[1], 1, RB, loff_) CM2R_APPEND((A0)[2], 2, RB, loff_) } \
        if (g1_) { CM2R_APPEND((A1)[0], 16, RB, loff_) CM2R_APPEND((A1)[1], 17, RB, loff_) CM2R_APPEND((A1)[2], 18, RB, loff_) } \
        if (g2_) CM2R_APPEND((A0)[3], 3, RB, loff_)                                                            \
        if (g3_) CM2R_APPEND((A1)[3], 19, RB, loff_)                                                           \
        cnt[J] = cj_;                                                                                          \
        if (adbg & 4) c_evcyc += __builtin_amdgcn_s_memtime() - te0_;                                         \
    }

// DBG = false is the production body (the CELLMAP_DBG modes are compiled out, as in knn_cand_f16q).
template <bool DBG>
__global__ __launch_bounds__(NT, 2) void knn_cand_f16r(P1Args a) {
    const int adbg = DBG ? a.dbg : 0;
    constexpr int KSTEPS = 4, KS2 = 2;   // K = 64: four 16-wide steps of the tile image = two 32-wide MFMA steps
    constexpr int K = KSTEPS * 16;
    constexpr int ROWB = img_row_bytes(KSTEPS, true);
    constexpr int GTILEB = img_tile_bytes(KSTEPS, true);
    constexpr int TILEB = GTILEB;
    constexpr int ROWS_B = BR * ROWB;
    constexpr int RING = ring_depth(true, KSTEPS);
    constexpr int BT = RING >= 8 ? CM2_BT : 1;
    constexpr int AHEAD = RING - BT;
    static_assert(AHEAD - 1 - BT >= 0, "ring too shallow for this barrier interval");
    constexpr int NCHUNK = ROWS_B / 1024;
    constexpr int NST = BR / 32;
    static_assert(NCHUNK % NW == 0 && NST % 2 == 0, "tile geometry");
    constexpr int NPIECE = NCHUNK / NW;
    static_assert(NPIECE <= NST, "one DMA piece per sub-tile");
    constexpr int NJ = 8;          // 16-query tiles per wave
    constexpr int NF = 2 * KS2;    // fragments per set: [row half][k-step]

    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char *ring = smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    uint2 *wkept = a.kept + (size_t(blockIdx.x) * NW + wave) * 128 * a.Kc;
    uint2 *wlbuf = a.lbuf + (size_t(blockIdx.x) * NW + wave) * 128 * 2 * CL;   // = 128 queries x 4 logs x CLR entries
    const unsigned ldsbase = (unsigned)(uintptr_t)(CM2_AS3 unsigned char *)ring;
    const unsigned dma_voff = wave * 1024 + lane * 16;

#define CM2R_DMA_TILE(t_, buf_)                                                                                 \
    {                                                                                                           \
        const char *gsrc_ = a.R + (size_t)(t_) * GTILEB;                                                        \
        unsigned char *ldst_ = ring + (buf_) * TILEB + wave * 1024;                                             \
        _Pragma("unroll") for (int q_ = 0; q_ < NPIECE; ++q_)                                                   \
            __builtin_amdgcn_global_load_lds((const CM2_AS1 void *)(gsrc_ + q_ * (NW * 1024) + dma_voff),       \
                                             (CM2_AS3 void *)(ldst_ + q_ * (NW * 1024)), 16, 0, 0);             \
    }
#define CM2R_DMA_PIECE(t_, buf_, q_)                                                                            \
    __builtin_amdgcn_global_load_lds(                                                                           \
        (const CM2_AS1 void *)(a.R + (size_t)(t_) * GTILEB + (q_) * (NW * 1024) + dma_voff),                    \
        (CM2_AS3 void *)(ring + (buf_) * TILEB + wave * 1024 + (q_) * (NW * 1024)), 16, 0, 0)
#define CM2R_PT(t_) (a.perm_mul > 1u ? (int)(((unsigned long long)(unsigned)(t_) * a.perm_mul) % (unsigned)a.n_tiles) : (int)(t_))

    unsigned long long c_br = 0, c_app = 0, c_pr = 0, c_prcyc = 0, c_evcyc = 0, c_barcyc = 0, c_dmacyc = 0;
    unsigned long long c_procyc = 0, c_loopcyc = 0, c_fincyc = 0;
    const unsigned long long c_t0 = DBG ? __builtin_amdgcn_s_memtime() : 0ull;
    const int n_work = a.nqb * a.S;
    for (int work = blockIdx.x; work < n_work; work += (int)gridDim.x) {
        const int qb = work / a.S, split = work - qb * a.S;
        const int t0 = split * a.tiles_per_split;
        const int t1 = (t0 + a.tiles_per_split < a.n_tiles) ? t0 + a.tiles_per_split : a.n_tiles;
        const int64_t qbase = (int64_t)qb * (NW * 128) + wave * 128;
        const unsigned long long tw0_ = (adbg & 4) ? __builtin_amdgcn_s_memtime() : 0ull;

        // ---- query fragments (B operand) of the wave's 128 queries: hi halves, k-chunk g of each 32-wide step
        half8 qh[NJ][KS2];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const _Float16 *qrow = a.Q + (qbase + j * 16 + c) * (2 * K);
#pragma unroll
            for (int s = 0; s < KS2; ++s) qh[j][s] = *reinterpret_cast<const half8 *>(qrow + (4 * s + g) * 8);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int s = 0; s < KS2; ++s) asm volatile("" ::"v"(qh[j][s]));
        const float thr_init = (((adbg & 1) | (CM2R_EXP & 64)) != 0) ? -INFINITY : INFINITY;
        float thr[NJ];
        int cnt[NJ], keptn[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) { thr[j] = thr_init; cnt[j] = 0; keptn[j] = 0; }
        if (a.thr0 && !(CM2R_EXP & 64)) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) thr[j] = a.thr0[qbase + j * 16 + c];
            asm volatile("" : "+v"(thr[0]), "+v"(thr[1]), "+v"(thr[2]), "+v"(thr[3]), "+v"(thr[4]), "+v"(thr[5]), "+v"(thr[6]), "+v"(thr[7]));
        }

        // merge the logs of every query whose lanes hold more than LIM_ entries (runtime loop over the query tiles)
#define CM2R_SEL(ARR, J) ((J) == 0 ? ARR[0] : ((J) == 1 ? ARR[1] : ((J) == 2 ? ARR[2] : ((J) == 3 ? ARR[3] : ((J) == 4 ? ARR[4] : ((J) == 5 ? ARR[5] : ((J) == 6 ? ARR[6] : ARR[7])))))))
#define CM2R_PUT(ARR, J, V)                                                                                    \
    {                                                                                                          \
        if ((J) == 0) ARR[0] = (V); else if ((J) == 1) ARR[1] = (V); else if ((J) == 2) ARR[2] = (V); else if ((J) == 3) ARR[3] = (V); \
        else if ((J) == 4) ARR[4] = (V); else if ((J) == 5) ARR[5] = (V); else if ((J) == 6) ARR[6] = (V); else ARR[7] = (V); \
    }
#define CM2R_MERGE_PASS(LIM_)                                                                                  \
    {                                                                                                          \
        unsigned long long tp0_ = (adbg & 4) ? __builtin_amdgcn_s_memtime() : 0ull;                           \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                       \
        _Pragma("unroll 1") for (int j = 0; j < NJ; ++j) {                                                     \
            int cj_ = CM2R_SEL(cnt, j), kj_ = CM2R_SEL(keptn, j);                                              \
            float tj_ = CM2R_SEL(thr, j);                                                                      \
            unsigned long long need = __ballot(cj_ > (LIM_));                                                  \
            need = (need | (need >> 16) | (need >> 32) | (need >> 48)) & 0xffffull;                            \
            while (need) {                                                                                     \
                const int ql_ = __ffsll((long long)need) - 1;                                                  \
                need &= need - 1;                                                                              \
                const int wq = j * 16 + ql_;                                                                   \
                const int n0 = __shfl(cj_, ql_), n1 = __shfl(cj_, ql_ + 16), n2 = __shfl(cj_, ql_ + 32), n3 = __shfl(cj_, ql_ + 48); \
                const int nk = __shfl(kj_, ql_);                                                               \
                float tq = __shfl(tj_, ql_);                                                                   \
                uint2 *kq = wkept + (size_t)wq * a.Kc;                                                         \
                int nn = wave_merge_prune4(kq, nk, wlbuf + (size_t)(wq * 4) * CLR, n0, n1, n2, n3, a.Kc, kq, nullptr, &tq, lane); \
                if (c == ql_) { cj_ = 0; kj_ = nn; tj_ = tq; }                                                 \
                if (adbg & 4) { ++c_pr; c_app += n0 + n1 + n2 + n3; }                                         \
            }                                                                                                  \
            CM2R_PUT(cnt, j, cj_) CM2R_PUT(keptn, j, kj_) CM2R_PUT(thr, j, tj_)                                \
        }                                                                                                      \
        if (adbg & 4) c_prcyc += __builtin_amdgcn_s_memtime() - tp0_;                                         \
    }

        __builtin_amdgcn_s_barrier();  // previous work item: every wave is done reading the ring
#pragma unroll
        for (int u = 0; u < AHEAD; ++u)
            if (t0 + u < t1) CM2R_DMA_TILE(CM2R_PT(t0 + u), (t0 + u) % RING);

        f32x4 acc[NJ][2];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[j][h2][e] = INFINITY;   // "nothing pending"
        int rb_cur = 0, rb_prev = 0;   // first reference row (+4*g) of the current / previous sub-tile

        // row c of a 16-row half at k-chunk g; the second half is 16 rows = 16 * ROWB bytes further (same swizzle: (row >> 1) & 7)
        const unsigned arow = c * ROWB + ((g ^ swz_b(c, ROWB)) * 16);
        const int sched_lim = a.no_sched ? 4 : 0;   // (four logs per query: half of knn_cand_f16q's 8 per log)
        int next_refresh = a.no_sched ? (a.no_sched == 2 ? t0 : 0x7fffffff) : t0 + 1;
        // fragment M (literal 0..3) = row half M / 2, k-step M % 2 of the sub-tile at LDS address BASE
#define CM2R_FRAG(DST, M, BASE) CM2_DSR128X(DST[M], arow, 64 * ((M) % 2), BASE, 16 * ROWB * ((M) / 2))
        static_assert(NF == 4, "fragment reads are written out for two row halves x two k-steps");
        half8 ah[NF], al[NF];   // the two fragment sets (current / next sub-tile, alternating): [row half * KS2 + k-step]
#pragma unroll
        for (int s = 0; s < NF; ++s) { ah[s] = half8{0, 0, 0, 0, 0, 0, 0, 0}; al[s] = ah[s]; }
        if (t0 < t1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const unsigned tile0 = ldsbase + (t0 % RING) * TILEB;
            CM2R_FRAG(ah, 0, tile0) CM2R_FRAG(ah, 1, tile0) CM2R_FRAG(ah, 2, tile0) CM2R_FRAG(ah, 3, tile0)
        }
        int ptile = CM2R_PT(t0), ptile_a = CM2R_PT(t0 + AHEAD);
        const int pstep = a.perm_mul > 1u ? (int)(a.perm_mul % (unsigned)a.n_tiles) : 1;
        const unsigned long long tw1_ = (adbg & 4) ? __builtin_amdgcn_s_memtime() : 0ull;

        // min tree of query tile P (accumulators P[0], P[1]) in 4 steps: U0 = min of P[0][0..2], U1 = of P[1][0..2]
#define CM2R_TREE_STEP(I, P, U0, U1, U2, TOT)                                                                  \
    {                                                                                                          \
        if ((I) == 0) U0 = vmin3(P[0][0], P[0][1], P[0][2]);                                                   \
        else if ((I) == 1) U1 = vmin3(P[1][0], P[1][1], P[1][2]);                                              \
        else if ((I) == 2) U2 = vmin3(P[0][3], P[1][3], U0);                                                   \
        else TOT = vmin3(U1, U2, U2);                                                                          \
    }
        // One sub-tile: CUR holds its fragments, NXT receives those of the next sub-tile.  The 8 query tiles are taken in PAIRS
        // (32 queries, 8 MFMAs = the pipe time of one accumulator chain of knn_cand_f16q): a gap = two independent MFMAs
        // (both row halves of one query tile and k-step; the dependent k-step follows four MFMAs later), then one fragment
        // read, two tree steps of the pair filtered in the shadow (the one finished 8 MFMAs earlier), ONE threshold test
        // per pair.  (A first version with one MFMA per gap and a test per query tile measured 168 / 183 / 198 ms for the
        // skeleton / + trees / + tests where the in-place probe had 137 / - / 187: profiles/r04_mfma_shape_probe.md.)
#define CM2R_SUBTILE(CUR, NXT, ST)                                                                             \
    {                                                                                                          \
        const int nst_ = ((ST) + 1 < NST) ? (ST) + 1 : 0;                                                      \
        const unsigned nbase_ = (((ST) + 1 < NST) ? tile : tile_n) + nst_ * 32 * ROWB;                         \
        rb_prev = rb_cur;                                                                                      \
        rb_cur = ptile * BR + (ST) * 32 + 4 * g;                                                               \
        wait_lds<0>(CUR);                                                                                      \
        _Pragma("unroll") for (int jj = 0; jj < NJ / 2; ++jj) {                                                \
            const int pp = (jj + 2) & 3;            /* pair filtered in the shadow of pair jj */              \
            const int ja = 2 * pp, jb = 2 * pp + 1;                                                            \
            float ua0_ = INFINITY, ua1_ = INFINITY, ua2_ = INFINITY, ta_ = INFINITY;                           \
            float ub0_ = INFINITY, ub1_ = INFINITY, ub2_ = INFINITY, tb_ = INFINITY;                           \
            const f32x4 zero_ = {0.f, 0.f, 0.f, 0.f};                                                          \
            _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                    \
                const int qt = 2 * jj + (s & 1), s2 = s >> 1;                                                  \
                acc[qt][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(CUR[s2], qh[qt][s2], s2 == 0 ? zero_ : acc[qt][0], 0, 0, 0); \
                acc[qt][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(CUR[KS2 + s2], qh[qt][s2], s2 == 0 ? zero_ : acc[qt][1], 0, 0, 0); \
                __builtin_amdgcn_sched_barrier(0);                                                             \
                if (jj == 0 && !(CM2R_EXP & 4)) {                                                              \
                    if (s == 0) CM2R_FRAG(NXT, 0, nbase_) else if (s == 1) CM2R_FRAG(NXT, 1, nbase_)           \
                    else if (s == 2) CM2R_FRAG(NXT, 2, nbase_) else CM2R_FRAG(NXT, 3, nbase_)                  \
                }                                                                                              \
                if (!(CM2R_EXP & 8)) {                                                                         \
                    CM2R_TREE_STEP(s, acc[ja], ua0_, ua1_, ua2_, ta_)                                          \
                    CM2R_TREE_STEP(s, acc[jb], ub0_, ub1_, ub2_, tb_)                                          \
                }                                                                                              \
                if (!(CM2R_EXP & 1) && jj == 1 && s == 0 && (ST) < NPIECE && t + AHEAD < t1) CM2R_DMA_PIECE(ptile_a, (t + AHEAD) % RING, (ST)); \
                __builtin_amdgcn_sched_barrier(0);                                                             \
            }                                                                                                  \
            if (CM2R_EXP & 8) asm volatile("" ::"v"(acc[ja][0][0]), "v"(acc[ja][1][3]), "v"(acc[jb][0][0]), "v"(acc[jb][1][3])); \
            else if (CM2R_EXP & 16) asm volatile("" ::"v"(ta_), "v"(tb_));                                     \
            if (!(adbg & 2) && !(CM2R_EXP & 16)) {                                                            \
                const bool pa_ = ta_ < thr[ja], pb_ = tb_ < thr[jb];                                           \
                /* the pair's verdict through ONE vector compare into VCC (as knn_cand_f16q's test): the first tile's verdict  \
                   is folded into the second compare's operand by a v_cndmask -- two compares into scalar registers plus an    \
                   s_or put a vector -> scalar -> scalar -> branch chain in front of every pair */                              \
                const float tb2_ = CM2R_TEST == 1 ? (pa_ ? -INFINITY : tb_) : tb_;                                             \
                if ((CM2R_TEST == 1 ? __ballot(tb2_ < thr[jb]) : __ballot(pa_ || pb_)) != 0ull) {                              \
                    const int rbj_ = pp >= 2 ? rb_prev : rb_cur;                                               \
                    if (__ballot(pa_) != 0ull) CM2R_EVENT(acc[ja][0], acc[ja][1], ua0_, ua1_, ja, rbj_)        \
                    if (__ballot(pb_) != 0ull) CM2R_EVENT(acc[jb][0], acc[jb][1], ub0_, ub1_, jb, rbj_)        \
                }                                                                                              \
            }                                                                                                  \
        }                                                                                                      \
    }

        for (int t = t0; t < t1; ++t) {
            unsigned long long tb0_ = (adbg & 4) ? __builtin_amdgcn_s_memtime() : 0ull;
            if (t > t0 && (BT == 1 || (t - t0) % BT == 0)) {
                if (t + AHEAD - 1 >= t1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"i"((AHEAD - 1 - BT) * NPIECE) : "memory");
                if (adbg & 4) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); c_dmacyc += t_ - tb0_; tb0_ = t_; }
                if (!(CM2R_EXP & 2)) __builtin_amdgcn_s_barrier();
                if (adbg & 4) c_barcyc += __builtin_amdgcn_s_memtime() - tb0_;
            }
            const unsigned tile = ldsbase + (t % RING) * TILEB;
            const unsigned tile_n = ldsbase + ((t + 1) % RING) * TILEB;
#pragma unroll
            for (int sp = 0; sp < NST; sp += 2) {
                CM2R_SUBTILE(ah, al, sp)
                CM2R_SUBTILE(al, ah, sp + 1)
            }
            if (a.no_sched) {
                // aligned merges of the seeded scan (see knn_cand_f16): a wave with a long log announces it, everybody merges
                // its longer logs at the end of the following tile
                const unsigned fl_ = ldsbase + RING * TILEB + 16;
                bool long_ = false;
#pragma unroll
                for (int j = 0; j < NJ; ++j) long_ = long_ || cnt[j] > 24;
                if (__ballot(long_) != 0ull) {
                    const int tn_ = t + 1;
                    asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(fl_ + 4 * (tn_ & 1)), "v"(tn_) : "memory");
                }
                int f_;
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(f_) : "v"(fl_ + 4 * (t & 1)) : "memory");
                if (__builtin_amdgcn_readfirstlane(f_) == t) next_refresh = t;
            }
            if (!(adbg & 3)) {   // end of tile: scheduled refresh, or logs that could overflow during the next tile
                constexpr int FULL_ = CLR - 8 * NST;   // a lane appends at most 8 values per sub-tile and query tile
                const bool sched_ = t == next_refresh;
                bool full_ = false;
#pragma unroll
                for (int j = 0; j < NJ; ++j) full_ = full_ || cnt[j] > FULL_;
                if (sched_ || __ballot(full_) != 0ull) {
                    if (sched_) next_refresh = t0 + 2 * (t - t0 + 1) - 1;
                    const int lim_ = sched_ ? sched_lim : FULL_;
                    CM2R_MERGE_PASS(lim_)
                }
            }
            if (a.no_sched) next_refresh = 0x7fffffff;
            ptile += pstep; if (ptile >= a.n_tiles) ptile -= a.n_tiles;
            ptile_a += pstep; if (ptile_a >= a.n_tiles) ptile_a -= a.n_tiles;
        }
        // drain: the query tiles of pairs 2 and 3 of the last sub-tile have not been filtered yet
        if (!(adbg & 2)) {
#pragma unroll
            for (int jp = 4; jp < 8; ++jp) {
                float u0_ = INFINITY, u1_ = INFINITY, u2_ = INFINITY, tot_ = INFINITY;
#pragma unroll
                for (int i = 0; i < 4; ++i) CM2R_TREE_STEP(i, acc[jp], u0_, u1_, u2_, tot_)
                if (__ballot(tot_ < thr[jp]) != 0ull) CM2R_EVENT(acc[jp][0], acc[jp][1], u0_, u1_, jp, rb_cur)
            }
        }
        if (adbg & 2) asm volatile("" ::"v"(acc[0][0][0]), "v"(acc[3][0][0]), "v"(acc[4][1][3]), "v"(acc[7][1][3]));
#undef CM2R_SUBTILE
#undef CM2R_TREE_STEP
#undef CM2R_FRAG

        const unsigned long long tw2_ = (adbg & 4) ? __builtin_amdgcn_s_memtime() : 0ull;
        // final merge: K' candidate rows (-1 padded) and the threshold of every query of the wave
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll 1
        for (int j = 0; j < NJ; ++j) {
            const int cj_ = CM2R_SEL(cnt, j), kj_ = CM2R_SEL(keptn, j);
            const float tj_ = CM2R_SEL(thr, j);
            for (int ql_ = 0; ql_ < 16; ++ql_) {
                const int wq = j * 16 + ql_;
                const int n0 = __shfl(cj_, ql_), n1 = __shfl(cj_, ql_ + 16), n2 = __shfl(cj_, ql_ + 32), n3 = __shfl(cj_, ql_ + 48);
                const int nk = __shfl(kj_, ql_);
                float tq = __shfl(tj_, ql_);
                const int64_t qg = qbase + wq;
                int *dst = a.cand + (qg * a.S + split) * a.Kc;
                int nn = wave_merge_prune4(wkept + (size_t)wq * a.Kc, nk, wlbuf + (size_t)(wq * 4) * CLR, n0, n1, n2, n3, a.Kc, nullptr, dst, &tq, lane);
                for (int i = nn + lane; i < a.Kc; i += 64) dst[i] = -1;
                if (lane == 0) a.cthr[qg * a.S + split] = tq;
            }
        }
        if (adbg & 4) {
            const unsigned long long tw3_ = __builtin_amdgcn_s_memtime();
            c_procyc += tw1_ - tw0_; c_loopcyc += tw2_ - tw1_; c_fincyc += tw3_ - tw2_;
        }
#undef CM2R_MERGE_PASS
    }
#undef CM2R_DMA_TILE
#undef CM2R_DMA_PIECE
#undef CM2R_PT
    if (DBG && (adbg & 4) && lane == 0) {
        atomicAdd(&a.ctr[0], c_br);
        atomicAdd(&a.ctr[1], c_app);
        atomicAdd(&a.ctr[2], c_pr);
        atomicAdd(&a.ctr[3], c_prcyc);
        atomicAdd(&a.ctr[4], __builtin_amdgcn_s_memtime() - c_t0);
        atomicAdd(&a.ctr[5], 1ull);
        atomicAdd(&a.ctr[6], c_evcyc);
        atomicAdd(&a.ctr[7], c_barcyc);
        atomicAdd(&a.ctr[8], c_dmacyc);
        atomicAdd(&a.ctr[10], c_procyc);
        atomicAdd(&a.ctr[11], c_loopcyc);
        atomicAdd(&a.ctr[12], c_fincyc);
    }
}

}  // namespace cmk2
