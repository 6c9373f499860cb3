// Phase 1, hi*hi level, FOUR accumulators per wave (round 2): the full-set scan of the K <= 64 split-fp16 filter with
// 128 queries per wave instead of 64.
//
// Same contract, tile images, LDS ring, candidate logs, merges and outputs as knn_cand_f16<KSTEPS, 1> (knn_phase1_f16.h);
// what changes is the register tile: one reference fragment (A operand, read from LDS) now feeds FOUR 32x32 accumulators
// (128 queries), so a workgroup of 8 waves covers 1 024 queries per pass over the reference image:
//   * half the LDS fragment reads and half the LDS-DMA / L2 traffic per MFMA (the skeleton of the two-accumulator kernel is
//     power-limited: DESIGN.md section 6), half as many work items, barriers and final merges per query;
//   * the chain is accumulator-major -- acc j runs its KSTEPS dependent MFMAs back to back (the matrix core forwards the
//     accumulator: a single chain runs at the full rate, MI355X_MICROARCH.md) -- so an accumulator is complete a quarter of
//     the way through the sub-tile and no "previous sub-tile" copy of the accumulators is needed: the min tree of
//     accumulator (j+2)&3 is dealt out over the MFMA gaps of accumulator j, its threshold test follows them, and the
//     registers are overwritten two accumulators later.  Accumulators 2 and 3 of a sub-tile are therefore filtered during
//     the first half of the NEXT sub-tile (their row base travels along).
// Used for full-set scans without a work list (knn_run_staged decides); everything else stays on knn_cand_f16.
#pragma once
#include "knn_phase1_f16.h"

namespace cmk2 {

constexpr int QBQ = NW * 128;   // queries per workgroup pass of the four-accumulator kernel
// (The timing experiments of rounds 2-4 -- stages switched off one by one, the per-group event path, wave priority, the
//  in-place 16x16x32 probe -- live in scripts/ablate/knn_phase1_q4_ablate.h, which a -DCM_ABLATE build of
//  knn_phase1_q4_inst.hip includes instead of this file: scripts/build_q4_variant.sh.  This is the production kernel only.)

// One accumulator's event: the partial minima T0..T4 (registers 3g..3g+2) and the accumulator itself.
#define CM2Q_EVENT(ACC, T0, T1, T2, T3, T4, J, RB)                                                             \
    {                                                                                                          \
        unsigned long long te0_ = 0;                                                                           \
        if (adbg & 4) { ++c_br; te0_ = __builtin_amdgcn_s_memtime(); }                                        \
        const float tj_ = thr[J];                                                                              \
        int cj_ = cnt[J];                                                                                      \
        const unsigned loff_ = (unsigned)((((J) * 32 + c) * 2 + hh) * CL * 8);                                 \
        {   /* all six group tests issued back to back into scalar masks: ONE vector -> scalar hand-over instead of one  \
               per group, then scalar-only branching (A/B against the per-group path: profiles/r03_filter_ablation.md) */ \
            const unsigned long long g0_ = __ballot((T0) < tj_), g1_ = __ballot((T1) < tj_), g2_ = __ballot((T2) < tj_); \
            const unsigned long long g3_ = __ballot((T3) < tj_), g4_ = __ballot((T4) < tj_), g5_ = __ballot((ACC)[15] < tj_); \
            if (g0_) { CM2_APPEND_B((ACC)[0], 0, RB, loff_) CM2_APPEND_B((ACC)[1], 1, RB, loff_) CM2_APPEND_B((ACC)[2], 2, RB, loff_) } \
            if (g1_) { CM2_APPEND_B((ACC)[3], 3, RB, loff_) CM2_APPEND_B((ACC)[4], 4, RB, loff_) CM2_APPEND_B((ACC)[5], 5, RB, loff_) } \
            if (g2_) { CM2_APPEND_B((ACC)[6], 6, RB, loff_) CM2_APPEND_B((ACC)[7], 7, RB, loff_) CM2_APPEND_B((ACC)[8], 8, RB, loff_) } \
            if (g3_) { CM2_APPEND_B((ACC)[9], 9, RB, loff_) CM2_APPEND_B((ACC)[10], 10, RB, loff_) CM2_APPEND_B((ACC)[11], 11, RB, loff_) } \
            if (g4_) { CM2_APPEND_B((ACC)[12], 12, RB, loff_) CM2_APPEND_B((ACC)[13], 13, RB, loff_) CM2_APPEND_B((ACC)[14], 14, RB, loff_) } \
            if (g5_) CM2_APPEND_B((ACC)[15], 15, RB, loff_)                                                    \
        }                                                                                                      \
        cnt[J] = cj_;                                                                                          \
        if (adbg & 4) c_evcyc += __builtin_amdgcn_s_memtime() - te0_;                                         \
    }

// DBG = false is the production body: the CELLMAP_DBG modes (adbg: 1 never append, 2 skip the filter, 4 in-kernel
// counters) are compiled out -- they cost two vector and a dozen scalar instructions per accumulator and sub-tile, and
// the counters' SGPRs spilled (round 3).  The host launches DBG = true only when CELLMAP_DBG is set.
// CEN (round 5, late): the bucket-centred form -- the B operand holds q'' = q' - c, the accumulators start from the side values
// -2 c.y' of the wave's bucket (knn_cand_f16, CEN).  A workgroup covers two bucket-aligned 512-query blocks (waves 0..3 / 4..7),
// so a tile carries TWO side blocks of 1 KiB, copied by waves 0 and 4, and every wave reads its own half's.
template <int KSTEPS, bool DBG, bool CEN = false>
__global__ __launch_bounds__(NT, 2) void knn_cand_f16q(P1Args a) {
    const int adbg = DBG ? a.dbg : 0;
    constexpr int K = KSTEPS * 16;
    constexpr int ROWB = img_row_bytes(KSTEPS, true);
    constexpr int GTILEB = img_tile_bytes(KSTEPS, true);
    constexpr int TILEB = GTILEB + (CEN ? 2048 : 0);   // stride of the LDS ring (CEN: + the two side blocks)
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
    constexpr int NJ = 4;

    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char *ring = smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, hh = lane >> 5;
    uint2 *wkept = a.kept + (size_t(blockIdx.x) * NW + wave) * 128 * a.Kc;
    uint2 *wlbuf = a.lbuf + (size_t(blockIdx.x) * NW + wave) * 128 * 2 * CL;
    const unsigned ldsbase = (unsigned)(uintptr_t)(CM2_AS3 unsigned char *)ring;
    const unsigned dma_voff = wave * 1024 + lane * 16;

#define CM2Q_DMA_SIDE(t_, buf_)                                                                                 \
    __builtin_amdgcn_global_load_lds(                                                                           \
        (const CM2_AS1 void *)(reinterpret_cast<const char *>(side_row + (size_t)(t_) * BR) + lane * 16),       \
        (CM2_AS3 void *)(ring + (buf_) * TILEB + ROWS_B + (wave >> 2) * 1024), 16, 0, 0)
#define CM2Q_DMA_TILE(t_, buf_)                                                                                 \
    {                                                                                                           \
        const char *gsrc_ = a.R + (size_t)(t_) * GTILEB;                                                        \
        unsigned char *ldst_ = ring + (buf_) * TILEB + wave * 1024;                                             \
        _Pragma("unroll") for (int q_ = 0; q_ < NPIECE; ++q_)                                                   \
            __builtin_amdgcn_global_load_lds((const CM2_AS1 void *)(gsrc_ + q_ * (NW * 1024) + dma_voff),       \
                                             (CM2_AS3 void *)(ldst_ + q_ * (NW * 1024)), 16, 0, 0);             \
        if (CEN && sidew) CM2Q_DMA_SIDE(t_, buf_);                                                              \
    }
#define CM2Q_DMA_PIECE(t_, buf_, q_)                                                                            \
    __builtin_amdgcn_global_load_lds(                                                                           \
        (const CM2_AS1 void *)(a.R + (size_t)(t_) * GTILEB + (q_) * (NW * 1024) + dma_voff),                    \
        (CM2_AS3 void *)(ring + (buf_) * TILEB + wave * 1024 + (q_) * (NW * 1024)), 16, 0, 0)
#define CM2Q_PT(t_) (a.perm_mul > 1u ? (int)(((unsigned long long)(unsigned)(t_) * a.perm_mul) % (unsigned)a.n_tiles) : (int)(t_))

    unsigned long long c_br = 0, c_app = 0, c_pr = 0, c_prcyc = 0, c_evcyc = 0, c_barcyc = 0, c_dmacyc = 0;
    unsigned long long c_procyc = 0, c_loopcyc = 0, c_fincyc = 0;
    const unsigned long long c_t0 = DBG ? __builtin_amdgcn_s_memtime() : 0ull;
    const int n_work = a.nqb * a.S;
    for (int work = blockIdx.x; work < n_work; work += (int)gridDim.x) {
        const int qb = work / a.S, split = work - qb * a.S;
        const int t0 = split * a.tiles_per_split;
        const int t1 = (t0 + a.tiles_per_split < a.n_tiles) ? t0 + a.tiles_per_split : a.n_tiles;
        const int64_t qbase = (int64_t)qb * QBQ + wave * 128;
        const unsigned long long tw0_ = (adbg & 4) ? __builtin_amdgcn_s_memtime() : 0ull;
        // CEN: side values of this wave's bucket (one float per row of the padded reference image); waves 0 and 4 copy them
        const float *side_row = nullptr;
        if constexpr (CEN) side_row = a.side + (size_t)a.blk_bucket[__builtin_amdgcn_readfirstlane(qb * 2 + (wave >> 2))] * (size_t)a.side_stride;
        const bool sidew = CEN && __builtin_amdgcn_readfirstlane(wave & 3) == 0;
        (void)side_row; (void)sidew;

        // ---- query fragments (B operand) of the wave's 128 queries: hi halves only (hi*hi level)
        half8 qh[NJ][KSTEPS];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const _Float16 *qrow = a.Q + (qbase + j * 32 + c) * (2 * K);
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) qh[j][s] = *reinterpret_cast<const half8 *>(qrow + (2 * s + hh) * 8);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) asm volatile("" ::"v"(qh[j][s]));
        const float thr_init = (adbg & 1) ? -INFINITY : INFINITY;
        float thr[NJ] = {thr_init, thr_init, thr_init, thr_init};
        if (a.thr0) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) thr[j] = a.thr0[qbase + j * 32 + c];
            asm volatile("" : "+v"(thr[0]), "+v"(thr[1]), "+v"(thr[2]), "+v"(thr[3]));
        }
        int cnt[NJ] = {0, 0, 0, 0};
        int keptn[NJ] = {0, 0, 0, 0};

        // merge the logs of every query whose lanes hold more than LIM_ entries (runtime loop over the accumulators)
#define CM2Q_SEL(ARR, J) ((J) == 0 ? ARR[0] : ((J) == 1 ? ARR[1] : ((J) == 2 ? ARR[2] : ARR[3])))
#define CM2Q_PUT(ARR, J, V) { if ((J) == 0) ARR[0] = (V); else if ((J) == 1) ARR[1] = (V); else if ((J) == 2) ARR[2] = (V); else ARR[3] = (V); }
#define CM2Q_MERGE_PASS(LIM_)                                                                                  \
    {                                                                                                          \
        unsigned long long tp0_ = (adbg & 4) ? __builtin_amdgcn_s_memtime() : 0ull;                           \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                       \
        _Pragma("unroll 1") for (int j = 0; j < NJ; ++j) {                                                     \
            int cj_ = CM2Q_SEL(cnt, j), kj_ = CM2Q_SEL(keptn, j);                                              \
            float tj_ = CM2Q_SEL(thr, j);                                                                      \
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
                if (adbg & 4) { ++c_pr; c_app += n0 + n1; }                                                   \
            }                                                                                                  \
            CM2Q_PUT(cnt, j, cj_) CM2Q_PUT(keptn, j, kj_) CM2Q_PUT(thr, j, tj_)                                \
        }                                                                                                      \
        if (adbg & 4) c_prcyc += __builtin_amdgcn_s_memtime() - tp0_;                                         \
    }

        __builtin_amdgcn_s_barrier();  // previous work item: every wave is done reading the ring
#pragma unroll
        for (int u = 0; u < AHEAD; ++u)
            if (t0 + u < t1) CM2Q_DMA_TILE(CM2Q_PT(t0 + u), (t0 + u) % RING);

        f32x16 acc[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = INFINITY;   // "nothing pending"
        int rb_cur = 0, rb_prev = 0;   // first reference row (+4*hh) of the current / previous sub-tile

        const unsigned arow = c * ROWB + ((hh ^ swz_b(c, ROWB)) * 16);
        const int sched_lim = a.no_sched ? 8 : 0;
        int next_refresh = a.no_sched ? (a.no_sched == 2 ? t0 : 0x7fffffff) : t0 + 1;
        half8 ah[KSTEPS], al[KSTEPS];   // the two fragment sets (current / next sub-tile, alternating)
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) { ah[s] = half8{0, 0, 0, 0, 0, 0, 0, 0}; al[s] = ah[s]; }
        // CEN: the lane's 16 side values of a sub-tile (rows 4*hh + {0..3} + 8*i), two sets alternating like the fragments
        f32x4 ya[4], yb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { ya[i] = f32x4{0.f, 0.f, 0.f, 0.f}; yb[i] = ya[i]; }
        const unsigned yrow = ROWS_B + (wave >> 2) * 1024 + 16 * hh;
        (void)yrow;
        if (t0 < t1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const unsigned tile0 = ldsbase + (t0 % RING) * TILEB;
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) CM2_DSR128X(ah[s], arow, 32 * s, tile0, 0);
            if constexpr (CEN) {
#pragma unroll
                for (int i = 0; i < 4; ++i) CM2_DSR128X(ya[i], yrow, 0, tile0, 32 * i);
            }
        }
        int ptile = CM2Q_PT(t0), ptile_a = CM2Q_PT(t0 + AHEAD);
        const int pstep = a.perm_mul > 1u ? (int)(a.perm_mul % (unsigned)a.n_tiles) : 1;
        const unsigned long long tw1_ = (adbg & 4) ? __builtin_amdgcn_s_memtime() : 0ull;

        // min tree of accumulator P in 8 steps: T0..T4 = minima of registers 3g..3g+2, U0 / U1 / TOT above them
#define CM2Q_TREE_STEP(I, P, T, U, TOT)                                                                        \
    {                                                                                                          \
        if ((I) < 5) T[(I) < 5 ? (I) : 0] = vmin3(P[3 * ((I) < 5 ? (I) : 0)], P[3 * ((I) < 5 ? (I) : 0) + 1], P[3 * ((I) < 5 ? (I) : 0) + 2]); \
        else if ((I) == 5) U[0] = vmin3(T[0], T[1], T[2]);                                                     \
        else if ((I) == 6) U[1] = vmin3(T[3], T[4], P[15]);                                                    \
        else TOT = vmin3(U[0], U[1], U[0]);                                                                    \
    }
        // One sub-tile: CUR holds its fragments, NXT receives those of the next sub-tile (base address NBASE).
#define CM2Q_SUBTILE(CUR, NXT, YC, YN, ST)                                                                     \
    {                                                                                                          \
        const int nst_ = ((ST) + 1 < NST) ? (ST) + 1 : 0;                                                      \
        const unsigned nbase_ = (((ST) + 1 < NST) ? tile : tile_n) + nst_ * 32 * ROWB;                         \
        const unsigned ybase_ = (((ST) + 1 < NST) ? tile : tile_n);                                            \
        rb_prev = rb_cur;                                                                                      \
        rb_cur = ptile * BR + (ST) * 32 + 4 * hh;                                                              \
        wait_lds<0>(CUR);                                                                                      \
        f32x16 yc_;                                                                                            \
        if constexpr (CEN) {                                                                                   \
            asm volatile("" : "+v"(YC[0]), "+v"(YC[1]), "+v"(YC[2]), "+v"(YC[3]));   /* (ordered after the wait) */ \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) { yc_[e] = YC[0][e]; yc_[4 + e] = YC[1][e]; yc_[8 + e] = YC[2][e]; yc_[12 + e] = YC[3][e]; } \
        }                                                                                                      \
        _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                       \
            constexpr int dummy_ = 0; (void)dummy_;                                                            \
            const int jp = (j + 2) & 3;             /* accumulator filtered in the shadow of accumulator j */ \
            float T_[5], U_[2], tot_ = INFINITY;                                                               \
            const f32x16 zero_ = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; \
            _Pragma("unroll") for (int s = 0; s < KSTEPS; ++s) {                                               \
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(CUR[s], qh[j][s], s == 0 ? (CEN ? yc_ : zero_) : acc[j], 0, 0, 0); \
                __builtin_amdgcn_sched_barrier(0);                                                             \
                if (j == 0) CM2_DSR128X(NXT[s], arow, 32 * s, nbase_, 0);                                      \
                if (CEN && j == 2) { _Pragma("unroll") for (int q_ = 0; q_ < 4 / KSTEPS; ++q_) CM2_DSR128X(YN[s * (4 / KSTEPS) + q_], yrow, 0, ybase_, nst_ * 128 + 32 * (s * (4 / KSTEPS) + q_)); } \
                _Pragma("unroll") for (int q_ = 0; q_ < 8 / KSTEPS; ++q_)                                      \
                    CM2Q_TREE_STEP(s * (8 / KSTEPS) + q_, acc[jp], T_, U_, tot_)                               \
                if (j == 1 && s == 0 && (ST) < NPIECE && t + AHEAD < t1) CM2Q_DMA_PIECE(ptile_a, (t + AHEAD) % RING, (ST)); \
                if (CEN && j == 1 && s == 0 && (ST) == 0 && sidew && t + AHEAD < t1) CM2Q_DMA_SIDE(ptile_a, (t + AHEAD) % RING); \
                __builtin_amdgcn_sched_barrier(0);                                                             \
            }                                                                                                  \
            if (!(adbg & 2)) {                                                                                 \
                const bool pass_ = tot_ < thr[jp];                                                             \
                if (__ballot(pass_) != 0ull) {                                                                 \
                    const int rbj_ = jp >= 2 ? rb_prev : rb_cur;                                               \
                    CM2Q_EVENT(acc[jp], T_[0], T_[1], T_[2], T_[3], T_[4], jp, rbj_)                           \
                }                                                                                              \
            }                                                                                                  \
        }                                                                                                      \
    }

        for (int t = t0; t < t1; ++t) {
            unsigned long long tb0_ = (adbg & 4) ? __builtin_amdgcn_s_memtime() : 0ull;
            if (t > t0 && (BT == 1 || (t - t0) % BT == 0)) {
                if (t + AHEAD - 1 >= t1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (CEN && sidew) asm volatile("s_waitcnt vmcnt(%0)" ::"i"((AHEAD - 1 - BT) * (NPIECE + 1)) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"i"((AHEAD - 1 - BT) * NPIECE) : "memory");
                if (adbg & 4) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); c_dmacyc += t_ - tb0_; tb0_ = t_; }
                __builtin_amdgcn_s_barrier();
                if (adbg & 4) c_barcyc += __builtin_amdgcn_s_memtime() - tb0_;
            }
            const unsigned tile = ldsbase + (t % RING) * TILEB;
            const unsigned tile_n = ldsbase + ((t + 1) % RING) * TILEB;
#pragma unroll
            for (int sp = 0; sp < NST; sp += 2) {
                CM2Q_SUBTILE(ah, al, ya, yb, sp)
                CM2Q_SUBTILE(al, ah, yb, ya, sp + 1)
            }
            if (a.no_sched) {
                // aligned merges of the seeded scan (see knn_cand_f16): a wave with a long log announces it, everybody merges
                // its longer logs at the end of the following tile
                const unsigned fl_ = ldsbase + RING * TILEB + 16;
                if (__ballot(cnt[0] > 48 || cnt[1] > 48 || cnt[2] > 48 || cnt[3] > 48) != 0ull) {
                    const int tn_ = t + 1;
                    asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(fl_ + 4 * (tn_ & 1)), "v"(tn_) : "memory");
                }
                int f_;
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(f_) : "v"(fl_ + 4 * (t & 1)) : "memory");
                if (__builtin_amdgcn_readfirstlane(f_) == t) next_refresh = t;
            }
            if (!(adbg & 3)) {   // end of tile: scheduled refresh, or logs that could overflow during the next tile
                constexpr int FULL_ = CL - 16 * NST;
                const bool sched_ = t == next_refresh;
                if (sched_ || __ballot(cnt[0] > FULL_ || cnt[1] > FULL_ || cnt[2] > FULL_ || cnt[3] > FULL_) != 0ull) {
                    if (sched_) next_refresh = t0 + 2 * (t - t0 + 1) - 1;
                    const int lim_ = sched_ ? sched_lim : FULL_;
                    CM2Q_MERGE_PASS(lim_)
                }
            }
            if (a.no_sched) next_refresh = 0x7fffffff;
            ptile += pstep; if (ptile >= a.n_tiles) ptile -= a.n_tiles;
            ptile_a += pstep; if (ptile_a >= a.n_tiles) ptile_a -= a.n_tiles;
        }
        // drain: accumulators 2 and 3 of the last sub-tile have not been filtered yet
        if (!(adbg & 2)) {
#pragma unroll
            for (int jp = 2; jp < 4; ++jp) {
                float T_[5], U_[2], tot_ = INFINITY;
#pragma unroll
                for (int i = 0; i < 8; ++i) CM2Q_TREE_STEP(i, acc[jp], T_, U_, tot_)
                if (__ballot(tot_ < thr[jp]) != 0ull) CM2Q_EVENT(acc[jp], T_[0], T_[1], T_[2], T_[3], T_[4], jp, rb_cur)
            }
        }
        if (adbg & 2) asm volatile("" ::"v"(acc[0][0]), "v"(acc[1][0]), "v"(acc[2][15]), "v"(acc[3][15]));
#undef CM2Q_SUBTILE
#undef CM2Q_TREE_STEP

        const unsigned long long tw2_ = (adbg & 4) ? __builtin_amdgcn_s_memtime() : 0ull;
        // final merge: K' candidate rows (-1 padded) and the threshold of every query of the wave
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll 1
        for (int j = 0; j < NJ; ++j) {
            const int cj_ = CM2Q_SEL(cnt, j), kj_ = CM2Q_SEL(keptn, j);
            const float tj_ = CM2Q_SEL(thr, j);
            for (int ql_ = 0; ql_ < 32; ++ql_) {
                const int wq = j * 32 + ql_;
                const int n0 = __shfl(cj_, ql_), n1 = __shfl(cj_, ql_ + 32), nk = __shfl(kj_, ql_);
                float tq = __shfl(tj_, ql_);
                const int64_t qg = qbase + wq;
                int *dst = a.cand + (qg * a.S + split) * a.Kc;
                int nn = wave_merge_prune(nullptr, wkept + (size_t)wq * a.Kc, nk, wlbuf + (size_t)(wq * 2) * CL, n0,
                                          wlbuf + (size_t)(wq * 2 + 1) * CL, n1, a.Kc, nullptr, dst, &tq, lane);
                for (int i = nn + lane; i < a.Kc; i += 64) dst[i] = -1;
                if (lane == 0) a.cthr[qg * a.S + split] = tq;
            }
        }
        if (adbg & 4) {
            const unsigned long long tw3_ = __builtin_amdgcn_s_memtime();
            c_procyc += tw1_ - tw0_; c_loopcyc += tw2_ - tw1_; c_fincyc += tw3_ - tw2_;
        }
#undef CM2Q_MERGE_PASS
    }
#undef CM2Q_DMA_TILE
#undef CM2Q_DMA_PIECE
#undef CM2Q_DMA_SIDE
#undef CM2Q_PT
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
