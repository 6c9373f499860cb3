// Instantiations + launcher of the split-fp16 phase-1 kernel (knn_phase1_f16.h).
#include "cm_internal.h"
#include "knn_phase1_f16.h"

template <int KSTEPS, int NP, bool CEN = false, int SEEDL = 0>
static cm_status launch_f16(cm_handle_t h, int grid, size_t lds, const cmk2::P1Args &a) {
    auto kern = cmk2::knn_cand_f16<KSTEPS, NP, CEN, SEEDL>;
    CM_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(cmk2::NT), lds, h->stream, a);
    CM_HIP(h, hipGetLastError());
    return CM_OK;
}

cm_status cm_p1f16_launch(cm_handle_t h, int KSTEPS, int grid, size_t lds, const cmk2::P1Args &a) {
    if (a.side) {   // bucket-centred hi*hi level
        if (a.nprod == 1 && KSTEPS == 2) return launch_f16<2, 1, true>(h, grid, lds, a);
        if (a.nprod == 1 && KSTEPS == 4) return launch_f16<4, 1, true>(h, grid, lds, a);
        CM_FAIL(h, CM_E_INTERNAL, "phase1/f16: centred mode exists for the hi*hi level at K <= 64 only");
    }
    if (a.seed_len > 0) {   // threshold-seeding scan (register-resident lists of sub-tile minima)
        if (a.seed_len == 4 && a.nprod == 1 && KSTEPS == 2) return launch_f16<2, 1, false, 4>(h, grid, lds, a);
        if (a.seed_len == 4 && a.nprod == 1 && KSTEPS == 4) return launch_f16<4, 1, false, 4>(h, grid, lds, a);
        if (a.seed_len == 4 && a.nprod == 1 && KSTEPS == 8) return launch_f16<8, 1, false, 4>(h, grid, lds, a);
        CM_FAIL(h, CM_E_INTERNAL, "phase1/f16: the seeding mode exists for the hi*hi level with 4-entry lists");
    }
    switch (KSTEPS) {
#define CM_F16_CASE(ks_)                                                                \
    case ks_:                                                                           \
        return a.nprod >= 3 ? launch_f16<ks_, 3>(h, grid, lds, a)                       \
                            : (a.nprod == 2 ? launch_f16<ks_, 2>(h, grid, lds, a) : launch_f16<ks_, 1>(h, grid, lds, a));
        CM_F16_CASE(2) CM_F16_CASE(4)
#undef CM_F16_CASE
    case 8:   // K = 128: only the hi*hi level runs on this kernel (64 queries per wave); level 3 uses knn_cand_f16w
        // (production body unless a CELLMAP_DBG mode is on: diagnostics compiled out, query fragments in the accumulation registers)
#ifdef CM2_NO_PROD   // (A/B builds: round 2's body)
        if (a.nprod == 1) return launch_f16<8, 1>(h, grid, lds, a);
#endif
        if (a.nprod == 1 && a.dbg) return launch_f16<8, 1>(h, grid, lds, a);
        if (a.nprod == 1) {
            auto kern = cmk2::knn_cand_f16p<8>;
            CM_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kern, dim3(grid), dim3(cmk2::NT), lds, h->stream, a);
            CM_HIP(h, hipGetLastError());
            return CM_OK;
        }
        break;
    }
    CM_FAIL(h, CM_E_INTERNAL, "phase1/f16: no instantiation for KSTEPS=%d with %d products", KSTEPS, a.nprod);
}

cm_status cm_p1direct_launch(cm_handle_t h, int D, int grid, size_t lds, const cmk2::P1Args &a);

template <int D>
static cm_status launch_lowd(cm_handle_t h, int grid, size_t lds, const cmk2::P1Args &a) {
    auto kern = cmk2::knn_cand_lowd<D>;
    CM_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(cmk2::NT), lds, h->stream, a);
    CM_HIP(h, hipGetLastError());
    return CM_OK;
}

cm_status cm_p1lowd_launch(cm_handle_t h, int D, int grid, size_t lds, const cmk2::P1Args &a) {
    switch (D) {
#define CM_LD_CASE(d_) case d_: return launch_lowd<d_>(h, grid, lds, a);
        CM_LD_CASE(1) CM_LD_CASE(2) CM_LD_CASE(3) CM_LD_CASE(4) CM_LD_CASE(5) CM_LD_CASE(6) CM_LD_CASE(7) CM_LD_CASE(8)
        CM_LD_CASE(9) CM_LD_CASE(10) CM_LD_CASE(11) CM_LD_CASE(12) CM_LD_CASE(13) CM_LD_CASE(14) CM_LD_CASE(15)
#undef CM_LD_CASE
    }
    if (D >= 16) return cm_p1direct_launch(h, D, grid, lds, a);   // padded widths 16..64: knn_phase1_direct_inst.hip
    CM_FAIL(h, CM_E_INTERNAL, "phase1/lowd: no instantiation for D=%d", D);
}

template <bool LO3>
static cm_status launch_f16w(cm_handle_t h, int grid, size_t lds, const cmk2::P1Args &a) {
    auto kern = cmk2::knn_cand_f16w<cmk2::KSTEPS_W, LO3>;
    CM_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(cmk2::NTW), lds, h->stream, a);
    CM_HIP(h, hipGetLastError());
    return CM_OK;
}

cm_status cm_p1f16w_launch(cm_handle_t h, int grid, size_t lds, const cmk2::P1Args &a) {
    // (the hi*hi level of the K = 128 family runs on knn_cand_f16<8, 1>; this kernel is its three-product level)
    if (a.nprod < 3) CM_FAIL(h, CM_E_INTERNAL, "phase1/f16w: only the three-product level is instantiated");
    return launch_f16w<true>(h, grid, lds, a);
}

// code-object preload (cm_create): one empty launch makes the runtime load this translation unit's kernels onto the
// device ahead of the first real call (the load used to land inside the first call's stage times)
static __global__ void cm_touch_f16() {}
void cm_preload_f16(hipStream_t st) { hipLaunchKernelGGL(cm_touch_f16, dim3(1), dim3(64), 0, st); }
