// Instantiations + launcher of the 16x16x32 hi*hi filter (knn_phase1_r16.h): its own translation unit.
#include "cm_internal.h"
#include "knn_phase1_r16.h"

template <bool DBG>
static cm_status launch_r16(cm_handle_t h, int grid, size_t lds, const cmk2::P1Args &a) {
    auto kern = cmk2::knn_cand_f16r<DBG>;
    CM_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(cmk2::NT), lds, h->stream, a);
    CM_HIP(h, hipGetLastError());
    return CM_OK;
}

// hi*hi level, K = 64, full-set scans without a work list.  a.dbg != 0 (CELLMAP_DBG) selects the diagnostics body.
cm_status cm_p1f16r_launch(cm_handle_t h, int KSTEPS, int grid, size_t lds, const cmk2::P1Args &a) {
    if (KSTEPS != 4 || a.nprod != 1 || a.side || a.wlist || a.seed_len)
        CM_FAIL(h, CM_E_INTERNAL, "phase1/f16r: instantiated for the hi*hi level at K = 64 without work lists");
    return a.dbg ? launch_r16<true>(h, grid, lds, a) : launch_r16<false>(h, grid, lds, a);
}

// code-object preload (cm_create)
static __global__ void cm_touch_r16() {}
void cm_preload_r16(hipStream_t st) { hipLaunchKernelGGL(cm_touch_r16, dim3(1), dim3(64), 0, st); }
