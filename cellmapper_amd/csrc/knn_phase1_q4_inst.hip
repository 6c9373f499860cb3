// Instantiations + launcher of the four-accumulator hi*hi filter (knn_phase1_q4.h): its own translation unit, so that work
// on this kernel does not recompile the rest of the split-fp16 family.
#include "cm_internal.h"
#ifdef CM_ABLATE   // timing experiments (scripts/build_q4_variant.sh): the copy of the kernel that carries the CM2Q_EXP / _EVMASK / _PRIO switches
#include "../../scripts/ablate/knn_phase1_q4_ablate.h"
#else
#include "knn_phase1_q4.h"
#endif

template <int KSTEPS, bool DBG, bool CEN = false>
static cm_status launch_q4(cm_handle_t h, int grid, size_t lds, const cmk2::P1Args &a) {
    auto kern = cmk2::knn_cand_f16q<KSTEPS, DBG, CEN>;
    CM_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(cmk2::NT), lds, h->stream, a);
    CM_HIP(h, hipGetLastError());
    return CM_OK;
}

// hi*hi level, K = 64 or 32, full-set scans without a work list.  a.dbg != 0 (CELLMAP_DBG) selects the diagnostics body.
cm_status cm_p1f16q_launch(cm_handle_t h, int KSTEPS, int grid, size_t lds, const cmk2::P1Args &a) {
    if ((KSTEPS != 4 && KSTEPS != 2) || a.nprod != 1 || a.wlist || a.seed_len)
        CM_FAIL(h, CM_E_INTERNAL, "phase1/f16q: instantiated for the hi*hi level at K <= 64 without work lists");
#ifndef CM_ABLATE
    if (a.side) {   // bucket-centred form (production body only)
        if (a.dbg) CM_FAIL(h, CM_E_INTERNAL, "phase1/f16q: the centred form has no diagnostics body");
        return KSTEPS == 4 ? launch_q4<4, false, true>(h, grid, lds, a) : launch_q4<2, false, true>(h, grid, lds, a);
    }
#else
    if (a.side) CM_FAIL(h, CM_E_INTERNAL, "phase1/f16q: the ablation copy has no centred form");
#endif
    if (a.dbg) return KSTEPS == 4 ? launch_q4<4, true>(h, grid, lds, a) : launch_q4<2, true>(h, grid, lds, a);
    return KSTEPS == 4 ? launch_q4<4, false>(h, grid, lds, a) : launch_q4<2, false>(h, grid, lds, a);
}

// code-object preload (cm_create): one empty launch makes the runtime load this translation unit's kernels onto the
// device ahead of the first real call (the load used to land inside the first call's stage times)
static __global__ void cm_touch_q4() {}
void cm_preload_q4(hipStream_t st) { hipLaunchKernelGGL(cm_touch_q4, dim3(1), dim3(64), 0, st); }
