// Instantiations of the split-fp16 phase-1 kernel (knn_phase1_f16.h) in its threshold-seeding mode: 10-entry lists for the dense
// seeding scan of a single-pass pruned stage (knn_kernels.hip, knn_run_staged: every R2-th tile of a block's own runs), and the
// bucket-centred form of the 4-entry sample scan (an unpruned centred stage seeds from centred values of the row sample).
// Own translation unit: build time.
#include "cm_internal.h"
#include "knn_phase1_f16.h"

template <int KSTEPS, bool CEN, int L = 10>
static cm_status launch_seed10(cm_handle_t h, int grid, size_t lds, const cmk2::P1Args &a) {
    auto kern = cmk2::knn_cand_f16<KSTEPS, 1, CEN, L>;
    CM_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(cmk2::NT), lds, h->stream, a);
    CM_HIP(h, hipGetLastError());
    return CM_OK;
}

cm_status cm_p1f16_seed10_launch(cm_handle_t h, int KSTEPS, int grid, size_t lds, const cmk2::P1Args &a) {
    if (a.seed_len == 4 && a.side && a.nprod == 1) {   // centred sample scan
        if (KSTEPS == 2) return launch_seed10<2, true, 4>(h, grid, lds, a);
        if (KSTEPS == 4) return launch_seed10<4, true, 4>(h, grid, lds, a);
    }
    if (a.seed_len != 10 || a.nprod != 1) CM_FAIL(h, CM_E_INTERNAL, "phase1/seed: 10-entry lists of the hi*hi level (or the centred 4-entry scan) only");
    if (a.side) {   // bucket-centred stage: the seeds are values of the centred filter itself (same image, same side values as the pass)
        if (KSTEPS == 2) return launch_seed10<2, true>(h, grid, lds, a);
        if (KSTEPS == 4) return launch_seed10<4, true>(h, grid, lds, a);
    } else {
        if (KSTEPS == 2) return launch_seed10<2, false>(h, grid, lds, a);
        if (KSTEPS == 4) return launch_seed10<4, false>(h, grid, lds, a);
    }
    CM_FAIL(h, CM_E_INTERNAL, "phase1/seed: no instantiation for KSTEPS=%d", KSTEPS);
}
