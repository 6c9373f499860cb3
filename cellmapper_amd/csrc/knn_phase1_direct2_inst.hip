// Instantiations of the direct-difference VALU filter (knn_cand_lowd) for padded row widths 68..128 (one query per lane): the
// direct level of the K = 128 family (knn_kernels.hip, knn_run_staged) and the manhattan / chebyshev filter at those widths.
// Own translation unit: build time.
#include "cm_internal.h"
#include "knn_phase1_f16.h"

template <int D, int OP>
static cm_status launch_wide_direct(cm_handle_t h, int grid, size_t lds, const cmk2::P1Args &a) {
    auto kern = cmk2::knn_cand_lowd<D, OP>;
    CM_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(cmk2::NT), lds, h->stream, a);
    CM_HIP(h, hipGetLastError());
    return CM_OK;
}

template <int OP>
static cm_status launch_wide_direct_d(cm_handle_t h, int D, int grid, size_t lds, const cmk2::P1Args &a) {
    switch (D) {
#define CM_WD_CASE(d_) case d_: return launch_wide_direct<d_, OP>(h, grid, lds, a);
        CM_WD_CASE(68) CM_WD_CASE(72) CM_WD_CASE(76) CM_WD_CASE(80) CM_WD_CASE(84) CM_WD_CASE(88) CM_WD_CASE(92) CM_WD_CASE(96)
        CM_WD_CASE(100) CM_WD_CASE(104) CM_WD_CASE(108) CM_WD_CASE(112) CM_WD_CASE(116) CM_WD_CASE(120) CM_WD_CASE(124) CM_WD_CASE(128)
#undef CM_WD_CASE
    }
    CM_FAIL(h, CM_E_INTERNAL, "phase1/direct: no instantiation for D=%d", D);
}

// op: 0 = squared Euclidean, 1 = Manhattan, 2 = Chebyshev, 3 = Canberra, 4 = Bray-Curtis.  Compiled twice (Makefile:
// -DCM_METRIC_GROUP=0 -> ops 0, 1, 2; =1 -> ops 3, 4).
#ifndef CM_METRIC_GROUP
#define CM_METRIC_GROUP 0
#endif
#if CM_METRIC_GROUP == 0
cm_status cm_p1direct_wide_launch_g1(cm_handle_t h, int D, int op, int grid, size_t lds, const cmk2::P1Args &a);
cm_status cm_p1direct_wide_launch(cm_handle_t h, int D, int op, int grid, size_t lds, const cmk2::P1Args &a) {
    if (op == 0) return launch_wide_direct_d<0>(h, D, grid, lds, a);
    if (op == 1) return launch_wide_direct_d<1>(h, D, grid, lds, a);
    if (op == 2) return launch_wide_direct_d<2>(h, D, grid, lds, a);
    if (op == 3 || op == 4) return cm_p1direct_wide_launch_g1(h, D, op, grid, lds, a);
    CM_FAIL(h, CM_E_INTERNAL, "phase1/direct: unknown op %d", op);
}
#else
cm_status cm_p1direct_wide_launch_g1(cm_handle_t h, int D, int op, int grid, size_t lds, const cmk2::P1Args &a) {
    if (op == 3) return launch_wide_direct_d<3>(h, D, grid, lds, a);
    if (op == 4) return launch_wide_direct_d<4>(h, D, grid, lds, a);
    CM_FAIL(h, CM_E_INTERNAL, "phase1/direct: unknown op %d", op);
}
#endif
