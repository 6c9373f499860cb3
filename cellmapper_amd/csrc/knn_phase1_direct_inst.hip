// Instantiations of the direct-difference VALU filter (knn_cand_lowd) for padded row widths 16..64: the high-precision
// level of the staged search for 16 <= d <= 64 (knn_kernels.hip, knn_run_staged).  Own translation unit: build time.
#include "cm_internal.h"
#include "knn_phase1_f16.h"

template <int D>
static cm_status launch_direct(cm_handle_t h, int grid, size_t lds, const cmk2::P1Args &a) {
    auto kern = cmk2::knn_cand_lowd<D>;
    CM_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(cmk2::NT), lds, h->stream, a);
    CM_HIP(h, hipGetLastError());
    return CM_OK;
}

cm_status cm_p1direct_launch(cm_handle_t h, int D, int grid, size_t lds, const cmk2::P1Args &a) {
    switch (D) {
#define CM_DD_CASE(d_) case d_: return launch_direct<d_>(h, grid, lds, a);
        CM_DD_CASE(16) CM_DD_CASE(20) CM_DD_CASE(24) CM_DD_CASE(28) CM_DD_CASE(32) CM_DD_CASE(36) CM_DD_CASE(40)
        CM_DD_CASE(44) CM_DD_CASE(48) CM_DD_CASE(52) CM_DD_CASE(56) CM_DD_CASE(60) CM_DD_CASE(64)
#undef CM_DD_CASE
    }
    CM_FAIL(h, CM_E_INTERNAL, "phase1/direct: no instantiation for D=%d", D);
}
