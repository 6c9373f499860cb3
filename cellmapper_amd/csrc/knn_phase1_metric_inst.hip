// Instantiations of the VALU direct-difference filter (knn_cand_lowd) for the Manhattan and Chebyshev metrics: the only
// filter level of those metrics (knn_kernels.hip, knn_run_staged), row widths 1..15 and the padded widths 16..64.
// Own translation unit: build time.
#include "cm_internal.h"
#include "knn_phase1_f16.h"

template <int D, int OP>
static cm_status launch_metric(cm_handle_t h, int grid, size_t lds, const cmk2::P1Args &a) {
    auto kern = cmk2::knn_cand_lowd<D, OP>;
    CM_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(cmk2::NT), lds, h->stream, a);
    CM_HIP(h, hipGetLastError());
    return CM_OK;
}

template <int OP>
static cm_status launch_metric_d(cm_handle_t h, int D, int grid, size_t lds, const cmk2::P1Args &a) {
    // (CM_HALF = 0 / 1: only the widths up to 28 / beyond are instantiated in this object)
    switch (D) {
#define CM_MD_CASE(d_) case d_: return launch_metric<d_, OP>(h, grid, lds, a);
#if !defined(CM_HALF) || CM_HALF == 0
        CM_MD_CASE(1) CM_MD_CASE(2) CM_MD_CASE(3) CM_MD_CASE(4) CM_MD_CASE(5) CM_MD_CASE(6) CM_MD_CASE(7) CM_MD_CASE(8)
        CM_MD_CASE(9) CM_MD_CASE(10) CM_MD_CASE(11) CM_MD_CASE(12) CM_MD_CASE(13) CM_MD_CASE(14) CM_MD_CASE(15)
        CM_MD_CASE(16) CM_MD_CASE(20) CM_MD_CASE(24) CM_MD_CASE(28)
#endif
#if !defined(CM_HALF) || CM_HALF == 1
        CM_MD_CASE(32) CM_MD_CASE(36) CM_MD_CASE(40)
        CM_MD_CASE(44) CM_MD_CASE(48) CM_MD_CASE(52) CM_MD_CASE(56) CM_MD_CASE(60) CM_MD_CASE(64)
#endif
#undef CM_MD_CASE
    }
    CM_FAIL(h, CM_E_INTERNAL, "phase1/metric: no instantiation for D=%d", D);
}

// op: 1 = Manhattan, 2 = Chebyshev, 3 = Canberra, 4 = Bray-Curtis.  One object per operator (Makefile: -DCM_OP=1 .. 4, round 5: the
// 4 x 28 instantiations were the long pole of a clean build) plus the dispatcher (no CM_OP).
#define CM_CAT2(a, b) a##b
#define CM_CAT(a, b) CM_CAT2(a, b)
#if defined(CM_OP) && defined(CM_HALF)
cm_status CM_CAT(CM_CAT(cm_p1lowd_metric_launch_op, CM_OP), CM_CAT(_h, CM_HALF))(cm_handle_t h, int D, int grid, size_t lds, const cmk2::P1Args &a) {
    return launch_metric_d<CM_OP>(h, D, grid, lds, a);
}
#elif defined(CM_OP)
cm_status CM_CAT(cm_p1lowd_metric_launch_op, CM_OP)(cm_handle_t h, int D, int grid, size_t lds, const cmk2::P1Args &a) {
    return launch_metric_d<CM_OP>(h, D, grid, lds, a);
}
#else
cm_status cm_p1lowd_metric_launch_op3_h0(cm_handle_t h, int D, int grid, size_t lds, const cmk2::P1Args &a);
cm_status cm_p1lowd_metric_launch_op3_h1(cm_handle_t h, int D, int grid, size_t lds, const cmk2::P1Args &a);
cm_status cm_p1lowd_metric_launch_op4_h0(cm_handle_t h, int D, int grid, size_t lds, const cmk2::P1Args &a);
cm_status cm_p1lowd_metric_launch_op4_h1(cm_handle_t h, int D, int grid, size_t lds, const cmk2::P1Args &a);
cm_status cm_p1lowd_metric_launch_op3(cm_handle_t h, int D, int grid, size_t lds, const cmk2::P1Args &a) {
    return D <= 28 ? cm_p1lowd_metric_launch_op3_h0(h, D, grid, lds, a) : cm_p1lowd_metric_launch_op3_h1(h, D, grid, lds, a);
}
cm_status cm_p1lowd_metric_launch_op4(cm_handle_t h, int D, int grid, size_t lds, const cmk2::P1Args &a) {
    return D <= 28 ? cm_p1lowd_metric_launch_op4_h0(h, D, grid, lds, a) : cm_p1lowd_metric_launch_op4_h1(h, D, grid, lds, a);
}
cm_status cm_p1lowd_metric_launch_op1(cm_handle_t h, int D, int grid, size_t lds, const cmk2::P1Args &a);
cm_status cm_p1lowd_metric_launch_op2(cm_handle_t h, int D, int grid, size_t lds, const cmk2::P1Args &a);
cm_status cm_p1lowd_metric_launch(cm_handle_t h, int D, int op, int grid, size_t lds, const cmk2::P1Args &a) {
    if (op == 1) return cm_p1lowd_metric_launch_op1(h, D, grid, lds, a);
    if (op == 2) return cm_p1lowd_metric_launch_op2(h, D, grid, lds, a);
    if (op == 3) return cm_p1lowd_metric_launch_op3(h, D, grid, lds, a);
    if (op == 4) return cm_p1lowd_metric_launch_op4(h, D, grid, lds, a);
    CM_FAIL(h, CM_E_INTERNAL, "phase1/metric: unknown op %d", op);
}
#endif
