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
    // (CM_HALF = 0 .. 3: only four of the sixteen widths are instantiated in this object -- a ratio operator compiles for ~25 s per width)
    switch (D) {
#define CM_WD_CASE(d_) case d_: return launch_wide_direct<d_, OP>(h, grid, lds, a);
#if !defined(CM_HALF) || CM_HALF == 0
        CM_WD_CASE(68) CM_WD_CASE(72) CM_WD_CASE(76) CM_WD_CASE(80)
#endif
#if !defined(CM_HALF) || CM_HALF == 1
        CM_WD_CASE(84) CM_WD_CASE(88) CM_WD_CASE(92) CM_WD_CASE(96)
#endif
#if !defined(CM_HALF) || CM_HALF == 2
        CM_WD_CASE(100) CM_WD_CASE(104) CM_WD_CASE(108) CM_WD_CASE(112)
#endif
#if !defined(CM_HALF) || CM_HALF == 3
        CM_WD_CASE(116) CM_WD_CASE(120) CM_WD_CASE(124) CM_WD_CASE(128)
#endif
#undef CM_WD_CASE
    }
    CM_FAIL(h, CM_E_INTERNAL, "phase1/direct: no instantiation for D=%d", D);
}

// op: 0 = squared Euclidean, 1 = Manhattan, 2 = Chebyshev, 3 = Canberra, 4 = Bray-Curtis.  One object per operator (Makefile:
// -DCM_OP=0 .. 4) plus the dispatcher (no CM_OP): see knn_phase1_metric_inst.hip.
#define CM_CAT2(a, b) a##b
#define CM_CAT(a, b) CM_CAT2(a, b)
#if defined(CM_OP) && defined(CM_HALF)
cm_status CM_CAT(CM_CAT(cm_p1direct_wide_launch_op, CM_OP), CM_CAT(_h, CM_HALF))(cm_handle_t h, int D, int grid, size_t lds, const cmk2::P1Args &a) {
    return launch_wide_direct_d<CM_OP>(h, D, grid, lds, a);
}
#elif defined(CM_OP)
cm_status CM_CAT(cm_p1direct_wide_launch_op, CM_OP)(cm_handle_t h, int D, int grid, size_t lds, const cmk2::P1Args &a) {
    return launch_wide_direct_d<CM_OP>(h, D, grid, lds, a);
}
#else
#define CM_DECL4(op_) \
    cm_status cm_p1direct_wide_launch_op##op_##_h0(cm_handle_t h, int D, int grid, size_t lds, const cmk2::P1Args &a); \
    cm_status cm_p1direct_wide_launch_op##op_##_h1(cm_handle_t h, int D, int grid, size_t lds, const cmk2::P1Args &a); \
    cm_status cm_p1direct_wide_launch_op##op_##_h2(cm_handle_t h, int D, int grid, size_t lds, const cmk2::P1Args &a); \
    cm_status cm_p1direct_wide_launch_op##op_##_h3(cm_handle_t h, int D, int grid, size_t lds, const cmk2::P1Args &a); \
    cm_status cm_p1direct_wide_launch_op##op_(cm_handle_t h, int D, int grid, size_t lds, const cmk2::P1Args &a) { \
        if (D <= 80) return cm_p1direct_wide_launch_op##op_##_h0(h, D, grid, lds, a); \
        if (D <= 96) return cm_p1direct_wide_launch_op##op_##_h1(h, D, grid, lds, a); \
        if (D <= 112) return cm_p1direct_wide_launch_op##op_##_h2(h, D, grid, lds, a); \
        return cm_p1direct_wide_launch_op##op_##_h3(h, D, grid, lds, a); \
    }
CM_DECL4(3)
CM_DECL4(4)
#undef CM_DECL4
cm_status cm_p1direct_wide_launch_op0(cm_handle_t h, int D, int grid, size_t lds, const cmk2::P1Args &a);
cm_status cm_p1direct_wide_launch_op1(cm_handle_t h, int D, int grid, size_t lds, const cmk2::P1Args &a);
cm_status cm_p1direct_wide_launch_op2(cm_handle_t h, int D, int grid, size_t lds, const cmk2::P1Args &a);
cm_status cm_p1direct_wide_launch(cm_handle_t h, int D, int op, int grid, size_t lds, const cmk2::P1Args &a) {
    if (op == 0) return cm_p1direct_wide_launch_op0(h, D, grid, lds, a);
    if (op == 1) return cm_p1direct_wide_launch_op1(h, D, grid, lds, a);
    if (op == 2) return cm_p1direct_wide_launch_op2(h, D, grid, lds, a);
    if (op == 3) return cm_p1direct_wide_launch_op3(h, D, grid, lds, a);
    if (op == 4) return cm_p1direct_wide_launch_op4(h, D, grid, lds, a);
    CM_FAIL(h, CM_E_INTERNAL, "phase1/direct: unknown op %d", op);
}
#endif
