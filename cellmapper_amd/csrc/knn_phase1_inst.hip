// Instantiations of the phase-1 kernel, split over translation units (CM_P1_GROUP) so they compile in parallel.
#include "cm_internal.h"
#include "knn_phase1.h"

#ifndef CM_P1_GROUP
#error "compile with -DCM_P1_GROUP=<0..3>"
#endif

namespace {
template <int KP, int NW>
cm_status launch(cm_handle_t h, dim3 grid, size_t lds, const float *Q, const float *R, int tps, int n_tiles, int Kc,
                 int C, int *cand, float *cthr) {
    auto kern = cmk::knn_cand_mfma<KP, NW>;
    CM_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds));
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, h->stream, Q, R, tps, n_tiles, Kc, C, cand, cthr);
    CM_HIP(h, hipGetLastError());
    return CM_OK;
}
template <int KP>
cm_status launch_nw(cm_handle_t h, int NW, dim3 grid, size_t lds, const float *Q, const float *R, int tps,
                    int n_tiles, int Kc, int C, int *cand, float *cthr) {
    switch (NW) {
    case 4: return launch<KP, 4>(h, grid, lds, Q, R, tps, n_tiles, Kc, C, cand, cthr);
    case 2: return launch<KP, 2>(h, grid, lds, Q, R, tps, n_tiles, Kc, C, cand, cthr);
    case 1: return launch<KP, 1>(h, grid, lds, Q, R, tps, n_tiles, Kc, C, cand, cthr);
    }
    CM_FAIL(h, CM_E_INTERNAL, "phase1: bad NW %d", NW);
}
}  // namespace

#define CM_P1_CASE(kp) \
    case kp: *st = launch_nw<kp>(h, NW, grid, lds, Q, R, tps, n_tiles, Kc, C, cand, cthr); return true;

// returns true when this group owns KP
#define CM_P1_FN_(g) cm_p1_launch_g##g
#define CM_P1_FN(g) CM_P1_FN_(g)
bool CM_P1_FN(CM_P1_GROUP)(cm_handle_t h, int KP, int NW, dim3 grid, size_t lds, const float *Q, const float *R,
                           int tps, int n_tiles, int Kc, int C, int *cand, float *cthr, cm_status *st) {
    switch (KP) {
#if CM_P1_GROUP == 0
        CM_P1_CASE(8) CM_P1_CASE(12) CM_P1_CASE(16) CM_P1_CASE(20) CM_P1_CASE(24) CM_P1_CASE(28) CM_P1_CASE(32)
#elif CM_P1_GROUP == 1
        CM_P1_CASE(36) CM_P1_CASE(40) CM_P1_CASE(44) CM_P1_CASE(48) CM_P1_CASE(52) CM_P1_CASE(56)
#elif CM_P1_GROUP == 2
        CM_P1_CASE(60) CM_P1_CASE(64) CM_P1_CASE(68) CM_P1_CASE(84) CM_P1_CASE(100)
#else
        CM_P1_CASE(132) CM_P1_CASE(164) CM_P1_CASE(196) CM_P1_CASE(260)
#endif
    }
    return false;
}

#if CM_P1_GROUP == 3
// wide rows (KP a multiple of 128 beyond 260): the K-chunked kernel
cm_status cm_p1_wide_launch(cm_handle_t h, int KP, int NW, dim3 grid, size_t lds, const float *Q, const float *R, int tps, int n_tiles,
                            int Kc, int C, int *cand, float *cthr) {
    if (KP % cmk::KCW != 0 || KP < cmk::KCW) CM_FAIL(h, CM_E_INTERNAL, "phase1 (wide): KP=%d is not a multiple of %d", KP, cmk::KCW);
#define CM_P1W(nw_)                                                                                                          \
    case nw_: {                                                                                                              \
        auto kern = cmk::knn_cand_mfma_wide<nw_>;                                                                            \
        CM_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(kern, grid, dim3(nw_ * 64), lds, h->stream, Q, R, KP, tps, n_tiles, Kc, C, cand, cthr);          \
        CM_HIP(h, hipGetLastError());                                                                                        \
        return CM_OK;                                                                                                        \
    }
    switch (NW) { CM_P1W(4) CM_P1W(2) CM_P1W(1) }
#undef CM_P1W
    CM_FAIL(h, CM_E_INTERNAL, "phase1 (wide): bad NW %d", NW);
}
#endif
