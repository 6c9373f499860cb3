// Cycles per MFMA instruction on one SIMD (one wave per SIMD, 4 independent accumulators), gfx950:
//   v_mfma_f32_32x32x16_f16 vs the legacy v_mfma_f32_32x32x8_f16.   Build: hipcc --offload-arch=gfx950 -O3 -o build_ab/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(256) void probe(float *out, unsigned long long *cyc, int iters) {
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    half8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (threadIdx.x + e)); b[e] = (_Float16)(0.002f * (threadIdx.x * 3 + e)); }
    half4 a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (MODE == 0) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
            else if (MODE == 1) acc[j] = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, acc[j], 0, 0, 0);
            else {   // 3 x16 + 1 x8 per accumulator (the K = 56 chain)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, acc[j], 0, 0, 0);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[MODE] = t1 - t0;
}
int main() {
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 64); hipMemset(cyc, 0, 64);
    const int iters = 20000;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(probe<0>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
        hipLaunchKernelGGL(probe<1>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
        hipLaunchKernelGGL(probe<2>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
    }
    unsigned long long h[8];
    hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf("x16: %.2f cycles/MFMA   x8: %.2f cycles/MFMA   (3 x16 + 1 x8): %.2f cycles per 4-MFMA group (4 x16 = %.2f)\n",
           h[0] / (double)(iters * 4), h[1] / (double)(iters * 4), h[2] / (double)(iters * 4), 4.0 * h[0] / (double)(iters * 4));
    return 0;
}
