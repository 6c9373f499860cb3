// What does the device allocator cost on this box?  hipMalloc / first touch / hipFree wall times by size, three rounds each,
// plus hipHostMalloc -- to attribute a slow FIRST call of a fresh handle (round-4 review: 693 ms against 244 ms on the driver's
// box, 240 against 239 on the builder's).  Run it alone and next to a monitoring loop (rocm-smi / amd-smi polling takes
// driver locks the allocator's ioctls may wait for).
// Build: hipcc -O2 --offload-arch=gfx950 -o build_ab/alloc_probe scripts/alloc_probe.hip
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
    } while (0)
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(uint4 *p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(1, 2, 3, 4);
}
int main(int argc, char **argv) {
    const char *tag = argc > 1 ? argv[1] : "quiet";
    double t0 = now_ms();
    CK(hipSetDevice(0));
    CK(hipFree(nullptr));
    double t_init = now_ms() - t0;
    hipStream_t st;
    t0 = now_ms();
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, st, (uint4 *)nullptr, (size_t)0);
    CK(hipStreamSynchronize(st));
    double t_stream = now_ms() - t0;
    printf("{\"tag\": \"%s\", \"runtime_init_ms\": %.2f, \"stream_create_first_launch_ms\": %.2f, \"rows\": [", tag, t_init, t_stream);
    const size_t sizes[] = {(size_t)64 << 20, (size_t)256 << 20, (size_t)1 << 30, (size_t)4 << 30, (size_t)8 << 30};
    bool first = true;
    for (int rep = 0; rep < 3; ++rep)
        for (size_t s : sizes) {
            void *p = nullptr;
            t0 = now_ms(); CK(hipMalloc(&p, s)); double tm = now_ms() - t0;
            t0 = now_ms(); hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, st, (uint4 *)p, s / 16); CK(hipStreamSynchronize(st)); double tt = now_ms() - t0;
            t0 = now_ms(); hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, st, (uint4 *)p, s / 16); CK(hipStreamSynchronize(st)); double tt2 = now_ms() - t0;
            t0 = now_ms(); CK(hipFree(p)); double tf = now_ms() - t0;
            printf("%s{\"rep\": %d, \"mb\": %zu, \"malloc_ms\": %.3f, \"first_touch_ms\": %.3f, \"second_touch_ms\": %.3f, \"free_ms\": %.3f}", first ? "" : ", ", rep, s >> 20, tm, tt, tt2, tf);
            first = false;
        }
    void *hp = nullptr;
    t0 = now_ms(); CK(hipHostMalloc(&hp, (size_t)64 << 20, hipHostMallocDefault)); double th = now_ms() - t0;
    CK(hipHostFree(hp));
    printf("], \"host_malloc_64mb_ms\": %.3f}\n", th);
    return 0;
}
