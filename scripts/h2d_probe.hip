// Host <-> device copy rates on the box, to size the pinned staging ring of the host tier (cm_api.hip: xfer_*):
// pageable hipMemcpy, pinned hipMemcpy, hipHostRegister + copy, and a threaded memcpy -> pinned slot -> DMA pipeline.
// Build: hipcc -O2 --offload-arch=gfx950 -o build_ab/h2d_probe scripts/h2d_probe.hip -lpthread
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
    } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Ring {
    int T = 0;
    size_t slot = 0;
    char *pinned = nullptr;
    std::vector<hipStream_t> st;
    std::vector<hipEvent_t> ev;
};

static void ring_make(Ring &r, int T, size_t slot) {
    r.T = T; r.slot = slot;
    CK(hipHostMalloc((void **)&r.pinned, slot * 2 * T, hipHostMallocDefault));
    r.st.resize(T); r.ev.resize(2 * T);
    for (auto &s : r.st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (auto &e : r.ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
}
static void ring_free(Ring &r) {
    for (auto &s : r.st) CK(hipStreamDestroy(s));
    for (auto &e : r.ev) CK(hipEventDestroy(e));
    CK(hipHostFree(r.pinned));
}

static void h2d_ring(Ring &r, char *dst, const char *src, size_t bytes) {
    const size_t nch = (bytes + r.slot - 1) / r.slot;
    std::vector<std::thread> th;
    for (int t = 0; t < r.T; ++t)
        th.emplace_back([&, t] {
            CK(hipSetDevice(0));
            int i = 0;
            for (size_t c = t; c < nch; c += r.T, ++i) {
                const int s = 2 * t + (i & 1);
                char *p = r.pinned + (size_t)s * r.slot;
                if (i >= 2) CK(hipEventSynchronize(r.ev[s]));
                const size_t off = c * r.slot, len = std::min(r.slot, bytes - off);
                memcpy(p, src + off, len);
                CK(hipMemcpyAsync(dst + off, p, len, hipMemcpyHostToDevice, r.st[t]));
                CK(hipEventRecord(r.ev[s], r.st[t]));
            }
            CK(hipStreamSynchronize(r.st[t]));
        });
    for (auto &x : th) x.join();
}

static void d2h_ring(Ring &r, char *dst, const char *src, size_t bytes) {
    const size_t nch = (bytes + r.slot - 1) / r.slot;
    std::vector<std::thread> th;
    for (int t = 0; t < r.T; ++t)
        th.emplace_back([&, t] {
            CK(hipSetDevice(0));
            std::vector<size_t> mine;
            for (size_t c = t; c < nch; c += r.T) mine.push_back(c);
            auto issue = [&](size_t i) {
                const int s = 2 * t + (int)(i & 1);
                const size_t off = mine[i] * r.slot, len = std::min(r.slot, bytes - off);
                CK(hipMemcpyAsync(r.pinned + (size_t)s * r.slot, src + off, len, hipMemcpyDeviceToHost, r.st[t]));
                CK(hipEventRecord(r.ev[s], r.st[t]));
            };
            if (!mine.empty()) issue(0);
            for (size_t i = 0; i < mine.size(); ++i) {
                if (i + 1 < mine.size()) issue(i + 1);
                const int s = 2 * t + (int)(i & 1);
                CK(hipEventSynchronize(r.ev[s]));
                const size_t off = mine[i] * r.slot, len = std::min(r.slot, bytes - off);
                memcpy(dst + off, r.pinned + (size_t)s * r.slot, len);
            }
        });
    for (auto &x : th) x.join();
}

int main(int argc, char **argv) {
    const size_t MB = argc > 1 ? (size_t)atol(argv[1]) : 300;
    const size_t bytes = MB << 20;
    CK(hipSetDevice(0));
    char *src = (char *)malloc(bytes), *back = (char *)malloc(bytes);
    for (size_t i = 0; i < bytes; i += 4096) src[i] = (char)i;   // touch
    memset(back, 1, bytes);
    char *dev = nullptr;
    CK(hipMalloc((void **)&dev, bytes));
    CK(hipMemset(dev, 0, bytes));
    CK(hipDeviceSynchronize());
    printf("cores visible to this process: %u\n", std::thread::hardware_concurrency());
    auto rate = [&](double dt) { return (double)bytes / dt / 1e9; };
    for (size_t mb : {1, 4, 16, 64, 16, 64}) {   // what does a pinned allocation cost, by size (and the second time)?
        char *pin = nullptr;
        double t0 = now();
        CK(hipHostMalloc((void **)&pin, mb << 20, hipHostMallocDefault));
        double ta = now() - t0;
        t0 = now();
        CK(hipHostFree(pin));
        printf("hipHostMalloc %3zu MB: %.2f ms (free %.2f ms)\n", mb, ta * 1e3, (now() - t0) * 1e3);
    }
    {   // fresh (never touched by the driver) pageable arrays: the first copy of each
        for (int rep = 0; rep < 3; ++rep) {
            char *fresh = (char *)malloc(bytes);
            memset(fresh, rep + 1, bytes);
            double t0 = now();
            CK(hipMemcpy(dev, fresh, bytes, hipMemcpyHostToDevice));
            double dt = now() - t0;
            printf("pageable H2D, fresh array #%d     %7.2f ms  %6.1f GB/s\n", rep, dt * 1e3, rate(dt));
            free(fresh);
        }
    }
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now();
        CK(hipMemcpy(dev, src, bytes, hipMemcpyHostToDevice));
        double dt = now() - t0;
        printf("pageable H2D hipMemcpy           %7.2f ms  %6.1f GB/s\n", dt * 1e3, rate(dt));
    }
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        CK(hipMemcpy(back, dev, bytes, hipMemcpyDeviceToHost));
        double dt = now() - t0;
        printf("pageable D2H hipMemcpy           %7.2f ms  %6.1f GB/s\n", dt * 1e3, rate(dt));
    }
    {
        char *pin = nullptr;
        double t0 = now();
        CK(hipHostMalloc((void **)&pin, bytes, hipHostMallocDefault));
        double ta = now() - t0;
        memcpy(pin, src, bytes);
        for (int rep = 0; rep < 2; ++rep) {
            t0 = now();
            CK(hipMemcpy(dev, pin, bytes, hipMemcpyHostToDevice));
            double dt = now() - t0;
            printf("pinned   H2D hipMemcpy           %7.2f ms  %6.1f GB/s   (hipHostMalloc %.1f ms)\n", dt * 1e3, rate(dt), ta * 1e3);
        }
        t0 = now();
        CK(hipMemcpy(pin, dev, bytes, hipMemcpyDeviceToHost));
        double dt = now() - t0;
        printf("pinned   D2H hipMemcpy           %7.2f ms  %6.1f GB/s\n", dt * 1e3, rate(dt));
        CK(hipHostFree(pin));
    }
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        CK(hipHostRegister(src, bytes, hipHostRegisterDefault));
        double tr = now() - t0;
        t0 = now();
        CK(hipMemcpy(dev, src, bytes, hipMemcpyHostToDevice));
        double dt = now() - t0;
        t0 = now();
        CK(hipHostUnregister(src));
        double tu = now() - t0;
        printf("hipHostRegister %.1f ms + H2D %.2f ms (%.1f GB/s) + unregister %.1f ms\n", tr * 1e3, dt * 1e3, rate(dt), tu * 1e3);
    }
    for (int T : {1, 2, 4, 8, 12}) {
        for (size_t slot_mb : {2, 4, 8}) {
            Ring r;
            ring_make(r, T, slot_mb << 20);
            double best = 1e9, bestd = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                double t0 = now();
                h2d_ring(r, dev, src, bytes);
                best = std::min(best, now() - t0);
            }
            for (int rep = 0; rep < 3; ++rep) {
                double t0 = now();
                d2h_ring(r, back, dev, bytes);
                bestd = std::min(bestd, now() - t0);
            }
            const bool ok = memcmp(back, src, bytes) == 0;
            printf("ring T=%2d slot=%zu MB: H2D %7.2f ms %6.1f GB/s | D2H %7.2f ms %6.1f GB/s  %s\n", T, slot_mb, best * 1e3, rate(best),
                   bestd * 1e3, rate(bestd), ok ? "ok" : "MISMATCH");
            ring_free(r);
        }
    }
    // plain multi-threaded memcpy rate (the staging copy alone)
    for (int T : {1, 4, 8}) {
        double t0 = now();
        std::vector<std::thread> th;
        const size_t per = bytes / T;
        for (int t = 0; t < T; ++t) th.emplace_back([&, t] { memcpy(back + t * per, src + t * per, per); });
        for (auto &x : th) x.join();
        double dt = now() - t0;
        printf("host memcpy T=%d: %.2f ms %.1f GB/s\n", T, dt * 1e3, rate(dt));
    }
    return 0;
}
