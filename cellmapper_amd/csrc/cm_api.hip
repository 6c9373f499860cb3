// C ABI of libcellmap_hip.so (declared in include/cellmap.h): handle management, the device tier (thin
// argument checks around the launchers) and the host tier (H2D, device tier, D2H).
#include "cm_internal.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <array>
#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>

#include <sched.h>

namespace {
// The CELLMAP_* knobs: an immutable record behind an atomic pointer.  Latched ONCE by the first cm_create of the process
// (std::call_once) and replaced as a whole by cm_reload_knobs (tests): a reader on another thread -- the per-device worker
// threads of a multi-device handle among them -- sees the old record or the new one, never a half-written one, and creating
// a second handle no longer changes the behaviour of a live one (round-3 advisor finding).  Superseded records are kept
// (a few hundred bytes per reload; readers may still hold a reference).
std::atomic<const cm_knobs *> g_knobs_ptr{nullptr};
std::once_flag g_knobs_once;
const cm_knobs g_knobs_default{};
void latch_knobs() {
    cm_knobs k;
    auto num = [](const char *n, long long dflt) { const char *e = getenv(n); return (e && *e) ? atoll(e) : dflt; };
    auto on = [](const char *n) { return getenv(n) != nullptr; };
    k.dbg = (int)num("CELLMAP_DBG", 0);
    k.fb_max = (int)num("CELLMAP_FB_MAX", 64);
    k.ivf_submin = (int)num("CELLMAP_IVF_SUBMIN", 32768);
    k.prune_max = (int)std::min<long long>(95, std::max<long long>(12, num("CELLMAP_PRUNE_MAX", 70)));
    k.prune_max_sub = (int)std::min<long long>(95, std::max<long long>(12, num("CELLMAP_PRUNE_MAX_SUB", 60)));
    k.prune_min_nq = (int)std::max<long long>(4096, num("CELLMAP_PRUNE_MIN_NQ", 16384));
    if (const char *e = getenv("CELLMAP_SEED_WIDEN")) k.seed_widen = (float)atof(e);
    k.filter_products = (int)num("CELLMAP_FILTER_PRODUCTS", 0);
    k.q4 = (int)num("CELLMAP_Q4", 1);
    k.r16 = (int)num("CELLMAP_R16", 0);
    k.slack = (int)num("CELLMAP_SLACK", -1);
    if (const char *e = getenv("CELLMAP_PHASE1")) k.phase1 = !strcmp(e, "f32") ? 1 : (!strcmp(e, "f16") ? 2 : (!strcmp(e, "direct") ? 3 : 0));
    k.spgemm_ct = (int)num("CELLMAP_SPGEMM_CT", 0);
    k.spgemm_dbg = (int)num("CELLMAP_SPGEMM_DBG", 0);
    k.a3_gather = (int)num("CELLMAP_A3_GATHER", 0);
    k.spmm_budget_mb = num("CELLMAP_SPMM_BUDGET_MB", -1);
    k.spgemm_budget_mb = num("CELLMAP_SPGEMM_BUDGET_MB", -1);
    k.pool_mb = num("CELLMAP_POOL_MB", -1);
    k.dbg_ws = on("CELLMAP_DBG_WS"); k.weights_lds = on("CELLMAP_WEIGHTS_LDS");
    if (const char *e = getenv("CELLMAP_LOWD")) k.lowd_brute = !strcmp(e, "brute");
    k.no_sort = on("CELLMAP_NO_SORT"); k.no_chain = on("CELLMAP_NO_CHAIN"); k.no_centre = on("CELLMAP_NO_CENTRE");
    k.no_ivf = on("CELLMAP_NO_IVF"); k.no_hint = on("CELLMAP_NO_HINT"); k.no_perm = on("CELLMAP_NO_PERM");
    k.no_seed = on("CELLMAP_NO_SEED"); k.no_subseed = on("CELLMAP_NO_SUBSEED"); k.no_skip3 = on("CELLMAP_NO_SKIP3");
    k.multi_staged = on("CELLMAP_MULTI_STAGED"); k.multi_check_fail = on("CELLMAP_MULTI_CHECK_FAIL"); k.multi_ref_shard = on("CELLMAP_MULTI_REF_SHARD");
    k.no_pretest = on("CELLMAP_NO_PRETEST"); k.no_preload = on("CELLMAP_NO_PRELOAD");
    k.xfer_mode = 3;   // default (round 5): downloads of 64 MB and more go through the pinned staging ring, everything else takes the plain path
    if (const char *e = getenv("CELLMAP_XFER")) k.xfer_mode = !strcmp(e, "d2h") ? 1 : ((!strcmp(e, "0") || !*e) ? 0 : 2);
    k.xfer_threads = (int)num("CELLMAP_XFER_THREADS", 0); k.xfer_slot_kb = (int)num("CELLMAP_XFER_SLOT_KB", 0);
    k.xfer_min_kb = (int)num("CELLMAP_XFER_MIN_KB", 0);
    g_knobs_ptr.store(new cm_knobs(k), std::memory_order_release);
}
std::mutex g_err_mu;
std::string g_create_err;
static inline double wall_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// every device allocation / release of the library goes through these two: the handle keeps the wall time by kind
// (cm_alloc_stats), so a slow first call can be attributed (bench.py: first_call_breakdown_ms)
hipError_t timed_malloc(cm_handle_t h, void **p, size_t bytes, int kind) {
    const double t0 = wall_ms();
    const hipError_t e = hipMalloc(p, bytes);
    if (h) { h->alloc_ms[kind] += wall_ms() - t0; h->alloc_bytes[kind] += (double)bytes; h->alloc_calls[kind] += 1; }
    return e;
}
void timed_free(cm_handle_t h, void *p) {
    const double t0 = wall_ms();
    (void)hipFree(p);
    if (h) { h->alloc_ms[2] += wall_ms() - t0; h->alloc_calls[2] += 1; }
}
void a3_session_drop(cm_handle_t h);
void xfer_drop(cm_handle_t h);

// Device allocation pool of the host tier.  A released block is kept on the handle and handed to the next request it
// fits (no more than twice the requested size); the pool is capped per handle (pool_cap, set by cm_create from the handle's
// own device) and can be returned to the allocator at any time with cm_pool_trim.
void *pool_take(cm_handle_t h, size_t bytes, size_t *got) {
    int best = -1;
    for (int i = 0; i < (int)h->pool.size(); ++i)
        if (h->pool[i].first >= bytes && h->pool[i].first <= 2 * bytes + 4096 && (best < 0 || h->pool[i].first < h->pool[best].first)) best = i;
    if (best < 0) return nullptr;
    void *p = h->pool[best].second;
    *got = h->pool[best].first;
    h->pool_bytes -= h->pool[best].first;
    h->pool.erase(h->pool.begin() + best);
    return p;
}
void pool_release(cm_handle_t h, void *p, size_t bytes) {
    const size_t cap = h ? h->pool_cap : 0;
    if (!h || bytes > cap / 2) { timed_free(h, p); return; }
    h->pool.emplace_back(bytes, p);
    h->pool_bytes += bytes;
    while (h->pool_bytes > cap && !h->pool.empty()) {   // evict the oldest blocks
        timed_free(h, h->pool.front().second);
        h->pool_bytes -= h->pool.front().first;
        h->pool.erase(h->pool.begin());
    }
}
void pool_drop(cm_handle_t h) {
    for (auto &b : h->pool) timed_free(h, b.second);
    h->pool.clear();
    h->pool_bytes = 0;
}

struct DevBuf {  // RAII device allocation used by the host tier (pooled per handle)
    void *p = nullptr;
    size_t cap = 0;
    cm_handle_t owner = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), cap(o.cap), owner(o.owner) { o.p = nullptr; }
    ~DevBuf() { release(); }
    void release() {
        if (p) {
            if (owner) pool_release(owner, p, cap); else timed_free(nullptr, p);
            p = nullptr;
        }
    }
    cm_status alloc(cm_handle_t h, size_t bytes) {
        release();
        if (bytes == 0) bytes = 16;
        bytes = (bytes + 255) & ~size_t(255);
        owner = h;
        if ((p = pool_take(h, bytes, &cap)) != nullptr) return CM_OK;
        hipError_t e = timed_malloc(h, &p, bytes, 1);
        if (e != hipSuccess && h && !h->pool.empty()) {   // give the pool back to the allocator and retry
            pool_drop(h);
            e = timed_malloc(h, &p, bytes, 1);
        }
        if (e != hipSuccess) { p = nullptr; CM_FAIL(h, CM_E_NOMEM, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e)); }
        cap = bytes;
        return CM_OK;
    }
    template <typename T> T *as() { return reinterpret_cast<T *>(p); }
};

cm_status begin_call(cm_handle_t h, size_t ws_bytes) {
    if (!h) return CM_E_INVALID;
    h->err.clear();
    CM_HIP(h, hipSetDevice(h->device));
    cm_ws_reset(h);
    return cm_ws_reserve(h, ws_bytes);
}

// ---- pinned staging ring of the host / graph tiers (round 4; OPT-IN: CELLMAP_XFER=1 | d2h) -----------------------------
// Built on the round-3 review's reading that the host-in -> host-out gap of config 2 (0.33 s against 0.24 s) was pageable
// PCIe traffic.  Measured (scripts/h2d_probe.hip, scripts/host_leg_probe.py, profiles/r04_host_leg.md): it was not.
// hipMemcpy from pageable memory moves 53-56 GB/s on this platform once the driver has seen the pages (15-24 GB/s the
// first time), the link's rate; the gap was numpy's single-threaded isfinite over both inputs (2 x 17 ms: now a device
// kernel, nonfinite_flag) plus the copies themselves (2 x 5.5 ms).  The ring -- T host threads, each with two pinned slots
// and a stream: memcpy chunk c into a slot, hand it to the DMA engine, fill the other slot meanwhile; D2H backwards --
// reaches the same 54 GB/s for uploads, needs ~60 ms to set up and three uses to warm up, so uploads never go through it
// by default.  Downloads into FRESH host arrays are where it pays (the threads fault the pages in parallel: 720 MB of
// k-NN lists 39.6 -> 15.6 ms, the mapping matrix 16.9 -> 8.8 ms), but not enough to cover the set-up in a one-shot
// `CellMapper.map()`: it is therefore opt-in, for processes that pull large results repeatedly.  Both directions are
// SYNCHRONOUS: they first wait for the handle's stream (the destination may be produced or still be in use by queued
// work) and return when the bytes have arrived -- a stronger promise than the plain asynchronous copy, so every call
// site stays correct.  Nothing of the caller's is retained.
struct Xfer {
    int T = 0;
    size_t slot = 0;
    char *pinned = nullptr;
    std::vector<hipStream_t> st;
    std::vector<hipEvent_t> ev;   // two per thread
};
static inline size_t xfer_min() { const int kb = cm_get_knobs().xfer_min_kb; return kb > 0 ? ((size_t)kb << 10) : ((size_t)8 << 20); }   // below this the plain asynchronous copy is as fast

void xfer_drop(cm_handle_t h) {
    Xfer *x = static_cast<Xfer *>(h->xfer);
    if (!x) return;
    for (auto s : x->st) (void)hipStreamDestroy(s);
    for (auto e : x->ev) (void)hipEventDestroy(e);
    if (x->pinned) (void)hipHostFree(x->pinned);
    delete x;
    h->xfer = nullptr;
}
cm_status xfer_get(cm_handle_t h, Xfer **out) {
    if (h->xfer) { *out = static_cast<Xfer *>(h->xfer); return CM_OK; }
    const cm_knobs &kn = cm_get_knobs();
    int T = kn.xfer_threads > 0 ? kn.xfer_threads : 8;   // (downloads into FRESH arrays are bound by page faults, which scale with the threads: 4 threads 34 GB/s, profiles/r05_host_layer.md)
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) T = std::min(T, std::max(1, CPU_COUNT(&set)));
    T = std::max(1, std::min(T, 32));
    Xfer *x = new (std::nothrow) Xfer();
    if (!x) CM_FAIL(h, CM_E_NOMEM, "staging ring: out of host memory");
    x->T = T;
    x->slot = kn.xfer_slot_kb > 0 ? ((size_t)kn.xfer_slot_kb << 10) : ((size_t)4 << 20);
    h->xfer = x;
    const double t0 = wall_ms();
    hipError_t e = hipHostMalloc((void **)&x->pinned, x->slot * 2 * (size_t)T, hipHostMallocDefault);
    h->alloc_ms[3] += wall_ms() - t0; h->alloc_bytes[3] += (double)(x->slot * 2 * (size_t)T); h->alloc_calls[3] += 1;
    if (e != hipSuccess) { x->pinned = nullptr; xfer_drop(h); CM_FAIL(h, CM_E_NOMEM, "hipHostMalloc(staging ring) failed: %s", hipGetErrorString(e)); }
    x->st.assign(T, nullptr); x->ev.assign(2 * T, nullptr);
    for (auto &s : x->st) if ((e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking)) != hipSuccess) break;
    if (e == hipSuccess) for (auto &v : x->ev) if ((e = hipEventCreateWithFlags(&v, hipEventDisableTiming)) != hipSuccess) break;
    if (e != hipSuccess) {
        x->st.erase(std::remove(x->st.begin(), x->st.end(), (hipStream_t) nullptr), x->st.end());
        x->ev.erase(std::remove(x->ev.begin(), x->ev.end(), (hipEvent_t) nullptr), x->ev.end());
        xfer_drop(h);
        CM_FAIL(h, CM_E_HIP, "staging ring: %s", hipGetErrorString(e));
    }
    *out = x;
    return CM_OK;
}
// rows x width bytes between a host array (row pitch hpitch) and a device array (row pitch dpitch); one chunk = whole rows
// that fit a slot, packed densely in the slot.  to_device = false runs the pipeline backwards.
cm_status xfer_run(cm_handle_t h, void *dev, size_t dpitch, void *host, size_t hpitch, size_t width, size_t rows, bool to_device) {
    Xfer *x = nullptr;
    CM_TRY(xfer_get(h, &x));
    CM_HIP(h, hipStreamSynchronize(h->stream));
    size_t rpc, nch;   // rows per chunk, chunks
    const bool flat = dpitch == width && hpitch == width;
    size_t total = rows * width;
    if (flat) { rpc = 0; nch = (total + x->slot - 1) / x->slot; }
    else {
        if (width > x->slot) CM_FAIL(h, CM_E_INTERNAL, "staging ring: a row of %zu bytes does not fit a slot", width);
        rpc = x->slot / width;
        nch = (rows + rpc - 1) / rpc;
    }
    const int T = (int)std::min<size_t>((size_t)x->T, nch);
    std::vector<hipError_t> err((size_t)T, hipSuccess);
    auto chunk_of = [&](size_t c, size_t *r0, size_t *nr, size_t *off, size_t *len) {
        if (flat) { *off = c * x->slot; *len = std::min(x->slot, total - *off); *r0 = 0; *nr = 1; }
        else { *r0 = c * rpc; *nr = std::min(rpc, rows - *r0); *off = 0; *len = *nr * width; }
    };
    auto worker = [&](int t) {
        hipError_t e = hipSetDevice(h->device);
        hipStream_t st = x->st[t];
        auto slot_of = [&](size_t i) { return x->pinned + (size_t)(2 * t + (int)(i & 1)) * x->slot; };
        auto ev_of = [&](size_t i) { return x->ev[2 * t + (int)(i & 1)]; };
        auto dma = [&](size_t c, size_t i) -> hipError_t {   // slot <-> device for chunk c
            size_t r0, nr, off, len;
            chunk_of(c, &r0, &nr, &off, &len);
            char *p = slot_of(i);
            hipError_t q;
            if (flat) q = to_device ? hipMemcpyAsync((char *)dev + off, p, len, hipMemcpyHostToDevice, st)
                                    : hipMemcpyAsync(p, (const char *)dev + off, len, hipMemcpyDeviceToHost, st);
            else if (dpitch == width) q = to_device ? hipMemcpyAsync((char *)dev + r0 * dpitch, p, len, hipMemcpyHostToDevice, st)
                                                    : hipMemcpyAsync(p, (const char *)dev + r0 * dpitch, len, hipMemcpyDeviceToHost, st);
            else q = to_device ? hipMemcpy2DAsync((char *)dev + r0 * dpitch, dpitch, p, width, width, nr, hipMemcpyHostToDevice, st)
                               : hipMemcpy2DAsync(p, width, (const char *)dev + r0 * dpitch, dpitch, width, nr, hipMemcpyDeviceToHost, st);
            if (q == hipSuccess) q = hipEventRecord(ev_of(i), st);
            return q;
        };
        auto host_copy = [&](size_t c, size_t i) {   // caller's array <-> slot for chunk c
            size_t r0, nr, off, len;
            chunk_of(c, &r0, &nr, &off, &len);
            char *p = slot_of(i);
            if (flat) { if (to_device) memcpy(p, (const char *)host + off, len); else memcpy((char *)host + off, p, len); }
            else for (size_t r = 0; r < nr; ++r) {
                if (to_device) memcpy(p + r * width, (const char *)host + (r0 + r) * hpitch, width);
                else memcpy((char *)host + (r0 + r) * hpitch, p + r * width, width);
            }
        };
        size_t i = 0;
        if (to_device) {
            for (size_t c = (size_t)t; c < nch && e == hipSuccess; c += (size_t)T, ++i) {
                if (i >= 2) e = hipEventSynchronize(ev_of(i));   // the DMA that last read this slot
                if (e != hipSuccess) break;
                host_copy(c, i);
                e = dma(c, i);
            }
        } else {
            if ((size_t)t < nch && e == hipSuccess) e = dma((size_t)t, 0);
            for (size_t c = (size_t)t; c < nch && e == hipSuccess; c += (size_t)T, ++i) {
                if (c + (size_t)T < nch) e = dma(c + (size_t)T, i + 1);   // next chunk into the other slot
                if (e == hipSuccess) e = hipEventSynchronize(ev_of(i));
                if (e == hipSuccess) host_copy(c, i);
            }
        }
        const hipError_t e2 = hipStreamSynchronize(st);
        err[(size_t)t] = e != hipSuccess ? e : e2;
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(worker, t);
    worker(0);
    for (auto &w : th) w.join();
    for (int t = 0; t < T; ++t)
        if (err[(size_t)t] != hipSuccess) CM_FAIL(h, CM_E_HIP, "staged %s copy failed: %s", to_device ? "host-to-device" : "device-to-host", hipGetErrorString(err[(size_t)t]));
    return CM_OK;
}

cm_status h2d(cm_handle_t h, void *dst, const void *src, size_t bytes) {
    if (bytes >= xfer_min() && cm_get_knobs().xfer_mode == 2) return xfer_run(h, dst, bytes, const_cast<void *>(src), bytes, bytes, 1, true);
    if (bytes) CM_HIP(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream));
    return CM_OK;
}
// does a download of `bytes` (into what is usually a freshly allocated host array) go through the staging ring?  Default: from
// 64 MB on -- the ring's threads fault the destination's pages in parallel, which is what bounds a plain copy into fresh memory
// (720 MB of k-NN lists: 39.6 -> 15.6 ms, profiles/r04_host_leg.md); the ring is created on first use, so calls that only bring
// small results down (labels, confidences) never pay its ~30 ms set-up.  CELLMAP_XFER=0 switches it off, =d2h / =1 lower the
// threshold to CELLMAP_XFER_MIN_KB (8 MB).
static inline bool ring_for_download(size_t bytes) {
    const int m = cm_get_knobs().xfer_mode;
    return m == 3 ? bytes >= ((size_t)64 << 20) : (m >= 1 && bytes >= xfer_min());
}
cm_status d2h(cm_handle_t h, void *dst, const void *src, size_t bytes) {
    if (ring_for_download(bytes)) return xfer_run(h, const_cast<void *>(src), bytes, dst, bytes, bytes, 1, false);
    if (bytes) CM_HIP(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
    return CM_OK;
}
cm_status sync(cm_handle_t h) {
    CM_HIP(h, hipStreamSynchronize(h->stream));
    return CM_OK;
}
cm_status check_fallback_overflow(cm_handle_t h) {
    h->knn_stats[0] = h->d_nflag_pinned[0];
    if (h->d_nflag_pinned[1]) CM_FAIL(h, CM_E_UNSUPPORTED, "k-NN exact fallback overflow: more than 16384 queries each have more than 4096 reference rows tied within their k-th distance");
    return CM_OK;
}

// Host tier of A3: what cm_spmm_csr_nnz uploaded (mapping matrix, payload) and computed (output indptr) stays on the
// device until the matching cm_spmm_csr call, so the count + fill pair uploads everything once.
struct A3Session {
    const void *key[6] = {};
    int64_t nq = 0, n_ref = 0, n_out = 0;
    int64_t token = 0;   // fingerprint of the output row pointer the count call produced (pointer identity alone is not
                         // enough: numpy / the pool hand out recycled addresses)
    DevBuf dp, dc, dv, bp, bc, bv, op;
    DevBuf oc, ov, pidx;   // single-pass form (round 3): the product's entries, made during the count call and kept for the fill call
    bool filled = false;
    // the mapping matrix on the device: the session's own upload (dp / dc / dv) or a cm_graph's arrays
    const int64_t *mp = nullptr;
    const int32_t *mc = nullptr;
    const float *mv = nullptr;
};
void a3_session_drop(cm_handle_t h) {
    if (h->a3_session) { delete static_cast<A3Session *>(h->a3_session); h->a3_session = nullptr; }
}
// second stream + two events for the tiled host-tier transfers (created on first use)
cm_status tile_streams(cm_handle_t h) {
    if (!h->copy_stream) CM_HIP(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i)
        if (!h->tile_ev[i]) CM_HIP(h, hipEventCreateWithFlags(&h->tile_ev[i], hipEventDisableTiming));
    return CM_OK;
}
size_t device_budget(cm_handle_t h, long long knob_mb) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t)8 << 30;
    size_t budget = free_b / 2;
    if (knob_mb >= 0) budget = (size_t)knob_mb << 20;   // testing: force tiling
    (void)h;
    return budget;
}

// NaN / infinity screen of an uploaded array (sklearn's check_array raises ValueError on such input before any search,
// knn.py:429-440 -> sklearn:utils/validation.py): one pass over the device copy (16-byte loads; 300 MB in ~0.1 ms) instead of
// numpy's single-threaded isfinite over the host array (17 ms per 300 MB: it was a third of the host-tier overhead).
__global__ __launch_bounds__(256) void nonfinite_flag(const uint4 *__restrict__ x, size_t n16, const unsigned char *__restrict__ tail,
                                                      int n_tail_elems, int is_f64, int *__restrict__ flag) {
    // verdict bits: 1 = an infinity, 2 = a NaN (round 5: CM_METRIC_NAN_EUCLIDEAN accepts NaN as "missing", nothing accepts infinities)
    bool inf = false, nan = false;
    auto f32 = [&](unsigned w) { if ((w & 0x7f800000u) == 0x7f800000u) { if (w & 0x007fffffu) nan = true; else inf = true; } };
    auto f64 = [&](unsigned lo, unsigned hi) { if ((hi & 0x7ff00000u) == 0x7ff00000u) { if ((hi & 0x000fffffu) | lo) nan = true; else inf = true; } };
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = x[i];
        if (is_f64) { f64(v.x, v.y); f64(v.z, v.w); }
        else { f32(v.x); f32(v.y); f32(v.z); f32(v.w); }
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < n_tail_elems) {   // the last < 16 bytes
        if (is_f64) { const uint2 w = reinterpret_cast<const uint2 *>(tail)[threadIdx.x]; f64(w.x, w.y); }
        else f32(reinterpret_cast<const unsigned *>(tail)[threadIdx.x]);
    }
    const int bits = (__ballot(inf) != 0ull ? 1 : 0) | (__ballot(nan) != 0ull ? 2 : 0);
    if (bits && (threadIdx.x & 63) == 0) atomicOr(flag, bits);
}
// enqueue the screen of `bytes` of device memory at d_x (16-byte aligned) on the handle's stream; the verdict lands in
// pinned word `slot` of the handle (read it after the next synchronisation)
cm_status nonfinite_enqueue(cm_handle_t h, const void *d_x, size_t bytes, int dtype, int *d_flag, int slot) {
    const size_t es = dtype == CM_F64 ? 8 : 4;
    const size_t n16 = bytes / 16;
    const int n_tail = (int)((bytes - 16 * n16) / es);
    CM_HIP(h, hipMemsetAsync(d_flag, 0, 4, h->stream));
    const unsigned grid = (unsigned)std::min<size_t>(std::max<size_t>((n16 + 255) / 256, 1), 4096);
    hipLaunchKernelGGL(nonfinite_flag, dim3(grid), dim3(256), 0, h->stream, (const uint4 *)d_x, n16, (const unsigned char *)d_x + 16 * n16, n_tail,
                       dtype == CM_F64 ? 1 : 0, d_flag);
    CM_HIP(h, hipGetLastError());
    CM_HIP(h, hipMemcpyAsync(h->d_nflag_pinned + slot, d_flag, 4, hipMemcpyDeviceToHost, h->stream));
    return CM_OK;
}
#define CM_NONFINITE_MSG "Input contains NaN or infinity."
#define CM_INF_MSG "Input contains infinity or a value too large."

// param for the graph kernel from the distances (device), honouring the reference's global statistics
cm_status kernel_param_dev(cm_handle_t h, const double *d_dist, const int64_t *d_idx, int64_t n, int kernel, double *d_m3,
                           double *d_param) {
    if (kernel != CM_KERNEL_GAUSSIAN && kernel != CM_KERNEL_SCARCHES) return CM_OK;
    CM_HIP(h, hipMemsetAsync(d_m3, 0, 3 * sizeof(double), h->stream));
    CM_TRY(cm_moments_run(h, d_dist, d_idx, n, nullptr, d_m3));
    CM_TRY(cm_param_run(h, d_m3, CM_KERNEL_GAUSSIAN, d_param));  // = mean
    if (kernel == CM_KERNEL_SCARCHES) {
        double *d_mean = d_m3 + 3;
        CM_HIP(h, hipMemcpyAsync(d_mean, d_param, sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        CM_HIP(h, hipMemsetAsync(d_m3, 0, 3 * sizeof(double), h->stream));
        CM_TRY(cm_moments_run(h, d_dist, d_idx, n, d_mean, d_m3));
        CM_TRY(cm_param_run(h, d_m3, CM_KERNEL_SCARCHES, d_param));
    }
    return CM_OK;
}
}  // namespace

const cm_knobs &cm_get_knobs() {
    const cm_knobs *p = g_knobs_ptr.load(std::memory_order_acquire);
    return p ? *p : g_knobs_default;
}
static __global__ void add_offset_i64_touch() {}   // (preload of this translation unit's kernels: cm_create)

void cm_mark(cm_handle_t h, int which) {
    if (!h->profiling || which < 0 || which >= CM_N_STAGE_EVENTS) return;
    if (!h->ev[which] && hipEventCreate(&h->ev[which]) != hipSuccess) return;
    if (hipEventRecord(h->ev[which], h->stream) == hipSuccess) h->ev_set[which] = true;
}

// ----------------------------------------------------------------------------- workspace ------------
static cm_status ws_add_segment(cm_handle_t h, size_t bytes) {
    char *p = nullptr;
    hipError_t e = timed_malloc(h, (void **)&p, bytes, 0);
    if (e != hipSuccess && !h->pool.empty()) {   // give the host tier's parked blocks back to the allocator and retry
        pool_drop(h);
        e = timed_malloc(h, (void **)&p, bytes, 0);
    }
    if (e != hipSuccess) CM_FAIL(h, CM_E_NOMEM, "hipMalloc(%zu bytes of workspace) failed: %s", bytes, hipGetErrorString(e));
    h->ws_segs.push_back({p, bytes, 0});
    h->ws_cap += bytes;
    return CM_OK;
}
cm_status cm_ws_reserve(cm_handle_t h, size_t bytes) {
    bytes = cm_align(bytes) + 4096;
    if (bytes <= h->ws_cap) return CM_OK;
    return ws_add_segment(h, std::max<size_t>(bytes - h->ws_cap, (size_t)1 << 20));
}
void cm_ws_reset(cm_handle_t h) {
    for (auto &s : h->ws_segs) s.off = 0;
    h->ws_used = 0;
}
void *cm_ws_alloc(cm_handle_t h, size_t bytes) {
    bytes = cm_align(bytes ? bytes : 1);
    cm_handle_s::ws_seg *hit = nullptr;
    for (auto &s : h->ws_segs)
        if (s.off + bytes <= s.cap) { hit = &s; break; }
    if (!hit) {   // no segment has room: grow (at least 64 MB, at most what is asked for plus a sixteenth of what the handle holds)
        const size_t seg = std::max(bytes, std::max<size_t>((size_t)64 << 20, h->ws_cap / 16));
        if (cm_get_knobs().dbg_ws) fprintf(stderr, "[cellmap dbg] workspace: request of %zu bytes does not fit the %zu segments (%zu bytes): adding %zu\n", bytes, h->ws_segs.size(), h->ws_cap, seg);
        (void)hipSetDevice(h->device);
        if (ws_add_segment(h, seg) != CM_OK) return nullptr;
        hit = &h->ws_segs.back();
    }
    void *p = hit->p + hit->off;
    hit->off += bytes;
    h->ws_used += bytes;
    if (h->ws_used > h->ws_peak) h->ws_peak = h->ws_used;
    return p;
}

extern "C" {

int cm_version(void) { return 200; }

int cm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

cm_status cm_reload_knobs(void) {
    latch_knobs();
    return CM_OK;
}

cm_status cm_create(int device, cm_handle_t *out) {
    if (!out) return CM_E_INVALID;
    *out = nullptr;
    const double t_create0 = wall_ms();
    std::call_once(g_knobs_once, latch_knobs);   // the CELLMAP_* knobs are read by the FIRST cm_create (and by cm_reload_knobs), never on a call path
    auto fail = [&](const char *what, hipError_t e) {
        std::lock_guard<std::mutex> g(g_err_mu);
        g_create_err = std::string(what) + ": " + hipGetErrorString(e);
        return CM_E_HIP;
    };
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) return fail("no HIP device available (libcellmap_hip needs an AMD GPU; gfx950 code objects)", e);
    if (device < 0 || device >= n) { std::lock_guard<std::mutex> g(g_err_mu); g_create_err = "device index out of range"; return CM_E_INVALID; }
    if ((e = hipSetDevice(device)) != hipSuccess) return fail("hipSetDevice", e);
    cm_handle_s *h = new (std::nothrow) cm_handle_s();
    if (!h) return CM_E_NOMEM;
    h->device = device;
    if ((e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking)) != hipSuccess) { delete h; return fail("hipStreamCreate", e); }
    h->stream = h->own_stream;
    if ((e = hipHostMalloc((void **)&h->d_nflag_pinned, 64, hipHostMallocDefault)) != hipSuccess) { (void)hipStreamDestroy(h->own_stream); delete h; return fail("hipHostMalloc", e); }
    memset(h->d_nflag_pinned, 0, 64);
    {   // pool cap from THIS device's memory (not from whichever device happens to be current later)
        size_t free_b = 0, total_b = 0;
        h->pool_cap = (hipMemGetInfo(&free_b, &total_b) == hipSuccess) ? std::min<size_t>(total_b / 8, (size_t)32 << 30) : ((size_t)8 << 30);
        if (cm_get_knobs().pool_mb >= 0) h->pool_cap = (size_t)cm_get_knobs().pool_mb << 20;
    }
    const double t_pre0 = wall_ms();
    if (!cm_get_knobs().no_preload) {   // load the hot path's code objects now, not inside the first call
        cm_preload_knn(h->stream); cm_preload_q4(h->stream); cm_preload_f16(h->stream); cm_preload_graph(h->stream);   // (the opt-in / K = 128 kernels load on first use)
        hipLaunchKernelGGL(add_offset_i64_touch, dim3(1), dim3(64), 0, h->stream);
        (void)hipStreamSynchronize(h->stream);
        (void)hipGetLastError();
    }
    h->preload_ms = wall_ms() - t_pre0;
    h->create_ms = wall_ms() - t_create0;
    *out = h;
    return CM_OK;
}

cm_status cm_alloc_stats(cm_handle_t h, double out[16]) {
    if (!h || !out) return CM_E_INVALID;
    for (int i = 0; i < 4; ++i) { out[i] = h->alloc_ms[i]; out[4 + i] = h->alloc_bytes[i]; out[8 + i] = h->alloc_calls[i]; }
    out[12] = h->create_ms; out[13] = h->preload_ms; out[14] = (double)h->ws_cap; out[15] = (double)h->ws_peak;
    return CM_OK;
}

int64_t cm_workspace_bytes(int64_t n_ref, int64_t nq, int32_t d, int32_t dtype, int32_t k, int32_t metric) {
    if (n_ref < 1 || nq < 0 || d < 1 || k < 1 || (dtype != CM_F32 && dtype != CM_F64)) return -1;
    const size_t nk = (size_t)nq * (size_t)k;   // + the fused pipeline's own lists (cm_map_labels_dev)
    return (int64_t)(cm_knn_ws_bytes(n_ref, nq, d, k, dtype, metric) + cm_align(8 * nk) * 2 + cm_align(4 * nk) * 2 + cm_align(4 * (size_t)nq) + 64 * 1024);
}

int64_t cm_workspace_bytes_all_routes(int64_t n_ref, int64_t nq, int32_t d, int32_t dtype, int32_t k, int32_t metric) {
    if (n_ref < 1 || nq < 0 || d < 1 || k < 1 || (dtype != CM_F32 && dtype != CM_F64)) return -1;
    const size_t nk = (size_t)nq * (size_t)k;
    return (int64_t)(cm_knn_ws_bytes_all_routes(n_ref, nq, d, k, dtype, metric) + cm_align(8 * nk) * 2 + cm_align(4 * nk) * 2 + cm_align(4 * (size_t)nq) + 64 * 1024);
}

cm_status cm_reserve_workspace(cm_handle_t h, int64_t bytes) {
    if (!h || bytes < 0) return CM_E_INVALID;
    h->err.clear();
    CM_HIP(h, hipSetDevice(h->device));
    return cm_ws_reserve(h, (size_t)bytes);
}

cm_status cm_pool_trim(cm_handle_t h) {
    if (!h) return CM_E_INVALID;
    for (auto *p : h->peers) (void)cm_pool_trim(p);
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    a3_session_drop(h);
    pool_drop(h);
    return CM_OK;
}

cm_status cm_destroy(cm_handle_t h) {
    if (!h) return CM_OK;
    for (auto *p : h->peers) (void)cm_destroy(p);
    h->peers.clear();
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    for (auto &sg : h->ws_segs) (void)hipFree(sg.p);
    h->ws_segs.clear();
    a3_session_drop(h);
    pool_drop(h);
    xfer_drop(h);
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    for (int i = 0; i < 2; ++i) if (h->tile_ev[i]) (void)hipEventDestroy(h->tile_ev[i]);
    if (h->d_nflag_pinned) (void)hipHostFree(h->d_nflag_pinned);
    for (int i = 0; i < CM_N_STAGE_EVENTS; ++i) if (h->ev[i]) (void)hipEventDestroy(h->ev[i]);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
    return CM_OK;
}

const char *cm_last_error(cm_handle_t h) {
    if (h) return h->err.c_str();
    std::lock_guard<std::mutex> g(g_err_mu);
    return g_create_err.c_str();
}

cm_status cm_set_stream(cm_handle_t h, void *hip_stream) {
    if (!h) return CM_E_INVALID;
    h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    return CM_OK;
}

cm_status cm_set_profiling(cm_handle_t h, int32_t on) {
    if (!h) return CM_E_INVALID;
    h->profiling = on != 0;
    for (int i = 0; i < CM_N_STAGE_EVENTS; ++i) h->ev_set[i] = false;
    return CM_OK;
}

cm_status cm_stage_times(cm_handle_t h, float out_ms[8]) {
    if (!h || !out_ms) return CM_E_INVALID;
    CM_HIP(h, hipSetDevice(h->device));
    CM_HIP(h, hipStreamSynchronize(h->stream));
    int prev = -1;
    for (int i = 0; i < CM_N_STAGE_EVENTS; ++i) {
        out_ms[i] = 0.f;
        if (!h->ev_set[i]) continue;
        if (prev >= 0) CM_HIP(h, hipEventElapsedTime(&out_ms[i], h->ev[prev], h->ev[i]));
        prev = i;
    }
    return CM_OK;
}

cm_status cm_synchronize(cm_handle_t h) {
    if (!h) return CM_E_INVALID;
    CM_HIP(h, hipSetDevice(h->device));
    CM_TRY(sync(h));
    return check_fallback_overflow(h);
}

cm_status cm_knn_stats(cm_handle_t h, int64_t out[8]) {
    if (!h || !out) return CM_E_INVALID;
    h->knn_stats[0] = h->d_nflag_pinned[0];
    for (int i = 0; i < 4; ++i) out[i] = h->knn_stats[i];
    out[4] = h->d_nflag_pinned[1];   // exact-fallback overflow (valid once the stream has been synchronised)
    out[5] = h->knn_sub_splits;
    out[6] = h->knn_wide_retries;
    out[7] = 0;
    return CM_OK;
}

// ------------------------------------------------------------------------------ device tier ---------
cm_status cm_knn_dev(cm_handle_t h, const void *d_ref, int64_t n_ref, const void *d_qry, int64_t nq, int32_t d,
                     int32_t dtype, int32_t k, int32_t metric, int32_t form, int64_t idx_offset, int32_t flags,
                     void *d_out_dist, int64_t *d_out_idx) {
    CM_TRY(begin_call(h, cm_knn_ws_bytes(n_ref, nq, d, k, dtype, metric)));
    return cm_knn_run(h, d_ref, n_ref, d_qry, nq, d, dtype, k, metric, form, idx_offset, flags, d_out_dist, d_out_idx);
}

cm_status cm_knn_merge_dev(cm_handle_t h, const double *d_dist, const int64_t *d_idx, int32_t nparts, int64_t nq,
                           int32_t k, int32_t sqrt_out, double *d_out_dist, int64_t *d_out_idx) {
    CM_TRY(begin_call(h, 0));
    return cm_knn_merge_run(h, d_dist, d_idx, nparts, nq * k, nq, k, sqrt_out, d_out_dist, d_out_idx);
}

cm_status cm_knn_merge_strided_dev(cm_handle_t h, const double *d_dist, const int64_t *d_idx, int32_t nparts,
                                   int64_t part_stride, int64_t nq, int32_t k, int32_t sqrt_out, double *d_out_dist,
                                   int64_t *d_out_idx) {
    CM_TRY(begin_call(h, 0));
    return cm_knn_merge_run(h, d_dist, d_idx, nparts, part_stride, nq, k, sqrt_out, d_out_dist, d_out_idx);
}

cm_status cm_dist_moments_dev(cm_handle_t h, const double *d_dist, const int64_t *d_idx, int64_t n,
                              const double *d_center, double *d_out3) {
    CM_TRY(begin_call(h, 64 * 1024));
    return cm_moments_run(h, d_dist, d_idx, n, d_center, d_out3);
}

cm_status cm_kernel_param_dev(cm_handle_t h, const double *d_moments3, int32_t kernel, double *d_param) {
    CM_TRY(begin_call(h, 0));
    return cm_param_run(h, d_moments3, kernel, d_param);
}

cm_status cm_weights_dev(cm_handle_t h, const double *d_dist, const int64_t *d_idx, int64_t nq, int32_t k,
                         int64_t n_ref, int32_t kernel, const double *d_param, double epsilon, int32_t *d_cols,
                         float *d_vals, int32_t *d_cnt, double *d_conn) {
    CM_TRY(begin_call(h, 0));
    return cm_weights_run(h, d_dist, d_idx, nq, k, n_ref, kernel, d_param, epsilon, d_cols, d_vals, d_cnt, d_conn);
}

cm_status cm_ell_to_csr_dev(cm_handle_t h, const int32_t *d_cols, const float *d_vals, const int32_t *d_cnt,
                            int64_t nq, int32_t k, int64_t *d_indptr, int32_t *d_out_cols, float *d_out_vals) {
    CM_TRY(begin_call(h, cm_scan_ws_bytes(nq)));
    return cm_ell_to_csr_run(h, d_cols, d_vals, nullptr, d_cnt, nq, k, d_indptr, d_out_cols, d_out_vals, nullptr);
}

cm_status cm_label_vote_dev(cm_handle_t h, const int64_t *d_indptr, const int32_t *d_cols, const float *d_vals,
                            int64_t nq, const int32_t *d_codes, int64_t /*n_ref*/, int32_t n_cat, int32_t *d_out_code,
                            float *d_out_conf) {
    CM_TRY(begin_call(h, 0));
    if (!d_indptr) CM_FAIL(h, CM_E_INVALID, "label vote: indptr is NULL");
    return cm_vote_run(h, d_indptr, d_cols, d_vals, nq, 0, nullptr, d_codes, n_cat, d_out_code, d_out_conf);
}

cm_status cm_spmm_dense_dev(cm_handle_t h, const int64_t *d_indptr, const int32_t *d_cols, const float *d_vals,
                            int64_t nq, const void *d_B, int64_t /*n_ref*/, int64_t n_out, int32_t b_dtype, void *d_out) {
    CM_TRY(begin_call(h, 0));
    return cm_spmm_dense_run(h, d_indptr, d_cols, d_vals, nq, d_B, n_out, b_dtype, d_out);
}

cm_status cm_locality_order_dev(cm_handle_t h, const int64_t *d_indptr, const int32_t *d_cols, int64_t nq, int64_t n_ref,
                                int32_t *d_order) {
    CM_TRY(begin_call(h, cm_locality_order_ws_bytes(n_ref)));
    if (nq < 0 || n_ref < 1 || !d_indptr || !d_order) CM_FAIL(h, CM_E_INVALID, "locality order: bad arguments");
    if (nq > 0x7fffffffLL) CM_FAIL(h, CM_E_UNSUPPORTED, "locality order: more than 2^31 rows");
    return cm_locality_order_run(h, d_indptr, d_cols, nq, n_ref, d_order);
}

cm_status cm_spmm_csr_nnz_dev(cm_handle_t h, const int64_t *d_indptr, const int32_t *d_cols, const float *d_vals, int64_t nq,
                              const int64_t *d_b_indptr, const int32_t *d_b_cols, const float *d_b_vals, int64_t /*n_ref*/,
                              int64_t n_out, int64_t *d_out_indptr, const int32_t *d_row_order) {
    CM_TRY(begin_call(h, cm_scan_ws_bytes(nq) + cm_align(4 * (size_t)std::max<int64_t>(nq, 1)) + 4096));
    if (nq < 0 || !d_out_indptr) CM_FAIL(h, CM_E_INVALID, "spmm_csr: bad arguments");
    int32_t *rn = (int32_t *)cm_ws_alloc(h, 4 * (size_t)std::max<int64_t>(nq, 1));
    if (!rn) CM_FAIL(h, CM_E_NOMEM, "spmm_csr: device workspace exhausted");
    CM_TRY(cm_spmm_csr_count_run(h, d_indptr, d_cols, d_vals, nq, d_b_indptr, d_b_cols, d_b_vals, n_out, rn, d_row_order));
    return cm_exclusive_scan_i32_to_i64(h, rn, nq, d_out_indptr);
}

cm_status cm_spmm_csr_dev(cm_handle_t h, const int64_t *d_indptr, const int32_t *d_cols, const float *d_vals, int64_t nq,
                          const int64_t *d_b_indptr, const int32_t *d_b_cols, const float *d_b_vals, int64_t /*n_ref*/,
                          int64_t n_out, const int64_t *d_out_indptr, int32_t *d_out_cols, float *d_out_vals,
                          const int32_t *d_row_order) {
    CM_TRY(begin_call(h, 0));
    return cm_spmm_csr_fill_run(h, d_indptr, d_cols, d_vals, nq, d_b_indptr, d_b_cols, d_b_vals, n_out, d_out_indptr, 0, d_out_cols,
                                d_out_vals, d_row_order);
}

cm_status cm_spmm_csr_fused_dev(cm_handle_t h, const int64_t *d_indptr, const int32_t *d_cols, const float *d_vals, int64_t nq,
                                const int64_t *d_b_indptr, const int32_t *d_b_cols, const float *d_b_vals, int64_t n_ref,
                                int64_t n_out, int64_t capacity, int64_t *d_out_indptr, int32_t *d_out_cols, float *d_out_vals,
                                const int32_t *d_row_order, int32_t *d_out_rows, const int32_t *d_payload_index) {
    CM_TRY(begin_call(h, cm_align(cm_a3_chain_bytes(nq)) + 4096));
    if (nq < 0 || capacity < 0 || !d_out_indptr || (capacity > 0 && (!d_out_cols || !d_out_vals)))
        CM_FAIL(h, CM_E_INVALID, "spmm_csr (single pass): bad arguments");
    if (!cm_spmm_csr_single_pass_ok(n_out))
        CM_FAIL(h, CM_E_INVALID, "spmm_csr (single pass): %lld output columns do not fit one accumulator pass; use "
                                 "cm_spmm_csr_nnz_dev + cm_spmm_csr_dev", (long long)n_out);
    unsigned long long *chain = (unsigned long long *)cm_ws_alloc(h, cm_a3_chain_bytes(nq));
    if (!chain) CM_FAIL(h, CM_E_NOMEM, "spmm_csr: device workspace exhausted");
    if (d_payload_index)
        return cm_spmm_csr_sliced_run(h, d_indptr, d_cols, d_vals, nq, d_b_indptr, d_b_cols, d_b_vals, n_ref, d_payload_index, n_out,
                                      d_out_indptr, d_out_cols, d_out_vals, capacity, d_row_order, chain, d_out_rows);
    return cm_spmm_csr_fused_run(h, d_indptr, d_cols, d_vals, nq, d_b_indptr, d_b_cols, d_b_vals, n_out, d_out_indptr, d_out_cols,
                                 d_out_vals, capacity, d_row_order, chain, d_out_rows);
}

int64_t cm_payload_index_len(int64_t n_ref, int64_t n_out) { return cm_payload_index_len_impl(n_ref, n_out); }

cm_status cm_payload_index_dev(cm_handle_t h, const int64_t *d_b_indptr, const int32_t *d_b_cols, int64_t n_ref, int64_t n_out,
                               int32_t *d_index) {
    CM_TRY(begin_call(h, 4096));
    if (n_ref < 0 || !d_b_indptr || !d_index) CM_FAIL(h, CM_E_INVALID, "payload index: bad arguments");
    if (cm_payload_index_len_impl(n_ref, n_out) < 0)
        CM_FAIL(h, CM_E_INVALID, "payload index: the single-pass product covers 1 .. 32768 output columns (got %lld)", (long long)n_out);
    int *flag = (int *)cm_ws_alloc(h, 64);
    if (!flag) CM_FAIL(h, CM_E_NOMEM, "payload index: device workspace exhausted");
    h->a3_index = nullptr;   // (an index rebuilt at the same address: the header kept on the handle is stale)
    CM_TRY(cm_payload_index_run(h, d_b_indptr, d_b_cols, n_ref, n_out, d_index, flag));
    int bad = 0;
    CM_HIP(h, hipMemcpyAsync(&bad, flag, 4, hipMemcpyDeviceToHost, h->stream));
    CM_TRY(sync(h));
    int max_slice; long long nnz;
    CM_TRY(cm_payload_index_header(h, d_index, &max_slice, &nnz));
    if (bad)
        CM_FAIL(h, CM_E_INVALID, "payload index: every payload row needs strictly ascending column indices inside [0, n_out) "
                                 "(canonical CSR: scipy's sum_duplicates() / sort_indices())");
    return CM_OK;
}

cm_status cm_map_labels_dev(cm_handle_t h, const void *d_ref, int64_t n_ref, const void *d_qry, int64_t nq, int32_t d,
                            int32_t dtype, int32_t k, int32_t metric, int32_t kernel, double epsilon,
                            const int32_t *d_codes, int32_t n_cat, int32_t *d_out_code, float *d_out_conf,
                            double *d_out_dist, int64_t *d_out_idx) {
    size_t nk = (size_t)nq * k;
    size_t extra = cm_align(8 * nk) * 2 + cm_align(4 * nk) * 2 + cm_align(4 * (size_t)nq) + 64 * 1024;
    CM_TRY(begin_call(h, cm_knn_ws_bytes(n_ref, nq, d, k, dtype, metric) + extra));
    double *dist = d_out_dist ? d_out_dist : (double *)cm_ws_alloc(h, 8 * nk);
    int64_t *idx = d_out_idx ? d_out_idx : (int64_t *)cm_ws_alloc(h, 8 * nk);
    int32_t *cols = (int32_t *)cm_ws_alloc(h, 4 * nk);
    float *vals = (float *)cm_ws_alloc(h, 4 * nk);
    int32_t *cnt = (int32_t *)cm_ws_alloc(h, 4 * (size_t)nq);
    double *m3 = (double *)cm_ws_alloc(h, 64);
    if (!dist || !idx || !cols || !vals || !cnt || !m3) CM_FAIL(h, CM_E_NOMEM, "fused pipeline: device workspace exhausted");
    double *param = m3 + 5;
    // (cosine on float32 input: the fused pipeline keeps the float64 values the search ranked by, CM_KNN_F64_OUT)
    CM_TRY(cm_knn_run(h, d_ref, n_ref, d_qry, nq, d, dtype, k, metric, CM_FORM_AUTO, 0, CM_KNN_F64_OUT, dist, idx));
    CM_TRY(kernel_param_dev(h, dist, idx, (int64_t)nk, kernel, m3, param));
    cm_mark(h, 5);
    CM_TRY(cm_weights_run(h, dist, idx, nq, k, n_ref, kernel, param, epsilon, cols, vals, cnt, nullptr));
    cm_mark(h, 6);
    CM_TRY(cm_vote_run(h, nullptr, cols, vals, nq, k, cnt, d_codes, n_cat, d_out_code, d_out_conf));
    cm_mark(h, 7);
    return CM_OK;
}

// -------------------------------------------------------------------------------- host tier ---------
cm_status cm_knn(cm_handle_t h, const void *ref, int64_t n_ref, const void *qry, int64_t nq, int32_t d,
                 int32_t dtype, int32_t k, int32_t metric, int32_t form, void *out_dist, int64_t *out_idx) {
    if (!h) return CM_E_INVALID;
    if (dtype != CM_F32 && dtype != CM_F64) CM_FAIL(h, CM_E_INVALID, "k-NN: unknown dtype id %d", dtype);
    if (n_ref < 1 || d < 1 || nq < 0) CM_FAIL(h, CM_E_INVALID, "k-NN: bad shape");
    if (k < 1 || k > n_ref)
        CM_FAIL(h, CM_E_INVALID, "Expected n_neighbors <= n_samples_fit, but n_neighbors = %d, n_samples_fit = %lld", k, (long long)n_ref);
    if (!h->peers.empty()) {   // multi-device handle: the graph tier shards the reference rows over the devices (falls back to one by itself)
        cm_rep_t r = nullptr, q = nullptr;
        cm_graph_t g = nullptr;
        const int32_t rflags = metric == CM_METRIC_NAN_EUCLIDEAN ? CM_REP_ALLOW_NAN : 0;
        cm_status rc = cm_rep_create_ex(h, ref, n_ref, d, dtype, rflags, &r);
        if (rc == CM_OK) rc = cm_rep_create_ex(h, qry, nq, d, dtype, rflags, &q);
        if (rc == CM_OK) rc = cm_graph_knn(h, r, q, k, metric, form, &g);
        if (rc == CM_OK) rc = cm_graph_knn_fetch(h, g, out_dist, out_idx);
        const std::string keep = h->err;
        if (g) (void)cm_graph_destroy(h, g);
        if (q) (void)cm_rep_destroy(h, q);
        if (r) (void)cm_rep_destroy(h, r);
        if (rc != CM_OK) h->err = keep;
        return rc;
    }
    const size_t es = dtype == CM_F64 ? 8 : 4;
    const size_t des = (metric == CM_METRIC_COSINE) ? es : 8;  // element size of the returned distances
    const int64_t QCH = 1 << 21;
    const int64_t chunk = std::min<int64_t>(nq, QCH);
    CM_TRY(begin_call(h, cm_knn_ws_bytes(n_ref, chunk, d, k, dtype, metric)));
    DevBuf dref, dq, dd, di;
    CM_TRY(dref.alloc(h, es * n_ref * d));
    CM_TRY(dq.alloc(h, es * chunk * d));
    CM_TRY(dd.alloc(h, des * chunk * k));
    CM_TRY(di.alloc(h, 8 * (size_t)chunk * k));
    CM_TRY(h2d(h, dref.p, ref, es * n_ref * d));
    // NaN / infinity screen of both inputs on their device copies (ValueError upstream: sklearn's check_array)
    DevBuf dflag;
    CM_TRY(dflag.alloc(h, 256));
    h->d_nflag_pinned[12] = h->d_nflag_pinned[13] = 0;
    CM_TRY(nonfinite_enqueue(h, dref.p, es * n_ref * d, dtype, dflag.as<int>(), 12));
    if (nq == 0) {   // no query chunk will read the reference's verdict: read it here (sklearn's check_array rejects the fit data before any query)
        CM_TRY(sync(h));
        if (h->d_nflag_pinned[12] & (metric == CM_METRIC_NAN_EUCLIDEAN ? 1 : 3)) CM_FAIL(h, CM_E_INVALID, metric == CM_METRIC_NAN_EUCLIDEAN ? CM_INF_MSG : CM_NONFINITE_MSG);
    }
    int64_t nflag = 0;
    for (int64_t q0 = 0; q0 < nq; q0 += chunk) {
        int64_t nqc = std::min<int64_t>(chunk, nq - q0);
        cm_ws_reset(h);
        CM_TRY(h2d(h, dq.p, (const char *)qry + es * q0 * d, es * nqc * d));
        CM_TRY(nonfinite_enqueue(h, dq.p, es * nqc * d, dtype, dflag.as<int>() + 1, 13));
        CM_TRY(sync(h));
        if ((h->d_nflag_pinned[12] | h->d_nflag_pinned[13]) & (metric == CM_METRIC_NAN_EUCLIDEAN ? 1 : 3))
            CM_FAIL(h, CM_E_INVALID, metric == CM_METRIC_NAN_EUCLIDEAN ? CM_INF_MSG : CM_NONFINITE_MSG);
        CM_TRY(cm_knn_run(h, dref.p, n_ref, dq.p, nqc, d, dtype, k, metric, form, 0, 0, dd.p, di.as<int64_t>()));
        CM_TRY(d2h(h, (char *)out_dist + des * q0 * k, dd.p, des * nqc * k));
        CM_TRY(d2h(h, out_idx + q0 * k, di.p, 8 * (size_t)nqc * k));
        CM_TRY(sync(h));
        CM_TRY(check_fallback_overflow(h));
        nflag += h->knn_stats[0];
    }
    h->knn_stats[0] = nflag;
    h->d_nflag_pinned[0] = (int32_t)std::min<int64_t>(nflag, 0x7fffffff);
    return CM_OK;
}

cm_status cm_mapping_matrix(cm_handle_t h, const double *dist, const int64_t *idx, int64_t nq, int32_t k,
                            int64_t n_ref, int32_t kernel, double epsilon, int64_t *out_indptr, int32_t *out_cols,
                            float *out_vals, double *out_conn, int64_t *out_nnz, double *out_param) {
    if (!h) return CM_E_INVALID;
    if (nq < 0 || k < 1) CM_FAIL(h, CM_E_INVALID, "mapping matrix: bad shape");
    if (kernel < 0 || kernel > CM_KERNEL_IDENTITY) CM_FAIL(h, CM_E_INVALID, "Unknown kernel id %d", kernel);
    size_t nk = (size_t)nq * k;
    // (phase clock: a phase that takes more than a second is reported on stderr -- DESIGN.md section 8, "host-tier calls that stall")
    double ph_t = wall_ms();
    auto phase = [&](const char *what) {
        const double now = wall_ms();
        if (now - ph_t > 1000.0) fprintf(stderr, "[cellmap] cm_mapping_matrix: %.1f s in phase '%s' (nq=%lld, k=%d)\n", (now - ph_t) * 1e-3, what, (long long)nq, (int)k);
        ph_t = now;
    };
    CM_TRY(begin_call(h, 64 * 1024 + cm_scan_ws_bytes(nq)));
    phase("begin_call");
    DevBuf dd, di, dcols, dvals, dconn, dcnt, dptr, ocols, ovals, oconn, dm;
    CM_TRY(dd.alloc(h, 8 * nk)); CM_TRY(di.alloc(h, 8 * nk));
    CM_TRY(dcols.alloc(h, 4 * nk)); CM_TRY(dvals.alloc(h, 4 * nk)); CM_TRY(dcnt.alloc(h, 4 * (size_t)nq));
    CM_TRY(dptr.alloc(h, 8 * ((size_t)nq + 1))); CM_TRY(ocols.alloc(h, 4 * nk)); CM_TRY(ovals.alloc(h, 4 * nk));
    CM_TRY(dm.alloc(h, 64));
    if (out_conn) { CM_TRY(dconn.alloc(h, 8 * nk)); CM_TRY(oconn.alloc(h, 8 * nk)); }
    phase("device buffers");
    CM_TRY(h2d(h, dd.p, dist, 8 * nk));
    CM_TRY(h2d(h, di.p, idx, 8 * nk));
    phase("uploads (enqueue)");
    double *m3 = dm.as<double>(), *param = m3 + 5;
    CM_HIP(h, hipMemsetAsync(m3, 0, 64, h->stream));
    CM_TRY(cm_moments_run(h, dd.as<double>(), di.as<int64_t>(), (int64_t)nk, nullptr, m3));
    double hm[3] = {0, 0, 0};
    CM_TRY(d2h(h, hm, m3, 24));
    phase("moments (enqueue) + read-back (enqueue)");
    CM_TRY(sync(h));
    phase("first stream wait");
    if (!(hm[2] > 0.0)) CM_FAIL(h, CM_E_NOFINITE, "No finite distances found in the neighborhood graph");
    CM_TRY(kernel_param_dev(h, dd.as<double>(), di.as<int64_t>(), (int64_t)nk, kernel, m3, param));
    CM_TRY(cm_weights_run(h, dd.as<double>(), di.as<int64_t>(), nq, k, n_ref, kernel, param, epsilon, dcols.as<int32_t>(),
                          dvals.as<float>(), dcnt.as<int32_t>(), out_conn ? dconn.as<double>() : nullptr));
    CM_TRY(cm_ell_to_csr_run(h, dcols.as<int32_t>(), dvals.as<float>(), out_conn ? dconn.as<double>() : nullptr,
                             dcnt.as<int32_t>(), nq, k, dptr.as<int64_t>(), ocols.as<int32_t>(), ovals.as<float>(),
                             out_conn ? oconn.as<double>() : nullptr));
    phase("weights + CSR kernels (enqueue)");
    CM_TRY(d2h(h, out_indptr, dptr.p, 8 * ((size_t)nq + 1)));
    double hp = 0.0;
    CM_TRY(d2h(h, &hp, param, 8));
    CM_TRY(sync(h));
    phase("second stream wait");
    int64_t nnz = out_indptr[nq];
    CM_TRY(d2h(h, out_cols, ocols.p, 4 * (size_t)nnz));
    CM_TRY(d2h(h, out_vals, ovals.p, 4 * (size_t)nnz));
    if (out_conn) CM_TRY(d2h(h, out_conn, oconn.p, 8 * (size_t)nnz));
    CM_TRY(sync(h));
    phase("downloads + last stream wait");
    if (out_nnz) *out_nnz = nnz;
    if (out_param) *out_param = (kernel == CM_KERNEL_GAUSSIAN || kernel == CM_KERNEL_SCARCHES) ? hp : 0.0;
    return CM_OK;
}

cm_status cm_csr_normalize(cm_handle_t h, const int64_t *indptr, const double *data, int64_t nq, float *out_vals) {
    if (!h) return CM_E_INVALID;
    CM_TRY(begin_call(h, 0));
    int64_t nnz = indptr[nq];
    DevBuf dp, dd, dout;
    CM_TRY(dp.alloc(h, 8 * ((size_t)nq + 1))); CM_TRY(dd.alloc(h, 8 * (size_t)nnz)); CM_TRY(dout.alloc(h, 4 * (size_t)nnz));
    CM_TRY(h2d(h, dp.p, indptr, 8 * ((size_t)nq + 1)));
    CM_TRY(h2d(h, dd.p, data, 8 * (size_t)nnz));
    CM_TRY(cm_csr_normalize_run(h, dp.as<int64_t>(), dd.as<double>(), nq, dout.as<float>()));
    CM_TRY(d2h(h, out_vals, dout.p, 4 * (size_t)nnz));
    return sync(h);
}

cm_status cm_presence_scores(cm_handle_t h, const double *dist, const int64_t *idx, int64_t nq, int32_t k, int64_t n_ref,
                             const int32_t *group, int32_t n_groups, double *out) {
    if (!h) return CM_E_INVALID;
    if (nq < 0 || k < 1 || n_ref < 1 || n_groups < 0) CM_FAIL(h, CM_E_INVALID, "presence scores: bad shape");
    CM_TRY(begin_call(h, 64 * 1024));
    size_t nk = (size_t)nq * k, no = (size_t)n_ref * (n_groups + 1);
    DevBuf dd, di, dg, dout, dm;
    CM_TRY(dd.alloc(h, 8 * nk)); CM_TRY(di.alloc(h, 8 * nk)); CM_TRY(dout.alloc(h, 8 * no)); CM_TRY(dm.alloc(h, 64));
    CM_TRY(h2d(h, dd.p, dist, 8 * nk));
    CM_TRY(h2d(h, di.p, idx, 8 * nk));
    if (group) { CM_TRY(dg.alloc(h, 4 * (size_t)nq)); CM_TRY(h2d(h, dg.p, group, 4 * (size_t)nq)); }
    CM_HIP(h, hipMemsetAsync(dout.p, 0, 8 * no, h->stream));
    double *m3 = dm.as<double>(), *param = m3 + 5;
    CM_HIP(h, hipMemsetAsync(m3, 0, 64, h->stream));
    CM_TRY(cm_moments_run(h, dd.as<double>(), di.as<int64_t>(), (int64_t)nk, nullptr, m3));
    double hm[3] = {0, 0, 0};
    CM_TRY(d2h(h, hm, m3, 24));
    CM_TRY(sync(h));
    if (!(hm[2] > 0.0)) CM_FAIL(h, CM_E_NOFINITE, "No finite distances found in the neighborhood graph");
    CM_TRY(cm_param_run(h, m3, CM_KERNEL_GAUSSIAN, param));
    CM_TRY(cm_presence_run(h, dd.as<double>(), di.as<int64_t>(), nq, k, param, group ? dg.as<int32_t>() : nullptr, n_groups,
                           dout.as<double>()));
    CM_TRY(d2h(h, out, dout.p, 8 * no));
    return sync(h);
}

static cm_status upload_csr(cm_handle_t h, const int64_t *indptr, const int32_t *cols, const float *vals, int64_t nq,
                            DevBuf &dp, DevBuf &dc, DevBuf &dv) {
    int64_t nnz = indptr[nq];
    CM_TRY(dp.alloc(h, 8 * ((size_t)nq + 1))); CM_TRY(dc.alloc(h, 4 * (size_t)nnz)); CM_TRY(dv.alloc(h, 4 * (size_t)nnz));
    CM_TRY(h2d(h, dp.p, indptr, 8 * ((size_t)nq + 1)));
    CM_TRY(h2d(h, dc.p, cols, 4 * (size_t)nnz));
    CM_TRY(h2d(h, dv.p, vals, 4 * (size_t)nnz));
    return CM_OK;
}

cm_status cm_label_vote(cm_handle_t h, const int64_t *indptr, const int32_t *cols, const float *vals, int64_t nq,
                        const int32_t *codes, int64_t n_ref, int32_t n_cat, int32_t *out_code, float *out_conf) {
    if (!h) return CM_E_INVALID;
    CM_TRY(begin_call(h, 0));
    DevBuf dp, dc, dv, dcodes, ocode, oconf;
    CM_TRY(upload_csr(h, indptr, cols, vals, nq, dp, dc, dv));
    CM_TRY(dcodes.alloc(h, 4 * (size_t)n_ref)); CM_TRY(ocode.alloc(h, 4 * (size_t)nq)); CM_TRY(oconf.alloc(h, 4 * (size_t)nq));
    CM_TRY(h2d(h, dcodes.p, codes, 4 * (size_t)n_ref));
    CM_TRY(cm_vote_run(h, dp.as<int64_t>(), dc.as<int32_t>(), dv.as<float>(), nq, 0, nullptr, dcodes.as<int32_t>(), n_cat,
                       ocode.as<int32_t>(), oconf.as<float>()));
    CM_TRY(d2h(h, out_code, ocode.p, 4 * (size_t)nq));
    CM_TRY(d2h(h, out_conf, oconf.p, 4 * (size_t)nq));
    return sync(h);
}

// c_begin / c_end: only the payload columns [c_begin, c_end) are transferred (gene-column sharding of the multi-device handle);
// `out` is always the full [nq, n_out] array
static cm_status spmm_dense_stream(cm_handle_t h, const int64_t *d_indptr, const int32_t *d_cols, const float *d_vals, int64_t nq,
                                   const void *B, int64_t n_ref, int64_t n_out, int32_t b_dtype, void *out, int64_t c_begin = 0,
                                   int64_t c_end = -1) {
    if (c_end < 0) c_end = n_out;
    if (b_dtype != CM_F32 && b_dtype != CM_F64) CM_FAIL(h, CM_E_INVALID, "spmm: unknown dtype id %d", b_dtype);
    if (n_out < 1) CM_FAIL(h, CM_E_INVALID, "spmm: payload has no columns");
    CM_TRY(tile_streams(h));
    const size_t es = b_dtype == CM_F64 ? 8 : 4;
    DevBuf dB, dO[2];
    // Streaming ("wide aggregation outputs"): the payload is cut into column slabs that fit HBM next
    // to two output tiles, the output into query tiles; tile i is copied to the caller's buffer on a second stream while
    // tile i+1 is computed.  Nothing of size nq x n_out is ever resident.
    const size_t budget = device_budget(h, cm_get_knobs().spmm_budget_mb);
    int64_t cw = std::max<int64_t>(c_end - c_begin, 1);
    int64_t qt = std::max<int64_t>(1, std::min<int64_t>(nq, 1 << 16));           // query rows per output tile
    auto need = [&](int64_t w, int64_t t) { return es * ((size_t)n_ref * w + 2 * (size_t)t * w); };
    while (need(cw, qt) > budget && qt > 1024) qt /= 2;
    if (need(cw, qt) > budget) {
        cw = (int64_t)(budget / (es * ((size_t)n_ref + 2 * (size_t)qt)));
        cw = cw / 64 * 64;
        if (cw < 64) CM_FAIL(h, CM_E_NOMEM, "spmm: not enough device memory for a 64-column slab");
    }
    CM_TRY(dB.alloc(h, es * (size_t)n_ref * cw));
    CM_TRY(dO[0].alloc(h, es * (size_t)qt * cw));
    CM_TRY(dO[1].alloc(h, es * (size_t)qt * cw));
    // the product comes down into a fresh [nq, n_out] host array: 64 MB and more of it go tile by tile through the staging
    // ring (parallel first touch of the destination; the plain copy ran at 20-30 GB/s over the link for that reason, DESIGN.md
    // section 3b).  The ring call waits for the tile's kernel and returns when the rows have arrived: the product itself is a
    // tenth of the transfer time, so nothing is lost by not overlapping them.
    const bool ring = ring_for_download(es * (size_t)nq * (size_t)(c_end - c_begin)) && es * (size_t)cw <= ((size_t)1 << 20);
    for (int64_t c0 = c_begin; c0 < c_end; c0 += cw) {
        const int64_t w = std::min<int64_t>(cw, c_end - c0);
        CM_HIP(h, hipMemcpy2DAsync(dB.p, es * w, (const char *)B + es * c0, es * n_out, es * w, (size_t)n_ref, hipMemcpyHostToDevice, h->stream));
        int it = 0;
        for (int64_t q0 = 0; q0 < nq; q0 += qt, ++it) {
            const int64_t nt = std::min<int64_t>(qt, nq - q0);
            const int b = it & 1;
            if (ring) {
                CM_TRY(cm_spmm_dense_run(h, d_indptr + q0, d_cols, d_vals, nt, dB.p, w, b_dtype, dO[0].p));
                CM_TRY(xfer_run(h, dO[0].p, es * w, (char *)out + es * ((size_t)q0 * n_out + c0), es * n_out, es * w, (size_t)nt, false));
                continue;
            }
            if (it >= 2) CM_HIP(h, hipStreamWaitEvent(h->stream, h->tile_ev[b], 0));   // the copy of tile it-2 has left this buffer
            CM_TRY(cm_spmm_dense_run(h, d_indptr + q0, d_cols, d_vals, nt, dB.p, w, b_dtype, dO[b].p));
            CM_HIP(h, hipEventRecord(h->tile_ev[b], h->stream));
            CM_HIP(h, hipStreamWaitEvent(h->copy_stream, h->tile_ev[b], 0));
            CM_HIP(h, hipMemcpy2DAsync((char *)out + es * ((size_t)q0 * n_out + c0), es * n_out, dO[b].p, es * w, es * w, (size_t)nt,
                                       hipMemcpyDeviceToHost, h->copy_stream));
            CM_HIP(h, hipEventRecord(h->tile_ev[b], h->copy_stream));
        }
        CM_HIP(h, hipStreamSynchronize(h->copy_stream));
        CM_TRY(sync(h));
    }
    return CM_OK;
}

cm_status cm_spmm_dense(cm_handle_t h, const int64_t *indptr, const int32_t *cols, const float *vals, int64_t nq,
                        const void *B, int64_t n_ref, int64_t n_out, int32_t b_dtype, void *out) {
    if (!h) return CM_E_INVALID;
    CM_TRY(begin_call(h, 0));
    DevBuf dp, dc, dv;
    CM_TRY(upload_csr(h, indptr, cols, vals, nq, dp, dc, dv));   // the mapping matrix goes up once
    return spmm_dense_stream(h, dp.as<int64_t>(), dc.as<int32_t>(), dv.as<float>(), nq, B, n_ref, n_out, b_dtype, out);
}

// shared by the host tier (mapping matrix in host arrays) and the graph tier (mapping matrix already on the device: d_m*)
// upper bound of the product's entry count: row i has at most min(n_out, sum of its neighbours' payload row lengths) entries
static __global__ void a3_row_upper_bound(const int64_t *__restrict__ mp, const int32_t *__restrict__ mc, int64_t nq,
                                          const int64_t *__restrict__ bp, int64_t n_out, unsigned long long *__restrict__ total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    long long ub = 0;
    if (i < nq) {
        for (int64_t p = mp[i]; p < mp[i + 1]; ++p) { const int64_t j = mc[p]; ub += bp[j + 1] - bp[j]; }
        if (ub > n_out) ub = n_out;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ub += __shfl_xor(ub, o);
    if ((threadIdx.x & 63) == 0 && ub) atomicAdd(total, (unsigned long long)ub);
}

// fingerprint of a host row pointer: its end and 64 evenly spaced entries
static int64_t indptr_token(const int64_t *p, int64_t nq) {
    uint64_t t = 0x9E3779B97F4A7C15ull ^ (uint64_t)p[nq];
    const int64_t step = std::max<int64_t>(1, nq / 64);
    for (int64_t i = 0; i <= nq; i += step) t = (t ^ (uint64_t)p[i]) * 0x100000001B3ull;
    return (int64_t)t;
}

static cm_status spmm_csr_nnz_impl(cm_handle_t h, const int64_t *indptr, const int32_t *cols, const float *vals,
                                   const int64_t *d_mp, const int32_t *d_mc, const float *d_mv, int64_t nq,
                                   const int64_t *b_indptr, const int32_t *b_cols, const float *b_vals, int64_t n_ref,
                                   int64_t n_out, int64_t *out_indptr) {
    if (nq < 0 || n_ref < 0 || !b_indptr || !out_indptr) CM_FAIL(h, CM_E_INVALID, "spmm_csr: bad arguments");
    CM_TRY(begin_call(h, cm_scan_ws_bytes(nq) + cm_align(cm_a3_chain_bytes(nq)) + 8192));
    a3_session_drop(h);
    A3Session *ss = new (std::nothrow) A3Session();
    if (!ss) CM_FAIL(h, CM_E_NOMEM, "spmm_csr: out of host memory");
    h->a3_session = ss;
    const void *key[6] = {d_mp ? (const void *)d_mp : (const void *)indptr, d_mp ? (const void *)d_mc : (const void *)cols,
                          d_mp ? (const void *)d_mv : (const void *)vals, b_indptr, b_cols, b_vals};
    memcpy(ss->key, key, sizeof(key));
    ss->nq = nq; ss->n_ref = n_ref; ss->n_out = n_out;
    DevBuf rn;
    cm_status rc = CM_OK;
    if (d_mp) { ss->mp = d_mp; ss->mc = d_mc; ss->mv = d_mv; }
    else {
        rc = upload_csr(h, indptr, cols, vals, nq, ss->dp, ss->dc, ss->dv);
        ss->mp = ss->dp.as<int64_t>(); ss->mc = ss->dc.as<int32_t>(); ss->mv = ss->dv.as<float>();
    }
    if (rc == CM_OK) rc = upload_csr(h, b_indptr, b_cols, b_vals, n_ref, ss->bp, ss->bc, ss->bv);
    if (rc == CM_OK) rc = ss->op.alloc(h, 8 * ((size_t)nq + 1));
    // Single-pass form (round 3): when a worst-case buffer for the WHOLE product fits the device budget, counts, offsets and
    // entries come from ONE gather (the wave-sliced kernel of cm_spmm_csr_fused_dev) during this call and stay on the device;
    // the fill call then only copies them out -- the count pass used to repeat the whole product to learn the row lengths.
    if (rc == CM_OK && nq > 0 && n_ref > 0 && cm_spmm_csr_single_pass_ok(n_out)) {
        unsigned long long *d_ub = (unsigned long long *)cm_ws_alloc(h, 64);
        unsigned long long ub = 0;
        if (d_ub && hipMemsetAsync(d_ub, 0, 8, h->stream) == hipSuccess) {
            hipLaunchKernelGGL(a3_row_upper_bound, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, h->stream, ss->mp, ss->mc, nq,
                               ss->bp.as<int64_t>(), n_out, d_ub);
            if (d2h(h, &ub, d_ub, 8) == CM_OK && sync(h) == CM_OK && ub > 0) {
                const int64_t ilen = cm_payload_index_len_impl(n_ref, n_out);
                const size_t need = 8 * (size_t)ub + 4 * (size_t)std::max<int64_t>(ilen, 0) + cm_a3_chain_bytes(nq);
                if (ilen > 0 && need <= device_budget(h, cm_get_knobs().spgemm_budget_mb) / 2) {
                    int *flag = (int *)cm_ws_alloc(h, 64);
                    unsigned long long *chain = (unsigned long long *)cm_ws_alloc(h, cm_a3_chain_bytes(nq));
                    int bad = 1;
                    if (flag && chain && ss->pidx.alloc(h, 4 * (size_t)ilen) == CM_OK && ss->oc.alloc(h, 4 * (size_t)ub) == CM_OK &&
                        ss->ov.alloc(h, 4 * (size_t)ub) == CM_OK) {
                        h->a3_index = nullptr;
                        if (cm_payload_index_run(h, ss->bp.as<int64_t>(), ss->bc.as<int32_t>(), n_ref, n_out, ss->pidx.as<int32_t>(), flag) == CM_OK &&
                            d2h(h, &bad, flag, 4) == CM_OK && sync(h) == CM_OK && !bad) {
                            rc = cm_spmm_csr_sliced_run(h, ss->mp, ss->mc, ss->mv, nq, ss->bp.as<int64_t>(), ss->bc.as<int32_t>(), ss->bv.as<float>(),
                                                        n_ref, ss->pidx.as<int32_t>(), n_out, ss->op.as<int64_t>(), ss->oc.as<int32_t>(),
                                                        ss->ov.as<float>(), (int64_t)ub, nullptr, chain, nullptr);
                            if (rc == CM_OK) rc = d2h(h, out_indptr, ss->op.p, 8 * ((size_t)nq + 1));
                            if (rc == CM_OK) rc = sync(h);
                            if (rc == CM_OK) { ss->filled = true; ss->token = indptr_token(out_indptr, nq); }
                            if (rc != CM_OK) a3_session_drop(h);
                            return rc;
                        }
                    }
                    ss->pidx.release(); ss->oc.release(); ss->ov.release();   // (no room, or a payload row is not canonical CSR: two passes)
                    h->err.clear();
                }
            }
        }
    }
    if (rc == CM_OK) rc = rn.alloc(h, 4 * (size_t)nq);
    if (rc == CM_OK)
        rc = cm_spmm_csr_count_run(h, ss->mp, ss->mc, ss->mv, nq, ss->bp.as<int64_t>(), ss->bc.as<int32_t>(), ss->bv.as<float>(), n_out,
                                   rn.as<int32_t>());
    if (rc == CM_OK) rc = cm_exclusive_scan_i32_to_i64(h, rn.as<int32_t>(), nq, ss->op.as<int64_t>());
    if (rc == CM_OK) rc = d2h(h, out_indptr, ss->op.p, 8 * ((size_t)nq + 1));
    if (rc == CM_OK) rc = sync(h);
    if (rc == CM_OK) ss->token = indptr_token(out_indptr, nq);
    if (rc != CM_OK) a3_session_drop(h);
    return rc;
}

static cm_status spmm_csr_fill_impl(cm_handle_t h, const int64_t *indptr, const int32_t *cols, const float *vals,
                                    const int64_t *d_mp, const int32_t *d_mc, const float *d_mv, int64_t nq,
                                    const int64_t *b_indptr, const int32_t *b_cols, const float *b_vals, int64_t n_ref,
                                    int64_t n_out, const int64_t *out_indptr, int32_t *out_cols, float *out_vals) {
    if (nq < 0 || n_ref < 0 || !b_indptr || !out_indptr) CM_FAIL(h, CM_E_INVALID, "spmm_csr: bad arguments");
    CM_TRY(begin_call(h, 4096));
    CM_TRY(tile_streams(h));
    const void *key[6] = {d_mp ? (const void *)d_mp : (const void *)indptr, d_mp ? (const void *)d_mc : (const void *)cols,
                          d_mp ? (const void *)d_mv : (const void *)vals, b_indptr, b_cols, b_vals};
    A3Session *ss = static_cast<A3Session *>(h->a3_session);
    A3Session local;
    if (!(ss && !memcmp(ss->key, key, sizeof(key)) && ss->nq == nq && ss->n_ref == n_ref && ss->n_out == n_out &&
          ss->token == indptr_token(out_indptr, nq))) {
        // no matching count call on this handle: upload here
        a3_session_drop(h);
        ss = &local;
        if (d_mp) { ss->mp = d_mp; ss->mc = d_mc; ss->mv = d_mv; }
        else {
            CM_TRY(upload_csr(h, indptr, cols, vals, nq, ss->dp, ss->dc, ss->dv));
            ss->mp = ss->dp.as<int64_t>(); ss->mc = ss->dc.as<int32_t>(); ss->mv = ss->dv.as<float>();
        }
        CM_TRY(upload_csr(h, b_indptr, b_cols, b_vals, n_ref, ss->bp, ss->bc, ss->bv));
        CM_TRY(ss->op.alloc(h, 8 * ((size_t)nq + 1)));
        CM_TRY(h2d(h, ss->op.p, out_indptr, 8 * ((size_t)nq + 1)));
    }
    if (ss != &local && ss->filled) {   // single-pass session: the entries are already on the device
        const int64_t nnz_ = out_indptr[nq];
        cm_status rcf = CM_OK;
        if (nnz_ > 0) {
            rcf = d2h(h, out_cols, ss->oc.p, 4 * (size_t)nnz_);
            if (rcf == CM_OK) rcf = d2h(h, out_vals, ss->ov.p, 4 * (size_t)nnz_);
        }
        if (rcf == CM_OK) rcf = sync(h);
        a3_session_drop(h);
        return rcf;
    }
    // Query tiles: the output (which approaches nq x n_out for dense neighbourhoods) is produced in tiles of at most `cap`
    // entries, copied to the caller's arrays on a second stream while the next tile is computed.
    const size_t budget = device_budget(h, cm_get_knobs().spgemm_budget_mb);
    const int64_t nnz = out_indptr[nq];
    int64_t cap = std::max<int64_t>((int64_t)(budget / 16), 1);   // two buffers of (int32 column + float32 value) per entry
    int64_t longest = 0;
    for (int64_t i = 0; i < nq; ++i) longest = std::max<int64_t>(longest, out_indptr[i + 1] - out_indptr[i]);
    if (cap < longest) CM_FAIL(h, CM_E_NOMEM, "spmm_csr: not enough device memory for one output row (%lld entries)", (long long)longest);
    cap = std::min<int64_t>(cap, std::max<int64_t>(nnz, 1));
    DevBuf oc[2], ov[2];
    for (int b = 0; b < 2; ++b) { CM_TRY(oc[b].alloc(h, 4 * (size_t)cap)); CM_TRY(ov[b].alloc(h, 4 * (size_t)cap)); }
    cm_status rc = CM_OK;
    int it = 0;
    for (int64_t q0 = 0; q0 < nq && rc == CM_OK; ++it) {
        // rows [q0, q1): as many as fit `cap` entries (binary search on the host copy of the output indptr)
        const int64_t base = out_indptr[q0];
        int64_t q1 = std::upper_bound(out_indptr + q0, out_indptr + nq + 1, base + cap) - out_indptr - 1;
        if (q1 <= q0) q1 = q0 + 1;
        const int64_t tn = out_indptr[q1] - base;
        const int b = it & 1;
        if (it >= 2) rc = hipStreamWaitEvent(h->stream, h->tile_ev[b], 0) == hipSuccess ? CM_OK : CM_E_HIP;
        if (rc == CM_OK)
            rc = cm_spmm_csr_fill_run(h, ss->mp + q0, ss->mc, ss->mv, q1 - q0, ss->bp.as<int64_t>(), ss->bc.as<int32_t>(),
                                      ss->bv.as<float>(), n_out, ss->op.as<int64_t>() + q0, base, oc[b].as<int32_t>(), ov[b].as<float>());
        if (rc == CM_OK && tn > 0) {
            CM_HIP(h, hipEventRecord(h->tile_ev[b], h->stream));
            CM_HIP(h, hipStreamWaitEvent(h->copy_stream, h->tile_ev[b], 0));
            CM_HIP(h, hipMemcpyAsync(out_cols + base, oc[b].p, 4 * (size_t)tn, hipMemcpyDeviceToHost, h->copy_stream));
            CM_HIP(h, hipMemcpyAsync(out_vals + base, ov[b].p, 4 * (size_t)tn, hipMemcpyDeviceToHost, h->copy_stream));
            CM_HIP(h, hipEventRecord(h->tile_ev[b], h->copy_stream));
        }
        q0 = q1;
    }
    if (rc == CM_OK) { CM_HIP(h, hipStreamSynchronize(h->copy_stream)); rc = sync(h); }
    else { (void)hipStreamSynchronize(h->copy_stream); (void)hipStreamSynchronize(h->stream); }
    a3_session_drop(h);
    return rc;
}

cm_status cm_spmm_csr_nnz(cm_handle_t h, const int64_t *indptr, const int32_t *cols, const float *vals, int64_t nq,
                          const int64_t *b_indptr, const int32_t *b_cols, const float *b_vals, int64_t n_ref,
                          int64_t n_out, int64_t *out_indptr) {
    if (!h) return CM_E_INVALID;
    if (!indptr) CM_FAIL(h, CM_E_INVALID, "spmm_csr: bad arguments");
    return spmm_csr_nnz_impl(h, indptr, cols, vals, nullptr, nullptr, nullptr, nq, b_indptr, b_cols, b_vals, n_ref, n_out, out_indptr);
}

cm_status cm_spmm_csr(cm_handle_t h, const int64_t *indptr, const int32_t *cols, const float *vals, int64_t nq,
                      const int64_t *b_indptr, const int32_t *b_cols, const float *b_vals, int64_t n_ref,
                      int64_t n_out, const int64_t *out_indptr, int32_t *out_cols, float *out_vals) {
    if (!h) return CM_E_INVALID;
    if (!indptr) CM_FAIL(h, CM_E_INVALID, "spmm_csr: bad arguments");
    return spmm_csr_fill_impl(h, indptr, cols, vals, nullptr, nullptr, nullptr, nq, b_indptr, b_cols, b_vals, n_ref, n_out, out_indptr,
                              out_cols, out_vals);
}

// ------------------------------------------------------------------ graph tier: device-resident staged flow ----
// What the drop-in Python classes hold between compute_neighbors -> compute_mapping_matrix -> map_obs / map_obsm /
// map_layers (one in-process flow upstream, cellmapper.py:465-484): the embeddings, the k-NN result and the mapping matrix
// stay on the device; host arrays are produced only when the caller asks for them (the *_fetch calls).
//
// Multi-device (cm_create_multi, SURVEY.md 8e behind the C ABI): a representation is uploaded to EVERY device of the handle;
// a graph is a list of PARTS, part p = the query rows [q_lo, q_lo + nq) whose merged lists, mapping-matrix rows and transfers
// device p owns.  A single-device graph is one part on the handle itself and runs exactly the code it always ran.
struct cm_rep_s {
    // bufs[0]: the WHOLE representation on the handle's own device (the one PCIe upload).  bufs[i], i > 0 (multi-device
    // handle): rows [lo[i], lo[i] + rows[i]) on peer device i, copied from bufs[0] over xGMI when a graph first needs them --
    // a reference shard for a rep used as the reference, every row for one that is queried (round 4; round 3 uploaded the
    // whole array to every device over PCIe whether or not the graph sharded at all)
    std::vector<DevBuf> bufs;
    std::vector<int64_t> lo, rows;
    int64_t n = 0;
    int32_t d = 0, dtype = 0;
    bool has_nan = false;   // uploaded with CM_REP_ALLOW_NAN and holds at least one NaN: only CM_METRIC_NAN_EUCLIDEAN may search it
    size_t row_bytes() const { return (size_t)d * (dtype == CM_F64 ? 8 : 4); }
};
struct GraphPart {
    cm_handle_t h = nullptr;
    int64_t q_lo = 0, nq = 0;
    int64_t ref_lo = 0, ref_rows = 0;              // reference rows this part's device searched (multi-device: its shard)
    int64_t ref_bytes = 0, qry_bytes = 0;          // bytes of the two representations resident on the part's device for the search
    DevBuf dist, idx;              // k-NN result as the reference returns it: float64 (float32: cosine on float32 input), int64
    DevBuf indptr, cols, vals;     // row-normalised mapping matrix rows of the part, canonical CSR (indptr starts at 0)
    int64_t nnz = 0;
    DevBuf f_indptr, f_cols, f_vals;   // multi-device: the WHOLE mapping matrix on this part's device (built on first use by a wide
    bool has_full = false;             // dense transfer: gene-column sharding needs every row on every device)
};
struct cm_graph_s {
    int64_t nq = 0, n_ref = 0;
    int32_t k = 0, metric = 0, dist_is_f32 = 0;
    int32_t exchange_route = 0;   // multi-device: 0 = peer copies (hipMemcpyPeerAsync), 1 = staged through host memory, 2 = none (query slices, reference replicated)
    int32_t self_check = 0;       // multi-device: 0 = not run for this graph, 1 = passed, 2 = FAILED (the graph was re-exchanged through host memory)
    std::vector<GraphPart> parts;
    bool has_map = false;
    int64_t nnz = 0;
};

}  // extern "C"

namespace {
std::vector<cm_handle_t> handle_devices(cm_handle_t h) {
    std::vector<cm_handle_t> v{h};
    for (auto *p : h->peers) v.push_back(p);
    return v;
}
// f(i) for every part, one host thread per part beyond the first (every HIP call of a part is issued by its own thread,
// which binds that part's device first); the first failure is reported through the caller's handle.
template <class F>
cm_status for_parts(cm_handle_t h, size_t n, const std::vector<cm_handle_t> &hs, F f) {
    if (n <= 1) return f((size_t)0);
    std::vector<cm_status> rc(n, CM_OK);
    std::vector<std::thread> th;
    for (size_t i = 1; i < n; ++i) th.emplace_back([&, i] { rc[i] = f(i); });
    rc[0] = f((size_t)0);
    for (auto &t : th) t.join();
    (void)hipSetDevice(h->device);
    for (size_t i = 0; i < n; ++i)
        if (rc[i] != CM_OK) {
            if (hs[i] != h) h->err = "device " + std::to_string(hs[i]->device) + ": " + hs[i]->err;
            return rc[i];
        }
    return CM_OK;
}
std::vector<cm_handle_t> part_handles(const cm_graph_s *g) {
    std::vector<cm_handle_t> v;
    for (auto &p : g->parts) v.push_back(p.h);
    return v;
}
__global__ void add_offset_i64(int64_t *p, int64_t n, int64_t base) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] += base;
}
// Every device gets the WHOLE mapping matrix (peer copies of the other parts' rows; row pointers re-based): what a
// gene-column-sharded transfer needs (366 MB at 1.5 M x 30, then no further communication).
cm_status ensure_full_matrix(cm_handle_t h, cm_graph_s *g) {
    const size_t P = g->parts.size();
    bool all = true;
    for (auto &p : g->parts) all = all && p.has_full;
    if (all) return CM_OK;
    std::vector<int64_t> base(P + 1, 0);
    for (size_t i = 0; i < P; ++i) base[i + 1] = base[i] + g->parts[i].nnz;
    return for_parts(h, P, part_handles(g), [&](size_t i) -> cm_status {
        GraphPart &p = g->parts[i];
        CM_TRY(begin_call(p.h, 0));
        CM_TRY(p.f_indptr.alloc(p.h, 8 * ((size_t)g->nq + 1)));
        CM_TRY(p.f_cols.alloc(p.h, 4 * (size_t)std::max<int64_t>(base[P], 1)));
        CM_TRY(p.f_vals.alloc(p.h, 4 * (size_t)std::max<int64_t>(base[P], 1)));
        for (size_t r = 0; r < P; ++r) {
            const GraphPart &s = g->parts[r];
            const int64_t n = s.nq + (r + 1 == P ? 1 : 0);   // the last part also brings the end pointer
            CM_HIP(p.h, hipMemcpyPeerAsync(p.f_indptr.as<int64_t>() + s.q_lo, p.h->device, s.indptr.p, s.h->device, 8 * (size_t)n, p.h->stream));
            if (base[r] && n > 0)
                hipLaunchKernelGGL(add_offset_i64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, p.h->stream, p.f_indptr.as<int64_t>() + s.q_lo, n, base[r]);
            if (s.nnz) {
                CM_HIP(p.h, hipMemcpyPeerAsync(p.f_cols.as<int32_t>() + base[r], p.h->device, s.cols.p, s.h->device, 4 * (size_t)s.nnz, p.h->stream));
                CM_HIP(p.h, hipMemcpyPeerAsync(p.f_vals.as<float>() + base[r], p.h->device, s.vals.p, s.h->device, 4 * (size_t)s.nnz, p.h->stream));
            }
        }
        CM_HIP(p.h, hipGetLastError());
        CM_TRY(sync(p.h));
        p.has_full = true;
        return CM_OK;
    });
}
// the part's distances as float64 on its device (float32 graphs: widened into the part's workspace)
cm_status part_dist64(GraphPart &p, const cm_graph_s *g, const double **out) {
    if (!g->dist_is_f32) { *out = p.dist.as<double>(); return CM_OK; }
    const size_t nk = (size_t)p.nq * g->k;
    double *w = (double *)cm_ws_alloc(p.h, 8 * nk);
    if (!w) CM_FAIL(p.h, CM_E_NOMEM, "graph: device workspace exhausted");
    CM_TRY(cm_widen_f32_run(p.h, p.dist.as<float>(), (int64_t)nk, w));
    *out = w;
    return CM_OK;
}
// rows [lo, lo + rows) of a representation on device index i of the handle (0 = the handle's own device, which holds
// everything); a peer gets them from device 0 by ONE peer copy (xGMI where the platform has it) the first time a graph
// needs them, and keeps them for the following graphs (the four-graph mode searches a representation several times)
cm_status rep_ensure(cm_handle_t h, const std::vector<cm_handle_t> &hs, cm_rep_s *r, size_t i, int64_t lo, int64_t rows, const void **out,
                     int64_t *resident_bytes) {
    if (i == 0) { *out = (const char *)r->bufs[0].p + (size_t)lo * r->row_bytes(); *resident_bytes = (int64_t)((size_t)r->n * r->row_bytes()); return CM_OK; }
    cm_handle_t hp = hs[i];
    if (!(r->bufs[i].p && r->lo[i] <= lo && lo + rows <= r->lo[i] + r->rows[i])) {
        CM_HIP(hp, hipSetDevice(hp->device));
        // a range that does not cover the request is replaced by the union with what the device already holds when the two
        // touch (a shard growing into "everything"), otherwise by the request
        int64_t nlo = lo, nhi = lo + rows;
        if (r->bufs[i].p && r->rows[i] > 0) { nlo = std::min(nlo, r->lo[i]); nhi = std::max(nhi, r->lo[i] + r->rows[i]); }
        r->bufs[i].release();
        CM_TRY(r->bufs[i].alloc(hp, (size_t)(nhi - nlo) * r->row_bytes()));
        CM_HIP(hp, hipMemcpyPeerAsync(r->bufs[i].p, hp->device, (const char *)r->bufs[0].p + (size_t)nlo * r->row_bytes(), h->device,
                                      (size_t)(nhi - nlo) * r->row_bytes(), hp->stream));
        CM_TRY(sync(hp));
        r->lo[i] = nlo; r->rows[i] = nhi - nlo;
    }
    *out = (const char *)r->bufs[i].p + (size_t)(lo - r->lo[i]) * r->row_bytes();
    *resident_bytes = (int64_t)((size_t)r->rows[i] * r->row_bytes());
    return CM_OK;
}
}  // namespace

extern "C" {

cm_status cm_create_multi(const int *devices, int n, cm_handle_t *out) {
    if (!out) return CM_E_INVALID;
    *out = nullptr;
    if (!devices || n < 1) { std::lock_guard<std::mutex> g(g_err_mu); g_create_err = "cm_create_multi: empty device list"; return CM_E_INVALID; }
    cm_handle_t h0 = nullptr;
    cm_status rc = cm_create(devices[0], &h0);
    if (rc != CM_OK) return rc;
    for (int i = 1; i < n; ++i) {
        cm_handle_t hp = nullptr;
        rc = cm_create(devices[i], &hp);
        if (rc != CM_OK) { (void)cm_destroy(h0); return rc; }
        h0->peers.push_back(hp);
    }
    // direct peer copies (xGMI) where the platform allows them; hipMemcpyPeerAsync works (staged) without
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            int can = 0;
            if (devices[i] == devices[j] || hipDeviceCanAccessPeer(&can, devices[i], devices[j]) != hipSuccess || !can) continue;
            if (hipSetDevice(devices[i]) == hipSuccess) (void)hipDeviceEnablePeerAccess(devices[j], 0);
        }
    (void)hipGetLastError();   // ("peer access already enabled" is not an error)
    (void)hipSetDevice(devices[0]);
    *out = h0;
    return CM_OK;
}

cm_status cm_rep_create(cm_handle_t h, const void *X, int64_t n, int32_t d, int32_t dtype, cm_rep_t *out) {
    return cm_rep_create_ex(h, X, n, d, dtype, 0, out);
}

cm_status cm_rep_create_ex(cm_handle_t h, const void *X, int64_t n, int32_t d, int32_t dtype, int32_t flags, cm_rep_t *out) {
    if (!h) return CM_E_INVALID;
    if (!out || !X || n < 1 || d < 1 || (dtype != CM_F32 && dtype != CM_F64)) CM_FAIL(h, CM_E_INVALID, "rep: bad arguments");
    cm_rep_s *r = new (std::nothrow) cm_rep_s();
    if (!r) CM_FAIL(h, CM_E_NOMEM, "rep: out of host memory");
    r->n = n; r->d = d; r->dtype = dtype;
    const size_t ndev = 1 + h->peers.size();
    r->bufs = std::vector<DevBuf>(ndev);
    r->lo.assign(ndev, 0); r->rows.assign(ndev, 0);
    const size_t bytes = (size_t)n * r->row_bytes();
    // ONE upload, to the handle's own device; peers get what a graph needs from there (rep_ensure)
    cm_status rc = [&]() -> cm_status {
        CM_TRY(begin_call(h, 0));
        CM_TRY(r->bufs[0].alloc(h, bytes));
        CM_TRY(h2d(h, r->bufs[0].p, X, bytes));
        // NaN / infinity screen on the device copy (the host-side check it replaces cost more than the upload)
        int *d_flag = (int *)cm_ws_alloc(h, 256);
        if (!d_flag) CM_FAIL(h, CM_E_NOMEM, "rep: device workspace exhausted");
        h->d_nflag_pinned[12] = 0;
        CM_TRY(nonfinite_enqueue(h, r->bufs[0].p, bytes, dtype, d_flag, 12));
        CM_TRY(sync(h));
        const int verdict = h->d_nflag_pinned[12];
        if (flags & CM_REP_ALLOW_NAN) {
            if (verdict & 1) CM_FAIL(h, CM_E_INVALID, CM_INF_MSG);
            r->has_nan = (verdict & 2) != 0;
        } else if (verdict) CM_FAIL(h, CM_E_INVALID, CM_NONFINITE_MSG);
        return CM_OK;
    }();
    if (rc != CM_OK) {
        r->bufs[0].release();
        delete r;
        return rc;
    }
    r->rows[0] = n;
    *out = r;
    return CM_OK;
}

cm_status cm_rep_destroy(cm_handle_t h, cm_rep_t r) {
    if (!h) return CM_E_INVALID;
    if (r)
        for (auto &b : r->bufs) { if (b.owner) (void)hipSetDevice(b.owner->device); b.release(); }
    (void)hipSetDevice(h->device);
    delete r;
    return CM_OK;
}

// one device's search of the queries [0, nq) against reference rows [r_lo, r_lo + n_rows): d_refrows points at row r_lo on
// the device, d_qry at query 0
// one device's search of `nq` queries (d_qry points at the first of them) against reference rows [r_lo, r_lo + n_rows)
static cm_status part_knn(cm_handle_t hp, const cm_rep_s *ref, const void *d_refrows, const void *d_qry, int64_t nq, int64_t r_lo,
                          int64_t n_rows, int32_t k, int32_t metric, int32_t form, int32_t flags, void *d_dist, size_t des, int64_t *d_idx,
                          int64_t *nflag_out) {
    const int64_t QCH = 1 << 21;   // queries in chunks of 2^21 rows like the host tier (bounds the workspace)
    const size_t es = ref->dtype == CM_F64 ? 8 : 4;
    int64_t nflag = 0;
    for (int64_t q0 = 0; q0 < nq; q0 += QCH) {
        const int64_t nqc = std::min<int64_t>(QCH, nq - q0);
        CM_TRY(begin_call(hp, cm_knn_ws_bytes(n_rows, nqc, ref->d, k, ref->dtype, metric)));
        CM_TRY(cm_knn_run(hp, d_refrows, n_rows, (const char *)d_qry + es * q0 * ref->d, nqc, ref->d,
                          ref->dtype, k, metric, form, r_lo, flags, (char *)d_dist + des * q0 * k, d_idx + q0 * k));
        CM_TRY(sync(hp));
        CM_TRY(check_fallback_overflow(hp));
        nflag += hp->knn_stats[0];
    }
    *nflag_out = nflag;
    return CM_OK;
}

namespace {
// Exchange + merge of one part (query slice p.q_lo .. + p.nq) of a multi-device graph: the slice's two planes of every
// device's packed buffer are brought to the part's device -- by peer copies (route 0: xGMI where peer access exists), or
// staged through host memory (route 1: platforms without peer access, and the fall-back of the self-check) -- and merged
// by (value, global index), the rule of a single search.
cm_status part_exchange_merge(cm_handle_t hp_i, const std::vector<cm_handle_t> &hp, const std::vector<DevBuf> &packed, size_t nk, int route,
                              const cm_graph_s *g, int64_t q_lo, int64_t nq_p, double *d_out_dist, int64_t *d_out_idx) {
    const size_t P = hp.size(), pk = (size_t)nq_p * g->k;
    CM_TRY(begin_call(hp_i, 0));
    DevBuf recv;
    CM_TRY(recv.alloc(hp_i, 16 * pk * P));
    std::vector<char> stage;
    if (route == 1) stage.resize(16 * pk);
    for (size_t r = 0; r < P; ++r) {
        char *dst = (char *)recv.p + 16 * pk * r;
        const char *src = (const char *)packed[r].p + 8 * (size_t)q_lo * g->k;
        if (route == 0) {
            CM_HIP(hp_i, hipMemcpyPeerAsync(dst, hp_i->device, src, hp[r]->device, 8 * pk, hp_i->stream));
            CM_HIP(hp_i, hipMemcpyPeerAsync(dst + 8 * pk, hp_i->device, src + 8 * nk, hp[r]->device, 8 * pk, hp_i->stream));
        } else {
            // (plain synchronous copies on the owning devices: this route is the exception)
            CM_HIP(hp_i, hipSetDevice(hp[r]->device));
            CM_HIP(hp_i, hipMemcpy(stage.data(), src, 8 * pk, hipMemcpyDeviceToHost));
            CM_HIP(hp_i, hipMemcpy(stage.data() + 8 * pk, src + 8 * nk, 8 * pk, hipMemcpyDeviceToHost));
            CM_HIP(hp_i, hipSetDevice(hp_i->device));
            CM_HIP(hp_i, hipMemcpy(dst, stage.data(), 16 * pk, hipMemcpyHostToDevice));
        }
    }
    CM_TRY(cm_knn_merge_run(hp_i, recv.as<double>(), recv.as<int64_t>() + pk, (int)P, (int64_t)(2 * pk), nq_p, g->k,
                            g->metric == CM_METRIC_EUCLIDEAN ? 1 : 0, d_out_dist, d_out_idx));
    return sync(hp_i);
}
}  // namespace

cm_status cm_graph_knn(cm_handle_t h, cm_rep_t ref, cm_rep_t qry, int32_t k, int32_t metric, int32_t form, cm_graph_t *out) {
    if (!h) return CM_E_INVALID;
    if (!ref || !qry || !out) CM_FAIL(h, CM_E_INVALID, "graph: bad arguments");
    if (ref->d != qry->d || ref->dtype != qry->dtype) CM_FAIL(h, CM_E_INVALID, "graph: the two representations differ in width or dtype");
    if ((ref->has_nan || qry->has_nan) && metric != CM_METRIC_NAN_EUCLIDEAN) CM_FAIL(h, CM_E_INVALID, CM_NONFINITE_MSG);
    if (k < 1 || k > ref->n)
        CM_FAIL(h, CM_E_INVALID, "Expected n_neighbors <= n_samples_fit, but n_neighbors = %d, n_samples_fit = %lld", k, (long long)ref->n);
    cm_graph_s *g = new (std::nothrow) cm_graph_s();
    if (!g) CM_FAIL(h, CM_E_NOMEM, "graph: out of host memory");
    g->nq = qry->n; g->n_ref = ref->n; g->k = k; g->metric = metric;
    g->dist_is_f32 = (metric == CM_METRIC_COSINE && ref->dtype == CM_F32) ? 1 : 0;
    const auto hs = handle_devices(h);
    size_t P = std::min(hs.size(), std::min(ref->bufs.size(), qry->bufs.size()));
    // Round 5: a multi-device graph splits the QUERIES when the reference fits a device (32 GB of it; CELLMAP_MULTI_REF_SHARD
    // forces the reference-sharded choreography below): every device gets the whole reference and its query slice by one peer
    // copy each, searches on its own, and owns that slice of the graph from then on -- no exchange, no merge, no limit from the
    // merge's list capacity, and float32 cosine keeps the one-device arithmetic per query by construction.  Measured shares of
    // one rank (profiles/r05_rank_shares.md): the search costs a fixed amount per (query, shard), so at 8 devices query slices
    // finish in 37 ms where reference shards take 65.
    if (P > 1 && !cm_get_knobs().multi_ref_shard && (size_t)ref->n * ref->row_bytes() <= ((size_t)32 << 30)) {
        size_t Pq = P;
        while (Pq > 1 && qry->n < (int64_t)Pq * 512) --Pq;   // (a slice should fill at least one 512-query block)
        if (Pq > 1) {
            g->parts.resize(Pq);
            std::vector<int64_t> nflag(Pq, 0);
            const std::vector<cm_handle_t> hp(hs.begin(), hs.begin() + Pq);
            const size_t des = g->dist_is_f32 ? 4 : 8;
            cm_status rc = for_parts(h, Pq, hp, [&](size_t i) -> cm_status {
                GraphPart &p = g->parts[i];
                const int64_t base = g->nq / (int64_t)Pq, extra = g->nq % (int64_t)Pq;
                p.h = hp[i];
                p.q_lo = (int64_t)i * base + std::min<int64_t>((int64_t)i, extra);
                p.nq = base + ((int64_t)i < extra ? 1 : 0);
                p.ref_lo = 0; p.ref_rows = ref->n;
                CM_TRY(begin_call(hp[i], 0));
                const void *d_ref = nullptr, *d_q = nullptr;
                CM_TRY(rep_ensure(h, hs, ref, i, 0, ref->n, &d_ref, &p.ref_bytes));
                if (ref == qry) { d_q = (const char *)d_ref + (size_t)p.q_lo * ref->row_bytes(); p.qry_bytes = p.ref_bytes; }
                else CM_TRY(rep_ensure(h, hs, qry, i, p.q_lo, p.nq, &d_q, &p.qry_bytes));
                CM_TRY(p.dist.alloc(hp[i], des * (size_t)p.nq * k));
                CM_TRY(p.idx.alloc(hp[i], 8 * (size_t)p.nq * k));
                return part_knn(hp[i], ref, d_ref, d_q, p.nq, 0, ref->n, k, metric, form, 0, p.dist.p, des, p.idx.as<int64_t>(), &nflag[i]);
            });
            (void)hipSetDevice(h->device);
            int64_t tot = 0;
            for (auto v : nflag) tot += v;
            h->knn_stats[0] = tot;
            g->exchange_route = 2;   // none: query slices
            if (rc != CM_OK) { (void)cm_graph_destroy(h, g); return rc; }
            *out = g;
            return CM_OK;
        }
    }
    // reference shards need room for k neighbours and enough rows to be worth a launch; float32 cosine keeps the one-device
    // arithmetic (sklearn's float32 distances)
    // ... and the merge ranks at most CM_MERGE_NCMAX = parts x k entries per query: a large k uses fewer devices (k = 97 on an
    // 8-GPU node: 7; k > 384: one) instead of failing after every shard has been searched
    P = std::min<size_t>(P, (size_t)std::max(1, CM_MERGE_NCMAX / k));
    if (g->dist_is_f32 || ref->n < (int64_t)P * 4096 || ref->n < (int64_t)P * k || qry->n < 1024) P = 1;
    cm_status rc = CM_OK;
    if (P == 1) {
        g->parts.resize(1);
        GraphPart &p = g->parts[0];
        p.h = h; p.q_lo = 0; p.nq = g->nq;
        p.ref_lo = 0; p.ref_rows = ref->n;
        p.ref_bytes = (int64_t)((size_t)ref->n * ref->row_bytes()); p.qry_bytes = (int64_t)((size_t)qry->n * qry->row_bytes());
        const size_t nk = (size_t)g->nq * k, des = g->dist_is_f32 ? 4 : 8;
        rc = p.dist.alloc(h, des * nk);
        if (rc == CM_OK) rc = p.idx.alloc(h, 8 * nk);
        int64_t nflag = 0;
        if (rc == CM_OK) rc = part_knn(h, ref, ref->bufs[0].p, qry->bufs[0].p, qry->n, 0, ref->n, k, metric, form, 0, p.dist.p, des, p.idx.as<int64_t>(), &nflag);
        h->knn_stats[0] = nflag;
    } else {
        // 1. every device: all queries against its reference shard, GLOBAL indices, float64 ranking values (squared for
        //    euclidean) into a packed buffer [2][nq][k] of 8-byte words -- the layout of the RCCL path's exchange buffer.
        //    Device i holds reference rows [lo_i, hi_i) only (+ every query): rep_ensure
        g->parts.resize(P);
        std::vector<DevBuf> packed(P);
        std::vector<int64_t> nflag(P, 0);
        const size_t nk = (size_t)g->nq * k;
        const std::vector<cm_handle_t> hp(hs.begin(), hs.begin() + P);
        rc = for_parts(h, P, hp, [&](size_t i) -> cm_status {
            GraphPart &p = g->parts[i];
            const int64_t base = ref->n / (int64_t)P, extra = ref->n % (int64_t)P;
            const int64_t lo = (int64_t)i * base + std::min<int64_t>((int64_t)i, extra), rows = base + ((int64_t)i < extra ? 1 : 0);
            CM_TRY(begin_call(hp[i], 0));
            const void *d_refrows = nullptr, *d_q = nullptr;
            // (a representation searched against itself -- the reference's xx / yy graphs -- needs every row on every device)
            if (ref == qry) {
                CM_TRY(rep_ensure(h, hs, qry, i, 0, qry->n, &d_q, &p.qry_bytes));
                d_refrows = (const char *)d_q + (size_t)lo * ref->row_bytes();
                p.ref_bytes = p.qry_bytes;
            } else {
                CM_TRY(rep_ensure(h, hs, ref, i, lo, rows, &d_refrows, &p.ref_bytes));
                CM_TRY(rep_ensure(h, hs, qry, i, 0, qry->n, &d_q, &p.qry_bytes));
            }
            p.ref_lo = lo; p.ref_rows = rows;
            CM_TRY(packed[i].alloc(hp[i], 16 * nk));
            return part_knn(hp[i], ref, d_refrows, d_q, qry->n, lo, rows, k, metric, form, CM_KNN_SQUARED | CM_KNN_F64_OUT,
                            packed[i].p, 8, packed[i].as<int64_t>() + nk, &nflag[i]);
        });
        // 2. + 3. device p owns query slice p: it brings that slice of every device's two planes over (peer copies, or
        //    staged through the host where the platform has no peer access) and merges the P lists per query by
        //    (value, global index) -- the rule of a single search, so the result is exact
        int route = (cm_get_knobs().multi_staged || h->multi_route_staged) ? 1 : 0;   // (an earlier graph's self-check failed: the handle stays on the host-staged route)
        for (size_t a = 0; a < P && !route; ++a)
            for (size_t b = 0; b < P && !route; ++b) {
                int can = 1;
                if (hp[a]->device != hp[b]->device && (hipDeviceCanAccessPeer(&can, hp[a]->device, hp[b]->device) != hipSuccess || !can)) route = 1;
            }
        (void)hipGetLastError();
        auto exchange_all = [&](int rt) {
            return for_parts(h, P, hp, [&](size_t i) -> cm_status {
                GraphPart &p = g->parts[i];
                const int64_t base = g->nq / (int64_t)P, extra = g->nq % (int64_t)P;
                p.h = hp[i];
                p.q_lo = (int64_t)i * base + std::min<int64_t>((int64_t)i, extra);
                p.nq = base + ((int64_t)i < extra ? 1 : 0);
                const size_t pk = (size_t)p.nq * k;
                CM_TRY(begin_call(hp[i], 0));
                CM_TRY(p.dist.alloc(hp[i], 8 * pk));
                CM_TRY(p.idx.alloc(hp[i], 8 * pk));
                return part_exchange_merge(hp[i], hp, packed, nk, rt, g, p.q_lo, p.nq, p.dist.as<double>(), p.idx.as<int64_t>());
            });
        };
        if (rc == CM_OK) rc = exchange_all(route);
        // Self-check of the peer-copy route, once per handle (the first graph that shards): query slice 0 is exchanged a
        // second time through host memory and merged again on devices[0]; the two results must agree bit for bit.  A
        // mismatch (or the test knob) re-runs the whole exchange through the host and keeps the handle on that route.
        if (rc == CM_OK && route == 0 && !h->multi_checked) {
            GraphPart &p0 = g->parts[0];
            const size_t pk = (size_t)p0.nq * k;
            DevBuf cd, ci;
            std::vector<char> a1(16 * pk), a2(16 * pk);
            rc = begin_call(h, 0);
            if (rc == CM_OK) rc = cd.alloc(h, 8 * pk);
            if (rc == CM_OK) rc = ci.alloc(h, 8 * pk);
            if (rc == CM_OK) rc = part_exchange_merge(h, hp, packed, nk, 1, g, p0.q_lo, p0.nq, cd.as<double>(), ci.as<int64_t>());
            if (rc == CM_OK) {
                rc = d2h(h, a1.data(), p0.dist.p, 8 * pk);
                if (rc == CM_OK) rc = d2h(h, a1.data() + 8 * pk, p0.idx.p, 8 * pk);
                if (rc == CM_OK) rc = d2h(h, a2.data(), cd.p, 8 * pk);
                if (rc == CM_OK) rc = d2h(h, a2.data() + 8 * pk, ci.p, 8 * pk);
                if (rc == CM_OK) rc = sync(h);
            }
            if (rc == CM_OK) {
                const bool same = memcmp(a1.data(), a2.data(), 16 * pk) == 0 && !cm_get_knobs().multi_check_fail;
                h->multi_checked = true;
                g->self_check = same ? 1 : 2;
                if (!same) {
                    h->multi_route_staged = true;
                    route = 1;
                    rc = exchange_all(1);
                }
            }
        }
        g->exchange_route = route;
        for (size_t i = 0; i < P; ++i) { (void)hipSetDevice(hp[i]->device); packed[i].release(); }
        (void)hipSetDevice(h->device);
        int64_t tot = 0;
        for (auto v : nflag) tot += v;
        h->knn_stats[0] = tot;
    }
    if (rc != CM_OK) { (void)cm_graph_destroy(h, g); return rc; }
    *out = g;
    return CM_OK;
}

cm_status cm_graph_destroy(cm_handle_t h, cm_graph_t g) {
    if (!h) return CM_E_INVALID;
    if (g)
        for (auto &p : g->parts) {
            if (!p.h) continue;
            (void)hipSetDevice(p.h->device);
            if (p.h->a3_session && static_cast<A3Session *>(p.h->a3_session)->mp == p.indptr.as<int64_t>()) a3_session_drop(p.h);
            p.dist.release(); p.idx.release(); p.indptr.release(); p.cols.release(); p.vals.release();
            p.f_indptr.release(); p.f_cols.release(); p.f_vals.release();
        }
    (void)hipSetDevice(h->device);
    delete g;
    return CM_OK;
}

cm_status cm_graph_knn_fetch(cm_handle_t h, cm_graph_t g, void *out_dist, int64_t *out_idx) {
    if (!h) return CM_E_INVALID;
    if (!g || !out_dist || !out_idx) CM_FAIL(h, CM_E_INVALID, "graph: bad arguments");
    const size_t des = g->dist_is_f32 ? 4 : 8;
    return for_parts(h, g->parts.size(), part_handles(g), [&](size_t i) -> cm_status {
        GraphPart &p = g->parts[i];
        CM_TRY(begin_call(p.h, 0));
        const size_t nk = (size_t)p.nq * g->k, off = (size_t)p.q_lo * g->k;
        CM_TRY(d2h(p.h, (char *)out_dist + des * off, p.dist.p, des * nk));
        CM_TRY(d2h(p.h, out_idx + off, p.idx.p, 8 * nk));
        return sync(p.h);
    });
}

cm_status cm_graph_mapping(cm_handle_t h, cm_graph_t g, int32_t kernel, double epsilon, int64_t *out_nnz, double *out_param) {
    if (!h) return CM_E_INVALID;
    if (!g) CM_FAIL(h, CM_E_INVALID, "graph: bad arguments");
    if (kernel < 0 || kernel > CM_KERNEL_IDENTITY) CM_FAIL(h, CM_E_INVALID, "Unknown kernel id %d", kernel);
    g->has_map = false;
    for (auto &p : g->parts) {  // a pending sparse-product session holds the OLD mapping-matrix arrays of this graph
        if (p.h->a3_session) { (void)hipSetDevice(p.h->device); a3_session_drop(p.h); }
        p.has_full = false;
    }
    (void)hipSetDevice(h->device);
    const size_t P = g->parts.size();
    const auto hs = part_handles(g);
    auto ws_need = [&](const GraphPart &p) {
        const size_t nk = (size_t)p.nq * g->k;
        return 64 * 1024 + cm_scan_ws_bytes(p.nq) + cm_align(8 * nk) + cm_align(4 * nk) * 2 + cm_align(4 * (size_t)p.nq);
    };
    // the GLOBAL bandwidth (knn.py:194-209): moments of every part, summed on the host in part order, handed back to every
    // device, which then computes the parameter with the kernel a single device uses
    std::vector<std::array<double, 3>> mom(P);
    std::vector<double *> d_m3(P, nullptr);
    std::vector<const double *> dd(P, nullptr);
    auto moments_pass = [&](bool centred) {
        return for_parts(h, P, hs, [&](size_t i) -> cm_status {
            GraphPart &p = g->parts[i];
            if (!centred) {
                CM_TRY(begin_call(p.h, ws_need(p)));
                CM_TRY(part_dist64(p, g, &dd[i]));
                d_m3[i] = (double *)cm_ws_alloc(p.h, 64);
                if (!d_m3[i]) CM_FAIL(p.h, CM_E_NOMEM, "graph: device workspace exhausted");
            } else {
                CM_HIP(p.h, hipSetDevice(p.h->device));
            }
            CM_HIP(p.h, hipMemsetAsync(d_m3[i], 0, 24, p.h->stream));
            CM_TRY(cm_moments_run(p.h, dd[i], p.idx.as<int64_t>(), (int64_t)p.nq * g->k, centred ? d_m3[i] + 5 : nullptr, d_m3[i]));
            CM_TRY(d2h(p.h, mom[i].data(), d_m3[i], 24));
            return sync(p.h);
        });
    };
    auto param_pass = [&](int which) {   // total moments -> every device -> param (d_m3 + 5)
        double tot[3] = {0, 0, 0};
        for (size_t i = 0; i < P; ++i) for (int j = 0; j < 3; ++j) tot[j] += mom[i][j];
        return for_parts(h, P, hs, [&](size_t i) -> cm_status {
            GraphPart &p = g->parts[i];
            CM_HIP(p.h, hipSetDevice(p.h->device));
            CM_TRY(h2d(p.h, d_m3[i], tot, 24));
            CM_TRY(cm_param_run(p.h, d_m3[i], which, d_m3[i] + 5));
            return sync(p.h);   // (tot lives on this stack frame)
        });
    };
    CM_TRY(moments_pass(false));
    double cnt = 0;
    for (size_t i = 0; i < P; ++i) cnt += mom[i][2];
    if (!(cnt > 0.0)) CM_FAIL(h, CM_E_NOFINITE, "No finite distances found in the neighborhood graph");
    if (kernel == CM_KERNEL_GAUSSIAN || kernel == CM_KERNEL_SCARCHES) {
        CM_TRY(param_pass(CM_KERNEL_GAUSSIAN));   // = mean
        if (kernel == CM_KERNEL_SCARCHES) {
            CM_TRY(moments_pass(true));
            CM_TRY(param_pass(CM_KERNEL_SCARCHES));
        }
    }
    std::vector<double> hp_(P, 0.0);
    CM_TRY(for_parts(h, P, hs, [&](size_t i) -> cm_status {
        GraphPart &p = g->parts[i];
        CM_HIP(p.h, hipSetDevice(p.h->device));
        const size_t nk = (size_t)p.nq * g->k;
        int32_t *ecols = (int32_t *)cm_ws_alloc(p.h, 4 * nk);
        float *evals = (float *)cm_ws_alloc(p.h, 4 * nk);
        int32_t *ecnt = (int32_t *)cm_ws_alloc(p.h, 4 * (size_t)p.nq);
        if (!ecols || !evals || !ecnt) CM_FAIL(p.h, CM_E_NOMEM, "graph: device workspace exhausted");
        double *param = d_m3[i] + 5;
        CM_TRY(cm_weights_run(p.h, dd[i], p.idx.as<int64_t>(), p.nq, g->k, g->n_ref, kernel, param, epsilon, ecols, evals, ecnt, nullptr));
        CM_TRY(p.indptr.alloc(p.h, 8 * ((size_t)p.nq + 1)));
        CM_TRY(p.cols.alloc(p.h, 4 * nk));
        CM_TRY(p.vals.alloc(p.h, 4 * nk));
        CM_TRY(cm_ell_to_csr_run(p.h, ecols, evals, nullptr, ecnt, p.nq, g->k, p.indptr.as<int64_t>(), p.cols.as<int32_t>(), p.vals.as<float>(),
                                 nullptr));
        CM_TRY(d2h(p.h, &p.nnz, p.indptr.as<int64_t>() + p.nq, 8));
        CM_TRY(d2h(p.h, &hp_[i], param, 8));
        return sync(p.h);
    }));
    int64_t nnz = 0;
    for (auto &p : g->parts) nnz += p.nnz;
    g->nnz = nnz;
    g->has_map = true;
    if (out_nnz) *out_nnz = nnz;
    if (out_param) *out_param = (kernel == CM_KERNEL_GAUSSIAN || kernel == CM_KERNEL_SCARCHES) ? hp_[0] : 0.0;
    return CM_OK;
}

#define CM_NEED_MAP(h, g)                                                                                         \
    if (!g || !g->has_map) CM_FAIL(h, CM_E_INVALID, "graph: the mapping matrix has not been computed (cm_graph_mapping)")

cm_status cm_graph_mapping_fetch(cm_handle_t h, cm_graph_t g, int64_t *out_indptr, int32_t *out_cols, float *out_vals) {
    if (!h) return CM_E_INVALID;
    CM_NEED_MAP(h, g);
    std::vector<int64_t> base(g->parts.size() + 1, 0);
    for (size_t i = 0; i < g->parts.size(); ++i) base[i + 1] = base[i] + g->parts[i].nnz;
    CM_TRY(for_parts(h, g->parts.size(), part_handles(g), [&](size_t i) -> cm_status {
        GraphPart &p = g->parts[i];
        CM_TRY(begin_call(p.h, 0));
        // (the row pointer of a part starts at 0; entry nq of part i and entry 0 of part i + 1 are the same output slot:
        // every part writes its rows' START offsets only, the last one also the end)
        CM_TRY(d2h(p.h, out_indptr + p.q_lo, p.indptr.p, 8 * ((size_t)p.nq + (i + 1 == g->parts.size() ? 1 : 0))));
        CM_TRY(d2h(p.h, out_cols + base[i], p.cols.p, 4 * (size_t)p.nnz));
        CM_TRY(d2h(p.h, out_vals + base[i], p.vals.p, 4 * (size_t)p.nnz));
        return sync(p.h);
    }));
    for (size_t i = 1; i < g->parts.size(); ++i) {
        const GraphPart &p = g->parts[i];
        const int64_t n = p.nq + (i + 1 == g->parts.size() ? 1 : 0);
        for (int64_t r = 0; r < n; ++r) out_indptr[p.q_lo + r] += base[i];
    }
    return CM_OK;
}

cm_status cm_graph_label_vote(cm_handle_t h, cm_graph_t g, const int32_t *codes, int64_t n_ref, int32_t n_cat, int32_t *out_code,
                              float *out_conf) {
    if (!h) return CM_E_INVALID;
    CM_NEED_MAP(h, g);
    if (n_ref != g->n_ref) CM_FAIL(h, CM_E_INVALID, "graph: %lld label codes for %lld reference cells", (long long)n_ref, (long long)g->n_ref);
    return for_parts(h, g->parts.size(), part_handles(g), [&](size_t i) -> cm_status {   // label codes replicated, votes by query slice
        GraphPart &p = g->parts[i];
        CM_TRY(begin_call(p.h, 0));
        DevBuf dcodes, ocode, oconf;
        CM_TRY(dcodes.alloc(p.h, 4 * (size_t)n_ref)); CM_TRY(ocode.alloc(p.h, 4 * (size_t)p.nq)); CM_TRY(oconf.alloc(p.h, 4 * (size_t)p.nq));
        CM_TRY(h2d(p.h, dcodes.p, codes, 4 * (size_t)n_ref));
        CM_TRY(cm_vote_run(p.h, p.indptr.as<int64_t>(), p.cols.as<int32_t>(), p.vals.as<float>(), p.nq, g->k, nullptr, dcodes.as<int32_t>(), n_cat,
                           ocode.as<int32_t>(), oconf.as<float>()));
        CM_TRY(d2h(p.h, out_code + p.q_lo, ocode.p, 4 * (size_t)p.nq));
        CM_TRY(d2h(p.h, out_conf + p.q_lo, oconf.p, 4 * (size_t)p.nq));
        return sync(p.h);
    });
}

cm_status cm_graph_spmm_dense(cm_handle_t h, cm_graph_t g, const void *B, int64_t n_ref, int64_t n_out, int32_t b_dtype, void *out) {
    if (!h) return CM_E_INVALID;
    CM_NEED_MAP(h, g);
    if (n_ref != g->n_ref) CM_FAIL(h, CM_E_INVALID, "graph: payload with %lld rows for %lld reference cells", (long long)n_ref, (long long)g->n_ref);
    const size_t es = b_dtype == CM_F64 ? 8 : 4;
    const size_t P = g->parts.size();
    if (P > 1 && n_out >= (int64_t)(64 * P)) {
        // wide payload (a layer): GENE COLUMNS sharded over the devices -- each uploads only its slab (the PCIe copy of the
        // payload is what a host-tier layer transfer waits for) and multiplies ALL query rows with it; the mapping matrix is
        // replicated once per graph by peer copies.  A column's sum is the same sequence of multiply-adds on any device.
        CM_TRY(ensure_full_matrix(h, g));
        return for_parts(h, P, part_handles(g), [&](size_t i) -> cm_status {
            GraphPart &p = g->parts[i];
            const int64_t cb = n_out / (int64_t)P, ce = n_out % (int64_t)P;
            const int64_t c_lo = (int64_t)i * cb + std::min<int64_t>((int64_t)i, ce), c_hi = c_lo + cb + ((int64_t)i < ce ? 1 : 0);
            CM_TRY(begin_call(p.h, 0));
            return spmm_dense_stream(p.h, p.f_indptr.as<int64_t>(), p.f_cols.as<int32_t>(), p.f_vals.as<float>(), g->nq, B, n_ref, n_out, b_dtype,
                                     out, c_lo, c_hi);
        });
    }
    return for_parts(h, P, part_handles(g), [&](size_t i) -> cm_status {   // narrow payload: replicated, rows by slice
        GraphPart &p = g->parts[i];
        CM_TRY(begin_call(p.h, 0));
        return spmm_dense_stream(p.h, p.indptr.as<int64_t>(), p.cols.as<int32_t>(), p.vals.as<float>(), p.nq, B, n_ref, n_out, b_dtype,
                                 (char *)out + es * (size_t)p.q_lo * (size_t)n_out);
    });
}

cm_status cm_graph_spmm_csr_nnz(cm_handle_t h, cm_graph_t g, const int64_t *b_indptr, const int32_t *b_cols, const float *b_vals,
                                int64_t n_ref, int64_t n_out, int64_t *out_indptr) {
    if (!h) return CM_E_INVALID;
    CM_NEED_MAP(h, g);
    const size_t P = g->parts.size();
    if (P == 1) {
        GraphPart &p = g->parts[0];
        return spmm_csr_nnz_impl(h, nullptr, nullptr, nullptr, p.indptr.as<int64_t>(), p.cols.as<int32_t>(), p.vals.as<float>(), g->nq, b_indptr,
                                 b_cols, b_vals, n_ref, n_out, out_indptr);
    }
    if (!out_indptr) CM_FAIL(h, CM_E_INVALID, "spmm_csr: bad arguments");
    std::vector<std::vector<int64_t>> loc(P);
    CM_TRY(for_parts(h, P, part_handles(g), [&](size_t i) -> cm_status {
        GraphPart &p = g->parts[i];
        loc[i].resize((size_t)p.nq + 1);
        return spmm_csr_nnz_impl(p.h, nullptr, nullptr, nullptr, p.indptr.as<int64_t>(), p.cols.as<int32_t>(), p.vals.as<float>(), p.nq, b_indptr,
                                 b_cols, b_vals, n_ref, n_out, loc[i].data());
    }));
    int64_t base = 0;
    for (size_t i = 0; i < P; ++i) {
        const GraphPart &p = g->parts[i];
        for (int64_t r = 0; r < p.nq; ++r) out_indptr[p.q_lo + r] = base + loc[i][(size_t)r];
        base += loc[i][(size_t)p.nq];
    }
    out_indptr[g->nq] = base;
    return CM_OK;
}

cm_status cm_graph_spmm_csr(cm_handle_t h, cm_graph_t g, const int64_t *b_indptr, const int32_t *b_cols, const float *b_vals,
                            int64_t n_ref, int64_t n_out, const int64_t *out_indptr, int32_t *out_cols, float *out_vals) {
    if (!h) return CM_E_INVALID;
    CM_NEED_MAP(h, g);
    const size_t P = g->parts.size();
    if (P == 1) {
        GraphPart &p = g->parts[0];
        return spmm_csr_fill_impl(h, nullptr, nullptr, nullptr, p.indptr.as<int64_t>(), p.cols.as<int32_t>(), p.vals.as<float>(), g->nq, b_indptr,
                                  b_cols, b_vals, n_ref, n_out, out_indptr, out_cols, out_vals);
    }
    if (!out_indptr) CM_FAIL(h, CM_E_INVALID, "spmm_csr: bad arguments");
    return for_parts(h, P, part_handles(g), [&](size_t i) -> cm_status {
        GraphPart &p = g->parts[i];
        const int64_t base = out_indptr[p.q_lo];
        std::vector<int64_t> loc((size_t)p.nq + 1);   // the part's row pointer, relative to its first entry
        for (int64_t r = 0; r <= p.nq; ++r) loc[(size_t)r] = out_indptr[p.q_lo + r] - base;
        return spmm_csr_fill_impl(p.h, nullptr, nullptr, nullptr, p.indptr.as<int64_t>(), p.cols.as<int32_t>(), p.vals.as<float>(), p.nq, b_indptr,
                                  b_cols, b_vals, n_ref, n_out, loc.data(), out_cols + base, out_vals + base);
    });
}

cm_status cm_graph_part_info(cm_handle_t h, cm_graph_t g, int32_t part, int64_t out[8]) {
    if (!h) return CM_E_INVALID;
    if (!g || !out || part < 0 || (size_t)part >= g->parts.size()) CM_FAIL(h, CM_E_INVALID, "graph: bad arguments");
    const GraphPart &p = g->parts[(size_t)part];
    out[0] = p.h ? p.h->device : -1; out[1] = p.q_lo; out[2] = p.nq; out[3] = p.ref_lo; out[4] = p.ref_rows;
    out[5] = p.ref_bytes; out[6] = p.qry_bytes;
    out[7] = g->exchange_route + 10 * g->self_check;   // route (0 peer copies, 1 host-staged, 2 none: the graph is split by QUERY slices) + 10 x self-check (0 not run, 1 passed, 2 failed)
    return CM_OK;
}

cm_status cm_graph_info(cm_handle_t h, cm_graph_t g, int64_t out[8]) {
    if (!h) return CM_E_INVALID;
    if (!g || !out) CM_FAIL(h, CM_E_INVALID, "graph: bad arguments");
    out[0] = g->nq; out[1] = g->n_ref; out[2] = g->k; out[3] = g->metric; out[4] = g->dist_is_f32; out[5] = g->has_map ? 1 : 0;
    out[6] = g->nnz; out[7] = (int64_t)g->parts.size();   // [7]: devices the graph is spread over
    return CM_OK;
}

}  // extern "C"

// ---------------------------------------------------------------- precomputed-distance ingest ----------
cm_status cm_ingest_distances(cm_handle_t h, const int64_t *indptr, const int32_t *indices, const double *data, int64_t n,
                              int32_t include_self, int32_t kmax, int64_t *out_idx, double *out_dist, int32_t *out_kmax) {
    if (!h) return CM_E_INVALID;
    if (n < 0 || include_self < -1 || include_self > 1 || !indptr || !out_kmax) CM_FAIL(h, CM_E_INVALID, "precomputed distances: bad arguments");
    CM_TRY(begin_call(h, 0));
    const int64_t nnz = n > 0 ? indptr[n] : 0;
    DevBuf dp, dc, dv, dl, dk, oi, od;
    CM_TRY(dp.alloc(h, 8 * ((size_t)n + 1))); CM_TRY(dc.alloc(h, 4 * (size_t)nnz)); CM_TRY(dl.alloc(h, 4 * (size_t)n)); CM_TRY(dk.alloc(h, 16));
    CM_TRY(h2d(h, dp.p, indptr, 8 * ((size_t)n + 1)));
    CM_TRY(h2d(h, dc.p, indices, 4 * (size_t)nnz));
    CM_HIP(h, hipMemsetAsync(dk.p, 0, 16, h->stream));
    CM_TRY(cm_ingest_count_run(h, dp.as<int64_t>(), dc.as<int32_t>(), n, include_self, dl.as<int>(), dk.as<int>()));
    int km = 0;
    CM_TRY(d2h(h, &km, dk.p, 4));
    CM_TRY(sync(h));
    *out_kmax = km;
    if (!out_idx || !out_dist) return CM_OK;          // first call: size query
    if (kmax != km) CM_FAIL(h, CM_E_INVALID, "precomputed distances: output buffers sized for %d columns, need %d", kmax, km);
    if (n == 0 || km == 0) return CM_OK;
    CM_TRY(dv.alloc(h, 8 * (size_t)nnz)); CM_TRY(oi.alloc(h, 8 * (size_t)n * km)); CM_TRY(od.alloc(h, 8 * (size_t)n * km));
    CM_TRY(h2d(h, dv.p, data, 8 * (size_t)nnz));
    if (km + 1 > CM_INGEST_LDS_MAX) {   // rows beyond the in-LDS sorts: staged stable radix sort (round 4; no row limit any more)
        DevBuf ws;
        const size_t wsb = cm_ingest_huge_ws_bytes(n, km);
        CM_TRY(ws.alloc(h, wsb));
        CM_TRY(cm_ingest_fill_huge_run(h, dp.as<int64_t>(), dc.as<int32_t>(), dv.as<double>(), n, include_self, km, ws.p, wsb, oi.as<int64_t>(),
                                       od.as<double>()));
        CM_TRY(sync(h));
    } else
    CM_TRY(cm_ingest_fill_run(h, dp.as<int64_t>(), dc.as<int32_t>(), dv.as<double>(), n, include_self, km, oi.as<int64_t>(),
                              od.as<double>()));
    CM_TRY(d2h(h, out_idx, oi.p, 8 * (size_t)n * km));
    CM_TRY(d2h(h, out_dist, od.p, 8 * (size_t)n * km));
    return sync(h);
}

// ---------------------------------------------------------------- expression-transfer evaluation -------
cm_status cm_expression_metrics_dev(cm_handle_t h, const float *d_orig, const float *d_imputed, int64_t n, int64_t n_genes,
                                    const int32_t *d_group, int32_t n_groups, int32_t method, float *d_out) {
    if (!h) return CM_E_INVALID;
    if (method == CM_EXPR_SPEARMAN) {
        CM_TRY(begin_call(h, cm_expr_spearman_ws_bytes(n, n_genes)));
        return cm_expr_spearman_run(h, d_orig, d_imputed, n, n_genes, d_group, n_groups, d_out);
    }
    CM_TRY(begin_call(h, cm_expr_metrics_ws_bytes(n, n_genes)));
    return cm_expr_metrics_run(h, d_orig, d_imputed, n, n_genes, d_group, n_groups, method, d_out);
}

cm_status cm_expression_metrics(cm_handle_t h, const float *orig, const float *imputed, int64_t n, int64_t n_genes,
                                const int32_t *group, int32_t n_groups, int32_t method, float *out) {
    if (!h) return CM_E_INVALID;
    if (n < 0 || n_genes < 1) CM_FAIL(h, CM_E_INVALID, "expression metrics: bad shape");
    CM_TRY(begin_call(h, method == CM_EXPR_SPEARMAN ? cm_expr_spearman_ws_bytes(n, n_genes) : cm_expr_metrics_ws_bytes(n, n_genes)));
    const size_t nb = 4 * (size_t)n * n_genes, ng = group ? (size_t)n_groups : 0;
    DevBuf da, db, dg, dout;
    CM_TRY(da.alloc(h, nb)); CM_TRY(db.alloc(h, nb)); CM_TRY(dout.alloc(h, 4 * (ng + 1) * (size_t)n_genes));
    CM_TRY(h2d(h, da.p, orig, nb)); CM_TRY(h2d(h, db.p, imputed, nb));
    if (group) { CM_TRY(dg.alloc(h, 4 * (size_t)n)); CM_TRY(h2d(h, dg.p, group, 4 * (size_t)n)); }
    if (method == CM_EXPR_SPEARMAN)
        CM_TRY(cm_expr_spearman_run(h, da.as<float>(), db.as<float>(), n, n_genes, group ? dg.as<int32_t>() : nullptr, (int)ng,
                                    dout.as<float>()));
    else
        CM_TRY(cm_expr_metrics_run(h, da.as<float>(), db.as<float>(), n, n_genes, group ? dg.as<int32_t>() : nullptr, (int)ng, method,
                                   dout.as<float>()));
    CM_TRY(d2h(h, out, dout.p, 4 * (ng + 1) * (size_t)n_genes));
    return sync(h);
}

// ---------------------------------------------------------------- diagnostics --------------------------
// Raw v_mfma_f32_32x32x16_f16 problems, one wave each: D = A(32x16) * B(16x32) + C(32x32).  Used by the tests to pin
// the instruction's accumulation arithmetic (the rounding model behind the phase-1 certificate).
namespace {
typedef _Float16 dbg_half8 __attribute__((ext_vector_type(8)));
typedef float dbg_f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(64) void debug_mfma_f16(const _Float16 *__restrict__ A, const _Float16 *__restrict__ B,
                                                     const float *__restrict__ C, float *__restrict__ D, int chain) {
    const int lane = threadIdx.x, c = lane & 31, hh = lane >> 5;
    const int64_t p = blockIdx.x;
    dbg_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = C[p * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * hh) * 32 + c];
    for (int s = 0; s < chain; ++s) {   // chain > 1: consecutive k-steps accumulate into the same registers
        dbg_half8 a, b;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            a[e] = A[((p * chain + s) * 32 + c) * 16 + 8 * hh + e];      // A[row c][k = 8 hh + e]
            b[e] = B[((p * chain + s) * 16 + 8 * hh + e) * 32 + c];      // B[k = 8 hh + e][col c]
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) D[p * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * hh) * 32 + c] = acc[r];
}
}  // namespace

// the same for v_mfma_f32_16x16x32_f16 (knn_cand_f16r): D = A(16x32) * B(32x16) + C(16x16) per problem and chain step
namespace {
typedef float dbg_f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void debug_mfma_f16_k32(const _Float16 *__restrict__ A, const _Float16 *__restrict__ B,
                                                         const float *__restrict__ C, float *__restrict__ D, int chain) {
    const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
    const int64_t p = blockIdx.x;
    dbg_f32x4 acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = C[p * 256 + (4 * g + r) * 16 + c];
    for (int s = 0; s < chain; ++s) {
        dbg_half8 a, b;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            a[e] = A[((p * chain + s) * 16 + c) * 32 + 8 * g + e];      // A[row c][k = 8 g + e]
            b[e] = B[((p * chain + s) * 32 + 8 * g + e) * 16 + c];      // B[k = 8 g + e][col c]
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) D[p * 256 + (4 * g + r) * 16 + c] = acc[r];
}
}  // namespace

cm_status cm_debug_mfma_f16_k32(cm_handle_t h, const uint16_t *a, const uint16_t *b, const float *c, float *d, int32_t n,
                                int32_t chain) {
    if (!h) return CM_E_INVALID;
    if (n < 1 || chain < 1) CM_FAIL(h, CM_E_INVALID, "debug mfma: bad shape");
    CM_TRY(begin_call(h, 0));
    DevBuf da, db, dc, dd;
    const size_t na = (size_t)n * chain * 512 * 2, nc = (size_t)n * 256 * 4;
    CM_TRY(da.alloc(h, na)); CM_TRY(db.alloc(h, na)); CM_TRY(dc.alloc(h, nc)); CM_TRY(dd.alloc(h, nc));
    CM_TRY(h2d(h, da.p, a, na)); CM_TRY(h2d(h, db.p, b, na)); CM_TRY(h2d(h, dc.p, c, nc));
    hipLaunchKernelGGL(debug_mfma_f16_k32, dim3((unsigned)n), dim3(64), 0, h->stream, da.as<_Float16>(), db.as<_Float16>(),
                       dc.as<float>(), dd.as<float>(), (int)chain);
    CM_HIP(h, hipGetLastError());
    CM_TRY(d2h(h, d, dd.p, nc));
    return sync(h);
}

cm_status cm_debug_mfma_f16(cm_handle_t h, const uint16_t *a, const uint16_t *b, const float *c, float *d, int32_t n,
                            int32_t chain) {
    if (!h) return CM_E_INVALID;
    if (n < 1 || chain < 1) CM_FAIL(h, CM_E_INVALID, "debug mfma: bad shape");
    CM_TRY(begin_call(h, 0));
    DevBuf da, db, dc, dd;
    const size_t na = (size_t)n * chain * 512 * 2, nc = (size_t)n * 1024 * 4;
    CM_TRY(da.alloc(h, na)); CM_TRY(db.alloc(h, na)); CM_TRY(dc.alloc(h, nc)); CM_TRY(dd.alloc(h, nc));
    CM_TRY(h2d(h, da.p, a, na)); CM_TRY(h2d(h, db.p, b, na)); CM_TRY(h2d(h, dc.p, c, nc));
    hipLaunchKernelGGL(debug_mfma_f16, dim3((unsigned)n), dim3(64), 0, h->stream, da.as<_Float16>(), db.as<_Float16>(),
                       dc.as<float>(), dd.as<float>(), (int)chain);
    CM_HIP(h, hipGetLastError());
    CM_TRY(d2h(h, d, dd.p, nc));
    return sync(h);
}
