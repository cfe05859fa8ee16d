// Context, error reporting, fork/join, hipGraph capture and event helpers of libavhot.so.
#include "common.h"

#include <chrono>

#include <cstring>

static thread_local char g_err[512] = "no error";

int av_simdet_ctx_init(av_ctx* ctx);        // simdet.hip: builds the per-seed draw table

void av_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {

int av_version(void) { return AV_VERSION; }
const char* av_last_error_string(void) { return g_err; }

int av_device_count(int* count) {
    AV_REQUIRE(count, AV_EINVAL, "av_device_count: null out pointer");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *count = n;
    return AV_OK;
}

int av_ctx_create(int device, av_ctx** out) {
    AV_REQUIRE(out, AV_EINVAL, "av_ctx_create: null out pointer");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        av_set_error("av_ctx_create: no HIP device visible (libavhot has no CPU path)");
        return AV_ENODEV;
    }
    AV_REQUIRE(device >= 0 && device < n, AV_EINVAL, "av_ctx_create: device %d out of range [0,%d)", device, n);
    AV_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    AV_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        av_set_error("av_ctx_create: device %d is %s; libavhot is built for gfx950 only", device, prop.gcnArchName);
        return AV_ENODEV;
    }
    av_ctx* c = new (std::nothrow) av_ctx();
    AV_REQUIRE(c, AV_ENOMEM, "av_ctx_create: out of host memory");
    c->device = device;
    c->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    AV_HIP(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
    AV_HIP(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    AV_HIP(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    // class cdf of the simulated detector (detector.py:159; legacy RandomState.choice: cumsum, /= last)
    const double p[8] = {0.6, 0.15, 0.1, 0.05, 0.03, 0.05, 0.01, 0.01};
    double cdf[8];
    double acc = 0.0;
    for (int i = 0; i < 8; ++i) {
        acc = acc + p[i];
        cdf[i] = acc;
    }
    const double last = cdf[7];
    for (int i = 0; i < 8; ++i) cdf[i] = cdf[i] / last;
    AV_HIP(hipMalloc(&c->d_cdf, sizeof(cdf)));
    AV_HIP(hipMemcpy(c->d_cdf, cdf, sizeof(cdf), hipMemcpyHostToDevice));
    const int rc = av_simdet_ctx_init(c);
    if (rc != AV_OK) return rc;
    *out = c;
    return AV_OK;
}

// Sub-contexts are released by the translation unit that owns them (lane.hip / yolo.hip); the
// weak defaults keep the library linkable when one of them is left out of a build.
__attribute__((weak)) int av_lane_ctx_free(av_ctx*) { return AV_OK; }
__attribute__((weak)) int av_yolo_ctx_free(av_ctx*) { return AV_OK; }

int av_ctx_destroy(av_ctx* ctx) {
    if (!ctx) return AV_OK;
    (void)hipSetDevice(ctx->device);
    for (hipGraphExec_t g : ctx->graphs)
        if (g) (void)hipGraphExecDestroy(g);
    av_lane_ctx_free(ctx);
    av_yolo_ctx_free(ctx);
    if (ctx->d_ptab) (void)hipFree(ctx->d_ptab);
    if (ctx->d_cdf) (void)hipFree(ctx->d_cdf);
    if (ctx->d_simtab) (void)hipFree(ctx->d_simtab);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->side) (void)hipStreamDestroy(ctx->side);
    delete ctx;
    return AV_OK;
}

int av_ctx_device(const av_ctx* ctx, int* device) {
    AV_REQUIRE(ctx && device, AV_EINVAL, "av_ctx_device: null argument");
    *device = ctx->device;
    return AV_OK;
}

int av_side_stream(av_ctx* ctx, av_stream_t* side) {
    AV_REQUIRE(ctx && side, AV_EINVAL, "av_side_stream: null argument");
    *side = (av_stream_t)ctx->side;
    return AV_OK;
}

int av_fork(av_ctx* ctx, av_stream_t main) {
    AV_REQUIRE(ctx, AV_EINVAL, "av_fork: null ctx");
    AV_HIP(hipEventRecord(ctx->ev_fork, as_stream(main)));
    AV_HIP(hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
    return AV_OK;
}

int av_join(av_ctx* ctx, av_stream_t main) {
    AV_REQUIRE(ctx, AV_EINVAL, "av_join: null ctx");
    AV_HIP(hipEventRecord(ctx->ev_join, ctx->side));
    AV_HIP(hipStreamWaitEvent(as_stream(main), ctx->ev_join, 0));
    return AV_OK;
}

int av_graph_begin(av_ctx* ctx, av_stream_t stream) {
    AV_REQUIRE(ctx && stream, AV_EINVAL, "av_graph_begin: ctx and a non-null stream are required");
    AV_HIP(hipStreamBeginCapture(as_stream(stream), hipStreamCaptureModeThreadLocal));
    return AV_OK;
}

int av_graph_end(av_ctx* ctx, av_stream_t stream, int* graph_id) {
    AV_REQUIRE(ctx && stream && graph_id, AV_EINVAL, "av_graph_end: null argument");
    hipGraph_t g = nullptr;
    AV_HIP(hipStreamEndCapture(as_stream(stream), &g));
    hipGraphExec_t ge = nullptr;
    hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) {
        av_set_error("hipGraphInstantiate -> %s", hipGetErrorString(e));
        return AV_EHIP;
    }
    ctx->graphs.push_back(ge);
    *graph_id = (int)ctx->graphs.size() - 1;
    return AV_OK;
}

int av_graph_launch(av_ctx* ctx, int graph_id, av_stream_t stream) {
    AV_REQUIRE(ctx && graph_id >= 0 && graph_id < (int)ctx->graphs.size() && ctx->graphs[graph_id], AV_EINVAL,
               "av_graph_launch: bad graph id %d", graph_id);
    AV_HIP(hipGraphLaunch(ctx->graphs[graph_id], as_stream(stream)));
    return AV_OK;
}

int av_graph_destroy(av_ctx* ctx, int graph_id) {
    AV_REQUIRE(ctx && graph_id >= 0 && graph_id < (int)ctx->graphs.size(), AV_EINVAL,
               "av_graph_destroy: bad graph id %d", graph_id);
    if (ctx->graphs[graph_id]) {
        AV_HIP(hipGraphExecDestroy(ctx->graphs[graph_id]));
        ctx->graphs[graph_id] = nullptr;
    }
    return AV_OK;
}

int av_event_create(void** ev) {
    AV_REQUIRE(ev, AV_EINVAL, "av_event_create: null out pointer");
    hipEvent_t e;
    AV_HIP(hipEventCreate(&e));
    *ev = (void*)e;
    return AV_OK;
}
int av_event_destroy(void* ev) {
    if (ev) AV_HIP(hipEventDestroy((hipEvent_t)ev));
    return AV_OK;
}
int av_event_record(void* ev, av_stream_t stream) {
    AV_REQUIRE(ev, AV_EINVAL, "av_event_record: null event");
    AV_HIP(hipEventRecord((hipEvent_t)ev, as_stream(stream)));
    return AV_OK;
}
int av_event_elapsed_ms(void* start, void* stop, float* ms) {
    AV_REQUIRE(start && stop && ms, AV_EINVAL, "av_event_elapsed_ms: null argument");
    AV_HIP(hipEventSynchronize((hipEvent_t)stop));
    AV_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return AV_OK;
}
int av_stream_sync(av_stream_t stream) {
    AV_HIP(hipStreamSynchronize(as_stream(stream)));
    return AV_OK;
}

// Waits for the stream by polling an event instead of sleeping on the completion interrupt: a per-frame call waits for
// ~10 us of work, and the interrupt path's wake-up latency is of that order.  The event belongs to the device that is
// current when it is recorded (one per device and thread, created on first use); the poll is bounded -- after 2 ms without
// completion the wait falls back to hipStreamSynchronize, whose error (a faulted or hung kernel) is returned instead of
// spinning a core forever.
int av_stream_sync_spin(av_stream_t stream) {
    constexpr int MAX_DEV = 16;
    static thread_local hipEvent_t evs[MAX_DEV] = {};
    int dev = 0;
    AV_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= MAX_DEV) {
        AV_HIP(hipStreamSynchronize(as_stream(stream)));
        return AV_OK;
    }
    hipEvent_t& ev = evs[dev];
    if (!ev) AV_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    AV_HIP(hipEventRecord(ev, as_stream(stream)));
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned it = 0;; ++it) {
        const hipError_t e = hipEventQuery(ev);
        if (e == hipSuccess) return AV_OK;
        if (e != hipErrorNotReady) {
            av_set_error("av_stream_sync_spin: hipEventQuery -> %s", hipGetErrorString(e));
            return AV_EHIP;
        }
        if ((it & 63) == 63 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) {
            AV_HIP(hipStreamSynchronize(as_stream(stream)));
            return AV_OK;
        }
        __builtin_ia32_pause();
    }
}

// Pinned host staging for the per-frame class surfaces: one packed upload and one packed download per call.
int av_host_alloc(void** p, size_t bytes) {
    AV_REQUIRE(p && bytes > 0, AV_EINVAL, "av_host_alloc: null out pointer or zero size");
    AV_HIP(hipHostMalloc(p, bytes, hipHostMallocDefault));
    return AV_OK;
}
int av_host_free(void* p) {
    if (p) AV_HIP(hipHostFree(p));
    return AV_OK;
}
int av_copy_h2d(void* dst_dev, const void* src_host, size_t bytes, av_stream_t stream) {
    AV_REQUIRE(dst_dev && src_host, AV_EINVAL, "av_copy_h2d: null pointer");
    AV_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, as_stream(stream)));
    return AV_OK;
}
int av_copy_d2h(void* dst_host, const void* src_dev, size_t bytes, av_stream_t stream, int sync) {
    AV_REQUIRE(dst_host && src_dev, AV_EINVAL, "av_copy_d2h: null pointer");
    AV_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    if (sync) AV_HIP(hipStreamSynchronize(as_stream(stream)));
    return AV_OK;
}

}  // extern "C"
