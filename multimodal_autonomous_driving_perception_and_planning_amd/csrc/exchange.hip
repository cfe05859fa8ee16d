// Wire format of the per-frame track tables that ranks all-gather (BASELINE config 5).
//
// The reference has no counterpart (no multi-process code at all); the consumer is what reads
// MultiObjectTracker.update()'s return value (src/tracking/multi_object_tracker.py:236-241), e.g. the interaction
// tagger.  A table travels as a 16-byte header + tcap x 32-byte rows ("Tcap=64 rows x 32 B"), half of
// the 64-byte rows the tracker keeps in HBM: HBM-bound byte shuffling, one thread per row, coalesced 64-B reads
// and 32-B writes.
#include "common.h"

#include <dlfcn.h>
#include <cstring>
#include <mutex>

namespace {

// one thread per (table, row); tables are [n_streams][n_sel] selected frames of snap[n_streams][n_frames][tcap]
__global__ void __launch_bounds__(256) pack_tracks_kernel(int n_streams, int n_frames, int tcap, int frame_lo, int n_sel,
                                                          int stream0, int frame0, const av_track_row* __restrict__ snap,
                                                          const int32_t* __restrict__ snap_n, const int32_t* __restrict__ frame_count,
                                                          uint8_t* __restrict__ wire) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)n_streams * n_sel * tcap;
    if (i >= total) return;
    const int r = (int)(i % tcap);
    const long long t = i / tcap;                       // table index = s * n_sel + k
    const int s = (int)(t / n_sel), k = (int)(t - (long long)s * n_sel);
    const int f = frame_lo + k;
    const size_t tb = AV_WIRE_HDR_BYTES + (size_t)tcap * AV_WIRE_ROW_BYTES;
    // header.frame: with the detector's counters, the stream's detector frame count at frame f (the counter holds the count
    // after the window's last frame) -- what the one-launch step stamps; without them, the index within the window
    const int frame = frame_count ? frame0 + frame_count[s] - (n_frames - 1 - f) : frame0 + f;
    wire_put(wire + (size_t)t * tb, r, snap_n[(size_t)s * n_frames + f], tcap, snap + ((size_t)s * n_frames + f) * tcap, stream0 + s, frame);
}

// ---- RCCL, opened on demand (no link-time dependency) ---------------------------------------------------------------------
struct NcclId { char b[128]; };                     // ncclUniqueId
struct Rccl {
    void* so = nullptr;
    int (*get_unique_id)(NcclId*) = nullptr;
    int (*comm_init_rank)(void**, int, NcclId, int) = nullptr;
    int (*comm_destroy)(void*) = nullptr;
    int (*all_gather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    const char* (*error_string)(int) = nullptr;
};
char g_rccl_why[256] = "symbols missing";     // why rccl() returned null (dlerror() is cleared by reading it: kept once)
Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // RTLD_NOLOAD first: inside a torch process this is torch's own copy of the library, not a second one beside it
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            r.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
            if (r.so) break;
        }
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            if (r.so) break;
            r.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (!r.so) {
                const char* e = dlerror();
                if (e) snprintf(g_rccl_why, sizeof(g_rccl_why), "%s", e);
            }
        }
        if (r.so) {
            r.get_unique_id = reinterpret_cast<int (*)(NcclId*)>(dlsym(r.so, "ncclGetUniqueId"));
            r.comm_init_rank = reinterpret_cast<int (*)(void**, int, NcclId, int)>(dlsym(r.so, "ncclCommInitRank"));
            r.comm_destroy = reinterpret_cast<int (*)(void*)>(dlsym(r.so, "ncclCommDestroy"));
            r.all_gather = reinterpret_cast<int (*)(const void*, void*, size_t, int, void*, hipStream_t)>(dlsym(r.so, "ncclAllGather"));
            r.error_string = reinterpret_cast<const char* (*)(int)>(dlsym(r.so, "ncclGetErrorString"));
        }
    });
    return (r.so && r.get_unique_id && r.comm_init_rank && r.comm_destroy && r.all_gather) ? &r : nullptr;
}
#define AV_RCCL(call, what)                                                                                          \
    do {                                                                                                             \
        const int rc_ = (call);                                                                                      \
        if (rc_ != 0) {                                                                                              \
            av_set_error("%s: %s", what, R->error_string ? R->error_string(rc_) : "RCCL error");                     \
            return AV_EHIP;                                                                                          \
        }                                                                                                            \
    } while (0)

}  // namespace

extern "C" {

size_t av_wire_table_bytes(int tcap) { return AV_WIRE_HDR_BYTES + (size_t)tcap * AV_WIRE_ROW_BYTES; }

int av_pack_tracks(av_ctx* ctx, av_stream_t stream, int n_streams, int n_frames, int tcap, int frame_lo, int n_sel,
                   int stream0, int frame0, const av_track_row* snap, const int32_t* snap_n, const int32_t* frame_count, void* wire) {
    AV_REQUIRE(ctx && snap && snap_n && wire, AV_EINVAL, "av_pack_tracks: null argument");
    AV_REQUIRE(n_streams > 0 && n_frames > 0 && tcap > 0 && tcap <= 1024, AV_EINVAL, "av_pack_tracks: bad dimensions");
    AV_REQUIRE(frame_lo >= 0 && n_sel > 0 && frame_lo + n_sel <= n_frames, AV_EINVAL,
               "av_pack_tracks: frames [%d, %d) outside the window of %d", frame_lo, frame_lo + n_sel, n_frames);
    static_assert(sizeof(av_wire_row) == AV_WIRE_ROW_BYTES && sizeof(av_wire_hdr) == AV_WIRE_HDR_BYTES, "wire layout");
    const long long total = (long long)n_streams * n_sel * tcap;
    const unsigned grid = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(pack_tracks_kernel, dim3(grid), dim3(256), 0, as_stream(stream), n_streams, n_frames, tcap,
                       frame_lo, n_sel, stream0, frame0, snap, snap_n, frame_count, (uint8_t*)wire);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

int av_comm_unique_id(void* id128) {
    AV_REQUIRE(id128, AV_EINVAL, "av_comm_unique_id: null argument");
    Rccl* R = rccl();
    AV_REQUIRE(R, AV_ESTATE, "av_comm_unique_id: librccl could not be opened (%s)", g_rccl_why);
    AV_RCCL(R->get_unique_id(reinterpret_cast<NcclId*>(id128)), "ncclGetUniqueId");
    return AV_OK;
}

int av_comm_create(av_ctx* ctx, const void* id128, int rank, int world, void** comm) {
    AV_REQUIRE(ctx && id128 && comm, AV_EINVAL, "av_comm_create: null argument");
    AV_REQUIRE(world >= 1 && rank >= 0 && rank < world, AV_EINVAL, "av_comm_create: rank %d of %d", rank, world);
    Rccl* R = rccl();
    AV_REQUIRE(R, AV_ESTATE, "av_comm_create: librccl could not be opened");
    AV_HIP(hipSetDevice(ctx->device));
    NcclId id;
    std::memcpy(&id, id128, sizeof(id));
    AV_RCCL(R->comm_init_rank(comm, world, id, rank), "ncclCommInitRank");
    return AV_OK;
}

int av_comm_destroy(void* comm) {
    if (!comm) return AV_OK;
    Rccl* R = rccl();
    AV_REQUIRE(R, AV_ESTATE, "av_comm_destroy: librccl could not be opened");
    AV_RCCL(R->comm_destroy(comm), "ncclCommDestroy");
    return AV_OK;
}

int av_allgather_tracks(av_ctx* ctx, void* comm, av_stream_t stream, const void* send, void* recv, size_t bytes_per_rank) {
    AV_REQUIRE(ctx && comm && send && recv, AV_EINVAL, "av_allgather_tracks: null argument");
    AV_REQUIRE(bytes_per_rank > 0, AV_EINVAL, "av_allgather_tracks: nothing to gather");
    Rccl* R = rccl();
    AV_REQUIRE(R, AV_ESTATE, "av_allgather_tracks: librccl could not be opened");
    AV_RCCL(R->all_gather(send, recv, bytes_per_rank, 1 /* ncclUint8 */, comm, as_stream(stream)), "ncclAllGather");
    return AV_OK;
}

}  // extern "C"
