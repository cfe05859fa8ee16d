// K1-K3: batched IoU tracker.
//
// Reference: MultiObjectTracker (src/tracking/multi_object_tracker.py)
//   _compute_iou :84-105   integer areas, one float64 divide, 0.0 when boxes do not overlap
//   _associate   :113-164  greedy: take the global arg-max of the T x D IoU matrix (first in
//                          row-major order on ties, :150) until max < iou_threshold (:147)
//   update       :166-241  matched / missed / born / dead / confirmed
//
// Mapping: one workgroup per video stream, one thread per track row (blockDim = tcap), frames of
// the window processed sequentially.  The table lives in registers for the whole window; rows are
// kept in ascending-id order (== the reference's dict insertion order) and compacted through LDS
// only on frames where a track dies.  The T x D matrix is never materialised: each row caches
// its best unused detection (lowest column on ties) and recomputes it only when that column is
// taken, which reproduces the reference's pick order exactly.  The float64 quotient inter/union is
// formed from exact integers, so every comparison sees the bits NumPy sees.
//
// REP > 1 (tcap 64, dcap <= REP): the per-frame work of ONE stream is a latency chain a single wave cannot
// hide, so REP waves each hold the same table (lane == row) and split the association by detection
// column: wave w tests column w against every row and resolves that column's contest, the results
// cross through LDS with one barrier per frame, and every wave then applies the identical update to its
// own copy (private LDS scratch, no further communication).  Only the LAST wave -- which owns a column
// only when a frame has REP detections -- keeps complete rows (ids, ages, hits, confidences, history rings)
// and does every global write; the others carry just what column ownership needs: boxes, miss counters
// and the row count, kept in step by the same matches, births and order-preserving compactions.
// Windows (PIPE): the complete rows move to a ninth wave that trails the column waves by one frame -- see tracker_body.
#include <mutex>
#include "common.h"

namespace {

constexpr int HDR_INTS = 16;

__host__ __device__ inline size_t state_bytes(int tcap, int L) {
    return (size_t)HDR_INTS * 4 + (size_t)tcap * sizeof(av_track_row) + (size_t)tcap * L * 4 * sizeof(double);
}

struct Row {
    int id, x1, y1, x2, y2, cls, age, hits, misses, slot, hlen;
    int hpos;          // hlen % trajectory_length, kept incrementally (no integer divide in the frame loop)
    float vx, vy;      // last velocity (valid when hlen >= 2)
    double conf;
};

__device__ __forceinline__ double iou_f64(int ax1, int ay1, int ax2, int ay2, int bx1, int by1, int bx2, int by2) {
    const int xi1 = ax1 > bx1 ? ax1 : bx1, yi1 = ay1 > by1 ? ay1 : by1;
    const int xi2 = ax2 < bx2 ? ax2 : bx2, yi2 = ay2 < by2 ? ay2 : by2;
    if (xi2 <= xi1 || yi2 <= yi1) return 0.0;
    const long long inter = (long long)(xi2 - xi1) * (long long)(yi2 - yi1);
    const long long a1 = (long long)(ax2 - ax1) * (long long)(ay2 - ay1);
    const long long a2 = (long long)(bx2 - bx1) * (long long)(by2 - by1);
    const long long uni = a1 + a2 - inter;
    return uni > 0 ? (double)inter / (double)uni : 0.0;
}

__device__ __forceinline__ int nth_set_bit(unsigned long long m, int n) {
    for (int i = 0; i < n; ++i) m &= m - 1;
    return __ffsll((long long)m) - 1;
}

struct Shared {
    int d2t[64];
    double w_iou[16];
    int w_row[16];
    int w_col[16];
    int w_cnt[16];
    unsigned slot_bits[32];
    int birth_slot[64];
    int misc[4];
};

// Workgroup synchronisation that orders LDS traffic only (a fence on the "local" address space: no wait on
// outstanding global stores whatever the compiler makes of a full __syncthreads(); the per-frame outputs are
// write-only).  A single-wave workgroup needs no barrier at all: one wave's DS operations execute in issue order.
template <bool MULTIWAVE>
__device__ __forceinline__ void lds_sync() {
    if (MULTIWAVE) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    }
}

// DREG > 0: the row's IoU against every detection (dcap <= DREG) is computed once per frame and kept in
// registers; DREG == 0: generic path for larger dcap (best candidate recomputed when its column is taken).
// (a device function of the stream index s: tracker_kernel runs it with s = blockIdx.x, the fused time-step kernel of step.hip
// with its own workgroup-to-stream map; smem = the workgroup's dynamic LDS)
// PIPE (replica kernel, windows): a ninth wave holds the complete rows and runs ONE FRAME BEHIND the eight column waves.  What a
// frame does to the table is decided by the light state alone (boxes, miss counters, row count); ids, ages, confidences, history
// rings, the snapshot rows and det2trk are derived from those decisions.  tools/ktime.py: the wave that kept the complete rows was
// the critical path of every frame (2 820 of 2 900 cycles busy; the others waited 600-800 cycles at the barrier for it).  The
// column waves now publish a frame's decisions (every row's matched column + the taken columns, 65 bytes, by wave 0) and go on;
// the ninth wave applies them after the next frame's barrier -- the same barrier, no other synchronisation -- from a double-
// buffered detection chunk.  Same arithmetic on the same values in the same order per row: results are bit-identical.
template <bool MULTIWAVE, int DREG, int REP, bool TIMED = false, int PIPE = 0>
__device__ __forceinline__ void tracker_body(const av_tracker_cfg& cfg, int n_frames, int dcap, const int32_t* __restrict__ det_n,
                               const int32_t* __restrict__ det_box, const int32_t* __restrict__ det_cls,
                               const double* __restrict__ det_conf, int tcap, unsigned char* __restrict__ state_all,
                               av_track_row* __restrict__ snap, int32_t* __restrict__ snap_n,
                               int32_t* __restrict__ det2trk, int chunk_frames, const int s, unsigned char* smem,
                               const unsigned long long wave_map = 0xFEDCBA9876543210ull, const unsigned char* init_src = nullptr,
                               const bool coherent_out = false, const long long det_sf0 = -1, int* publish_flag = nullptr,
                               const int publish_value = 0, const bool publish_clock = false, const double* det_area = nullptr) {
    // det_area (replica kernel, n_frames == 1, det_sf0 == 0): the detection arrays ARE the staged frame -- they sit in this workgroup's
    // LDS, complete with the boxes' areas, behind a barrier the caller has passed: no copy into the chunk buffer, no barriers around it
    // publish_flag (n_frames == 1 only; the overlapped time-step): the wave that keeps the complete rows stores publish_value there as
    // soon as the stream's record (rows + header) is written AND acknowledged, and writes the frame's outputs (snapshot rows, det2trk)
    // after that -- the successor launch waits for the record, not for the outputs
    // det_sf0 >= 0 (replica kernel): the detection arrays are indexed from this frame slot instead of s * n_frames (the one-launch
    // time-step hands over the LDS copy its detector role wrote, slot 0)
    // init_src: where the stream's header and rows are read from at entry instead of its state record (the one-launch time-step of
    // step.hip hands over an LDS copy); everything is written to the state record as always -- coherent_out: with device-scope
    // (write-through) stores, for a successor launch already in flight, possibly on another XCD
    constexpr bool REPL = REP > 1;
    static_assert(!(REPL && MULTIWAVE), "replica waves hold the whole table: tcap must be 64");
    static_assert(REP == 1 || REP == 8, "the exchange buffers are laid out for 8 columns");
    static_assert(!PIPE || REPL, "the trailing wave belongs to the replica kernel");
    // wave_map: role (column / trailing wave) of every hardware wave, a nibble each.  Waves go to the CU's four SIMDs round-robin, and a
    // lone wave on a SIMD runs its dependent chain fastest: the map pairs the columns that usually hold a detection (0, 1, 2) with the
    // ones that rarely do, and keeps the trailing wave(s) away from them (launch_replicas).  Identity for every other use.
    const int lane = threadIdx.x & 63;
    const int wid = REPL ? (int)((wave_map >> (4 * __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6))) & 15ull) : (int)(threadIdx.x >> 6);
    const int tid = REPL ? wid * 64 + lane : (int)threadIdx.x;
    const int row = REPL ? lane : tid;                 // table row this thread holds
    const bool full = !REPL || wid == (PIPE ? REP : REP - 1);    // this wave keeps complete rows and writes the outputs
    // per replica: Shared | stage[tcap];  then the detection chunk;  then (REPL) the exchange buffers
    const size_t sh_bytes = (sizeof(Shared) + 63) & ~size_t(63);
    const size_t rep_bytes = sh_bytes + (size_t)tcap * sizeof(av_track_row);
    unsigned char* rbase = smem + (REPL ? (size_t)wid * rep_bytes : 0);
    Shared& sh = *reinterpret_cast<Shared*>(rbase);
    av_track_row* stage = reinterpret_cast<av_track_row*>(rbase + sh_bytes);
    // detections of a chunk of FC frames: n[FC] | box[FC][dcap][4] | cls[FC][dcap] | conf[FC][dcap] | area[FC][dcap]
    unsigned char* chunk = smem + (size_t)(REP + PIPE) * rep_bytes;
    const int FC = chunk_frames;
    const size_t chunk_sz = ((size_t)((FC + 3) & ~3) * 4 + (size_t)FC * dcap * 16 + (((size_t)FC * dcap + 1) & ~size_t(1)) * 4 +
                             (size_t)FC * dcap * 16 + 15) & ~size_t(15);
    int *c_n, *c_box, *c_cls;
    double *c_conf, *c_area;                           // area: (x2-x1)*(y2-y1) of every staged detection, exact in f64
    auto set_chunk = [&](unsigned char* base) {
        c_n = reinterpret_cast<int*>(base);
        c_box = c_n + ((FC + 3) & ~3);
        c_cls = c_box + (size_t)FC * dcap * 4;
        c_conf = reinterpret_cast<double*>(c_cls + (((size_t)FC * dcap + 1) & ~size_t(1)));
        c_area = c_conf + (size_t)FC * dcap;
    };
    set_chunk(chunk);
    int cpar = 1;                                      // PIPE: chunk buffer in use (toggled at every chunk start)
    // exchange, one buffer per frame parity: cand[64 rows][8 cols] bytes | count[8] | any[8] | winner[8]
    constexpr int XB = 576;
    unsigned char* xchg = chunk + (PIPE ? 2 : 1) * chunk_sz;
    // PIPE: a frame's decisions for the trailing wave, by frame parity: matched column of every row (signed byte) | taken columns
    constexpr int RB = 80;
    signed char* rec = reinterpret_cast<signed char*>(xchg + 2 * XB);

    // a dependent per-frame chain: let these few waves issue ahead of throughput kernels sharing the SIMD
    __builtin_amdgcn_s_setprio(3);
    const int nwaves = (tcap + 63) >> 6;
    const int L = cfg.trajectory_length;

    unsigned char* st = state_all + (size_t)s * state_bytes(tcap, L);
    int* hdr = reinterpret_cast<int*>(st);
    av_track_row* rows = reinterpret_cast<av_track_row*>(st + HDR_INTS * 4);
    double* hist = reinterpret_cast<double*>(st + HDR_INTS * 4 + (size_t)tcap * sizeof(av_track_row));

    const int* hdr_in = init_src ? reinterpret_cast<const int*>(init_src) : hdr;
    const av_track_row* rows_in = init_src ? reinterpret_cast<const av_track_row*>(init_src + HDR_INTS * 4) : rows;
    int T = hdr_in[0], next_id = hdr_in[1], frame_count = hdr_in[2], status = hdr_in[3];
    Row r{};
    if (row < T) {
        const av_track_row g = rows_in[row];
        r.id = g.id, r.x1 = g.x1, r.y1 = g.y1, r.x2 = g.x2, r.y2 = g.y2, r.cls = g.cls;
        r.age = g.age, r.hits = g.hits, r.misses = g.misses, r.slot = g.slot, r.hlen = g.hist_len;
        r.hpos = g.hist_len % L;
        r.conf = g.conf, r.vx = g.vx, r.vy = g.vy;
    }
    // Retire the table loads here.  Otherwise the compiler places their wait at the first use inside
    // the frame loop, where (vmcnt counts loads and stores in order) it would also drain the previous
    // frame's output stores on every iteration.
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0), expcnt/lgkmcnt untouched
    if (row < 32) sh.slot_bits[row] = 0;
    lds_sync<MULTIWAVE>();
    if (row < T) atomicOr(&sh.slot_bits[r.slot >> 5], 1u << (r.slot & 31));
    lds_sync<MULTIWAVE>();

    // snapshot rows, live count and detection->track ids of one frame, from the state as it stands
    auto emit = [&](size_t sfo) {
        if (snap) {
            if (row < T) {
                av_track_row g;
                g.id = r.id, g.x1 = r.x1, g.y1 = r.y1, g.x2 = r.x2, g.y2 = r.y2, g.cls = r.cls;
                g.age = r.age, g.hits = r.hits, g.misses = r.misses, g.slot = r.slot, g.hist_len = r.hlen;
                g.flags = (r.hits >= cfg.min_hits) ? 1 : 0;
                g.conf = r.conf, g.vx = r.vx, g.vy = r.vy;
                snap[sfo * tcap + row] = g;
            }
            if (row == 0) snap_n[sfo] = T;
        }
        if (det2trk && row < dcap) det2trk[sfo * dcap + row] = sh.d2t[row];
    };
    // replica kernel: next chunk's detections, one element per thread, in flight during the current chunk
    int4 pf_box = make_int4(0, 0, 0, 0);
    int pf_cls = 0, pf_n = 0;
    double pf_conf = 0.0;
    auto prefetch = [&](size_t sf0, int nfr) {
        if (tid < nfr) pf_n = det_n[sf0 + tid];
        if (tid < nfr * dcap) {
            pf_box = reinterpret_cast<const int4*>(det_box)[sf0 * dcap + tid];
            pf_cls = det_cls[sf0 * dcap + tid];
            pf_conf = det_conf[sf0 * dcap + tid];
        }
    };
    const size_t dsf0 = det_sf0 >= 0 ? (size_t)det_sf0 : (size_t)s * n_frames;
    if (REPL && n_frames > 0 && !det_area) prefetch(dsf0, n_frames < FC ? n_frames : FC);
    int fl = -1;                               // frame within the staged chunk
    // TIMED (AVHOT_TRACKER_TIMED, tools/ktime.py): cycles of every wave of stream 0 by phase, summed over the window, left in det2trk
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
    if (TIMED) tprev = __builtin_amdgcn_s_memtime();
#define TRK_STAMP(k)                                                   \
    if (TIMED) {                                                       \
        const unsigned long long tn_ = __builtin_amdgcn_s_memtime();  \
        tacc[k] += tn_ - tprev;                                        \
        tprev = tn_;                                                   \
    }
    // ---- what a frame does to the table once its association is known (matched column of every row, taken columns) ----------------
    // (a lambda: the frame loop below runs it in place; with PIPE the trailing wave runs it one frame late)
    auto apply = [&](int matched_j, unsigned long long used, int nd, const int* dbox, const int* dcls, const double* dconf, size_t sf) {
        const bool active = row < T;
        // ---- matched / missed (:182-211) ---------------------------------------------------------
        if (active) {
            if (matched_j >= 0) {
                const int4 nb4 = *reinterpret_cast<const int4*>(dbox + matched_j * 4);
                if (full) {
                    const double ocx = (double)(r.x1 + r.x2) / 2.0, ocy = (double)(r.y1 + r.y2) / 2.0;
                    const double ncx = (double)(nb4.x + nb4.z) / 2.0, ncy = (double)(nb4.y + nb4.w) / 2.0;
                    r.conf = dconf[matched_j];
                    r.age += 1, r.hits += 1;
                    double4* h = reinterpret_cast<double4*>(hist + ((size_t)r.slot * L + r.hpos) * 4);
                    *h = make_double4(ncx, ncy, ncx - ocx, ncy - ocy);
                    r.vx = (float)(ncx - ocx), r.vy = (float)(ncy - ocy);
                    r.hlen += 1;
                    r.hpos = (r.hpos + 1 == L) ? 0 : r.hpos + 1;
                    sh.d2t[matched_j] = r.id;
                }
                r.x1 = nb4.x, r.y1 = nb4.y, r.x2 = nb4.z, r.y2 = nb4.w;
                r.misses = 0;
            } else {
                r.age += 1, r.misses += 1;
            }
        }

        TRK_STAMP(4)                                        // matched / missed
        // ---- births (:214-225) ------------------------------------------------------------------
        const unsigned long long dmask = nd >= 64 ? ~0ull : ((1ull << nd) - 1ull);
        const unsigned long long unm = ~used & dmask;
        const int nb = __popcll(unm);
        int nb_fit = nb;
        if (T + nb > tcap) nb_fit = tcap - T, status |= 1;
        if (nb > 0 && !full) {
            // light copy: the new rows' boxes only
            if (row >= T && row < T + nb_fit) {
                const int4 b4 = *reinterpret_cast<const int4*>(dbox + nth_set_bit(unm, row - T) * 4);
                r.x1 = b4.x, r.y1 = b4.y, r.x2 = b4.z, r.y2 = b4.w;
                r.misses = 0;
            }
            T += nb_fit;
        } else if (nb > 0) {
            // rank of each free history slot; the b-th birth takes the b-th free slot
            {
                const unsigned word = sh.slot_bits[row >> 5];
                const bool is_free = !((word >> (row & 31)) & 1u);
                int below = __popc(~word & ((1u << (row & 31)) - 1u));
                for (int w = 0; w < (row >> 5); ++w) below += 32 - __popc(sh.slot_bits[w]);
                if (is_free && below < nb_fit) sh.birth_slot[below] = row;
            }
            if (row < nd && ((unm >> row) & 1ull)) {
                const int b = __popcll(unm & ((1ull << row) - 1ull));
                sh.d2t[row] = next_id + b;
            }
            lds_sync<MULTIWAVE>();
            if (row >= T && row < T + nb_fit) {
                const int b = row - T;
                const int j = nth_set_bit(unm, b);
                r.id = next_id + b;
                r.x1 = dbox[j * 4 + 0], r.y1 = dbox[j * 4 + 1], r.x2 = dbox[j * 4 + 2], r.y2 = dbox[j * 4 + 3];
                r.cls = dcls[j];
                r.conf = dconf[j];
                r.age = 0, r.hits = 1, r.misses = 0;
                r.slot = sh.birth_slot[b];
                r.hlen = 1;
                r.hpos = (L == 1) ? 0 : 1;
                r.vx = 0.f, r.vy = 0.f;
                double4* h = reinterpret_cast<double4*>(hist + (size_t)r.slot * L * 4);
                *h = make_double4((double)(r.x1 + r.x2) / 2.0, (double)(r.y1 + r.y2) / 2.0, 0.0, 0.0);
                atomicOr(&sh.slot_bits[r.slot >> 5], 1u << (r.slot & 31));
            }
            next_id += nb;
            T += nb_fit;
        }

        TRK_STAMP(5)                                        // births
        // ---- deaths (:228-233): order-preserving compaction --------------------------------------
        const bool live_row = row < T;
        const bool dead = live_row && (r.misses > cfg.max_age);
        int any_dead;
        if (MULTIWAVE) {
            if (row == 0) sh.misc[0] = 0;
            lds_sync<MULTIWAVE>();
            if (dead) sh.misc[0] = 1;
            lds_sync<MULTIWAVE>();
            any_dead = sh.misc[0];
        } else {
            any_dead = __ballot(dead) != 0ull;
        }
        if (any_dead) {
            const bool keep = live_row && !dead;
            const unsigned long long kb = __ballot(keep);
            int pos = __popcll(kb & ((1ull << lane) - 1ull));
            int total = __popcll(kb);
            if (MULTIWAVE) {
                if (lane == 0) sh.w_cnt[wid] = total;
                lds_sync<MULTIWAVE>();
                total = 0;
                for (int w = 0; w < nwaves; ++w) {
                    if (w < wid) pos += sh.w_cnt[w];
                    total += sh.w_cnt[w];
                }
            }
            if (!full) {
                // light copy: boxes and miss counters move down the same way
                int* ls = reinterpret_cast<int*>(stage);
                if (keep) {
                    *reinterpret_cast<int4*>(ls + pos * 8) = make_int4(r.x1, r.y1, r.x2, r.y2);
                    ls[pos * 8 + 4] = r.misses;
                }
                lds_sync<MULTIWAVE>();
                T = total;
                if (row < T) {
                    const int4 b4 = *reinterpret_cast<const int4*>(ls + row * 8);
                    r.x1 = b4.x, r.y1 = b4.y, r.x2 = b4.z, r.y2 = b4.w;
                    r.misses = ls[row * 8 + 4];
                }
            } else {
                if (dead) atomicAnd(&sh.slot_bits[r.slot >> 5], ~(1u << (r.slot & 31)));
                if (keep) {
                    av_track_row g;
                    g.id = r.id, g.x1 = r.x1, g.y1 = r.y1, g.x2 = r.x2, g.y2 = r.y2, g.cls = r.cls;
                    g.age = r.age, g.hits = r.hits, g.misses = r.misses, g.slot = r.slot, g.hist_len = r.hlen;
                    g.flags = r.hpos, g.conf = r.conf, g.vx = r.vx, g.vy = r.vy;      // flags carries hpos through the staging only
                    stage[pos] = g;
                }
                lds_sync<MULTIWAVE>();
                T = total;
                if (row < T) {
                    const av_track_row g = stage[row];
                    r.id = g.id, r.x1 = g.x1, r.y1 = g.y1, r.x2 = g.x2, r.y2 = g.y2, r.cls = g.cls;
                    r.age = g.age, r.hits = g.hits, r.misses = g.misses, r.slot = g.slot, r.hlen = g.hist_len;
                    r.hpos = g.flags;
                    r.conf = g.conf, r.vx = g.vx, r.vy = g.vy;
                }
            }
        }

        TRK_STAMP(6)                                        // deaths
        // ---- per-frame outputs --------------------------------------------------------------------
        if (full && !publish_flag) emit(sf);
        lds_sync<MULTIWAVE>();          // sh.d* are rewritten by the next frame
        TRK_STAMP(7)                                        // outputs
    };
    if (PIPE && wid == REP) {
        // ---- the trailing wave: the column waves' barriers of frame f, then frame f - 1 applied to the complete rows ------------------
        int pl = -1, gl = -1;
        for (int f = 0; f <= n_frames; ++f) {
            if (f < n_frames) {
                pl = (pl + 1 == FC) ? 0 : pl + 1;
                if (pl == 0) {                             // chunk hand-over of the column waves (buffer of the other parity)
                    lds_sync<true>();
                    lds_sync<true>();
                }
            }
            lds_sync<true>();                              // frame f's exchange barrier (f == n_frames: the closing one)
            TRK_STAMP(2)
            if (f == 0) continue;
            const int g = f - 1;
            const size_t sg = (size_t)s * n_frames + g;
            gl = (gl + 1 == FC) ? 0 : gl + 1;
            if (gl == 0) cpar ^= 1, set_chunk(chunk + (size_t)cpar * chunk_sz);
            int nd = c_n[gl];
            nd = nd < 0 ? 0 : (nd > dcap ? dcap : nd);
            frame_count += 1;
            if (row < dcap) sh.d2t[row] = -1;
            const int mj = (int)rec[(g & 1) * RB + lane];
            const unsigned long long used = (unsigned long long)(unsigned char)rec[(g & 1) * RB + 64];
            lds_sync<false>();
            TRK_STAMP(3)
            apply(mj, used, nd, c_box + (size_t)gl * dcap * 4, c_cls + (size_t)gl * dcap, c_conf + (size_t)gl * dcap, sg);
        }
    } else
    for (int f = 0; f < n_frames; ++f) {
        const size_t sf = (size_t)s * n_frames + f;
        fl = (fl + 1 == FC) ? 0 : fl + 1;
        if (REPL && det_area) {
            c_n = const_cast<int*>(det_n), c_box = const_cast<int*>(det_box), c_cls = const_cast<int*>(det_cls);
            c_conf = const_cast<double*>(det_conf), c_area = const_cast<double*>(det_area);
        } else if (REPL && fl == 0) {
            // the chunk was fetched into registers while the previous one was being processed (one element per
            // thread: FC * dcap <= 512): hand it to LDS and start fetching the next one
            lds_sync<true>();
            if (PIPE) cpar ^= 1, set_chunk(chunk + (size_t)cpar * chunk_sz);
            if (tid < FC) c_n[tid] = pf_n;
            if (tid < FC * dcap) {
                reinterpret_cast<int4*>(c_box)[tid] = pf_box;
                c_area[tid] = (double)(pf_box.z - pf_box.x) * (double)(pf_box.w - pf_box.y);
                c_cls[tid] = pf_cls, c_conf[tid] = pf_conf;
            }
            lds_sync<true>();
            if (f + FC < n_frames) prefetch(dsf0 + f + FC, (n_frames - f - FC) < FC ? (n_frames - f - FC) : FC);
        } else if (fl == 0) {
            // one cooperative load of the next FC frames' detections; the only global reads of the loop
            const int nfr = (n_frames - f) < FC ? (n_frames - f) : FC;
            lds_sync<MULTIWAVE || REPL>();
            for (int i = tid; i < nfr; i += blockDim.x) c_n[i] = det_n[sf + i];
            const int4* gb = reinterpret_cast<const int4*>(det_box) + sf * dcap;
            for (int i = tid; i < nfr * dcap; i += blockDim.x) {
                const int4 b = gb[i];
                reinterpret_cast<int4*>(c_box)[i] = b;
                c_area[i] = (double)(b.z - b.x) * (double)(b.w - b.y);
            }
            for (int i = tid; i < nfr * dcap; i += blockDim.x) {
                c_cls[i] = det_cls[sf * dcap + i];
                c_conf[i] = det_conf[sf * dcap + i];
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);
            lds_sync<MULTIWAVE || REPL>();
        }
        TRK_STAMP(0)                                        // chunk hand-over
        int nd = c_n[fl];
        const int* dbox = c_box + (size_t)fl * dcap * 4;        // [dcap][4]
        const int* dcls = c_cls + (size_t)fl * dcap;
        const double* dconf = c_conf + (size_t)fl * dcap;
        const double* darea = c_area + (size_t)fl * dcap;
        nd = nd < 0 ? 0 : (nd > dcap ? dcap : nd);
        frame_count += 1;
        if (full) {
            if (row < dcap) sh.d2t[row] = -1;
            lds_sync<MULTIWAVE>();
        }

        // ---- association (multi_object_tracker.py:113-164) ------------------------------------
        unsigned long long used = 0;          // columns already taken; identical in every thread
        int matched_j = -1;
        const bool active = row < T;
        double best = -1.0;
        int best_j = -1;
        double ious[DREG > 0 ? DREG : 1];
        bool need_loop = true;                // run the all-pairs IoU pass + generic greedy arg-max loop below
        if (REPL) {
            // ---- replica waves: wave w owns detection column w --------------------------------------------
            // Same three facts as the single-wave front end below (guard-banded threshold test; with no row
            // holding two candidates the columns are independent; a contested column's winner is its arg-max),
            // with the columns spread over the waves.  Published per column: the candidate bit of every row,
            // the candidate count and the winning row.  After the barrier each wave rebuilds its rows'
            // candidate sets; if some row has two candidates all waves take the generic path (identically).
            const double thr = cfg.iou_threshold;
            const double thr_hi = thr * (1.0 + 0x1p-40), thr_lo = thr * (1.0 - 0x1p-40);
            unsigned char* xb = xchg + (f & 1) * XB;
            bool c = false;
            int pc = 0, winner = 255;
            if (wid < nd) {
                const int j = wid;
                const int4 b = *reinterpret_cast<const int4*>(dbox + j * 4);
                const int xi1 = r.x1 > b.x ? r.x1 : b.x, yi1 = r.y1 > b.y ? r.y1 : b.y;
                const int xi2 = r.x2 < b.z ? r.x2 : b.z, yi2 = r.y2 < b.w ? r.y2 : b.w;
                const int iw = xi2 - xi1, ih = yi2 - yi1;
                const double a1 = (double)(r.x2 - r.x1) * (double)(r.y2 - r.y1);
                const double inter = (double)iw * (double)ih;
                const double uni = a1 + darea[j] - inter;
                const bool valid = active && iw > 0 && ih > 0;      // then uni >= max(a1, a2) > 0 (:95-105)
                c = valid && inter >= thr_hi * uni;
                double v = 0.0;
                bool have_v = false;
                if (__ballot(valid && !c && inter >= thr_lo * uni)) {     // uniform, rare: too close to call
                    v = valid ? inter / uni : 0.0;
                    c = valid && v >= thr;                                   // :147
                    have_v = true;
                }
                const unsigned long long m = __ballot(c);
                pc = __popcll(m);
                if (pc == 1) {
                    winner = __ffsll((long long)m) - 1;
                } else if (pc >= 2) {
                    // Order the candidates by an f32 quotient first: |q/Q - 1| < 2^-21 (two conversions, v_rcp_f32
                    // at 1 ulp, one multiply), so when every other candidate is more than 2^-19 below the largest
                    // q the float64 quotients (:105) are ordered the same way, strictly -- no tie, no divide.
                    const float q = c ? (float)inter * __builtin_amdgcn_rcpf((float)uni) : 0.0f;
                    const unsigned qmax = wave_max_u32(__float_as_uint(q));          // q >= 0: bit patterns order like values
                    const unsigned long long near = __ballot(c && q >= __uint_as_float(qmax) * (1.0f - 0x1p-19f));
                    if (__popcll(near) == 1 && !have_v) {
                        winner = __ffsll((long long)near) - 1;
                    } else {
                        if (!have_v) v = c ? inter / uni : 0.0;              // the reference's float64 divide (:105)
                        winner = __ffsll((long long)wave_argmax_nonneg(v, c)) - 1;
                    }
                }
            }
            xb[lane * 8 + wid] = c ? (unsigned char)(1u << wid) : (unsigned char)0;
            if (lane == 0) {
                xb[512 + wid] = (unsigned char)(pc ? (1u << wid) : 0u);
                xb[520 + wid] = (unsigned char)winner;
            }
            TRK_STAMP(1)                                    // own column tested and published
            lds_sync<true>();
            TRK_STAMP(2)                                    // barrier
            auto orfold = [](unsigned long long x) {
                unsigned t = (unsigned)x | (unsigned)(x >> 32);
                t |= t >> 16;
                t |= t >> 8;
                return t & 0xFFu;
            };
            const unsigned cm = orfold(*reinterpret_cast<const unsigned long long*>(xb + lane * 8));
            const unsigned long long anys = *reinterpret_cast<const unsigned long long*>(xb + 512);
            const unsigned long long wins = *reinterpret_cast<const unsigned long long*>(xb + 520);
            if (__ballot(__popc(cm) >= 2) == 0ull) {
                need_loop = false;
                used = (unsigned long long)__builtin_amdgcn_readfirstlane((int)orfold(anys));
                const bool has = cm != 0u;
                const int my_col = has ? __ffs((int)cm) - 1 : 0;
                if (has && (int)((wins >> (8 * my_col)) & 0xFFull) == lane) matched_j = my_col;
            }
        } else if (DREG > 0 && !MULTIWAVE && cfg.iou_threshold > 0.0) {
            // ---- divide-free front end ---------------------------------------------------------------
            // (1) candidate test: v = fl(inter/uni) >= thr is decided by comparing inter with thr*uni
            //     outside the band thr*(1 +- 2^-40)*uni (the quotient is then further from thr than any
            //     rounding can move it); only columns with a lane inside that band pay for the divide.
            // (2) while no row holds two candidates a row carries ONE edge (its column, inter, uni), and
            //     a row taken by the greedy loop removes nothing from any other column: the columns are
            //     independent, each one's winner is its arg-max (lowest row on ties == first in row-major
            //     order, :150), and a column with a single candidate needs no quotient at all.
            // (3) the contested columns share one exact divide and are resolved by a u32 DPP arg-max each.
            // A frame in which some row has two candidates takes the generic path below instead.
            const double thr = cfg.iou_threshold;
            const double thr_hi = thr * (1.0 + 0x1p-40), thr_lo = thr * (1.0 - 0x1p-40);
            const double a1 = (double)(r.x2 - r.x1) * (double)(r.y2 - r.y1);
            int cnt = 0, my_col = 0;
            double s_inter = 0.0, s_uni = 1.0;
            unsigned long long colcnt4 = 0;                       // per column: candidate count, 4 bits, saturating
            unsigned anyc = 0;                                    // columns with any candidate
#pragma unroll 1
            for (int j = 0; j < nd; ++j) {
                const int4 b = *reinterpret_cast<const int4*>(dbox + j * 4);
                const int xi1 = r.x1 > b.x ? r.x1 : b.x, yi1 = r.y1 > b.y ? r.y1 : b.y;
                const int xi2 = r.x2 < b.z ? r.x2 : b.z, yi2 = r.y2 < b.w ? r.y2 : b.w;
                const int iw = xi2 - xi1, ih = yi2 - yi1;
                const double inter = (double)iw * (double)ih;
                const double uni = a1 + darea[j] - inter;
                // iw, ih > 0 implies both areas > 0 and uni >= max(a1, a2) > 0 (:95-105)
                const bool valid = active && iw > 0 && ih > 0;
                bool c = valid && inter >= thr_hi * uni;
                if (__ballot(valid && !c && inter >= thr_lo * uni)) {     // uniform, rare: too close to call
                    const double v = valid ? inter / uni : 0.0;
                    c = valid && v >= thr;                                   // :147
                }
                const int pc = __popcll(__ballot(c));
                colcnt4 |= (unsigned long long)(pc > 15 ? 15 : pc) << (4 * j);
                anyc |= (pc ? 1u : 0u) << j;
                cnt += c ? 1 : 0;
                my_col = c ? j : my_col;
                s_inter = c ? inter : s_inter;
                s_uni = c ? uni : s_uni;
            }
            if (__ballot(cnt >= 2) == 0ull) {
                need_loop = false;
                used = anyc;                                          // every column with a candidate gets matched
                const int mycnt = (int)((colcnt4 >> (4 * my_col)) & 15ull);
                const bool has = cnt == 1;
                if (has && mycnt == 1) matched_j = my_col;
                const bool fights = has && mycnt >= 2;
                unsigned long long rem = __ballot(fights);
                if (rem) {
                    const double v = fights ? s_inter / s_uni : 0.0;  // the reference's one float64 divide (:105)
                    while (rem) {
                        const int j = __builtin_amdgcn_readlane(my_col, __ffsll((long long)rem) - 1);
                        const bool in = fights && my_col == j;
                        const unsigned long long bal = wave_argmax_nonneg(v, in);
                        if (lane == __ffsll((long long)bal) - 1) matched_j = j;
                        rem &= ~__ballot(in);
                    }
                }
            }
        }
        if (need_loop && DREG > 0) {
            // all detection boxes of the frame in one batch of LDS reads (uniform addresses), then a
            // branch-free IoU pass: slots >= nd and rows >= T get -1
            int4 db[DREG > 0 ? DREG : 1];
#pragma unroll
            for (int j = 0; j < (DREG > 0 ? DREG : 1); ++j)
                db[j] = *reinterpret_cast<const int4*>(dbox + (j < dcap ? j : 0) * 4);
            const double a1 = (double)(r.x2 - r.x1) * (double)(r.y2 - r.y1);
#pragma unroll
            for (int j = 0; j < (DREG > 0 ? DREG : 1); ++j) {
                const int xi1 = r.x1 > db[j].x ? r.x1 : db[j].x, yi1 = r.y1 > db[j].y ? r.y1 : db[j].y;
                const int xi2 = r.x2 < db[j].z ? r.x2 : db[j].z, yi2 = r.y2 < db[j].w ? r.y2 : db[j].w;
                const int iw = xi2 - xi1, ih = yi2 - yi1;
                const double inter = (double)iw * (double)ih;
                const double a2 = (double)(db[j].z - db[j].x) * (double)(db[j].w - db[j].y);
                const double uni = a1 + a2 - inter;
                double v = (iw > 0 && ih > 0 && uni > 0.0) ? inter / uni : 0.0;     // :95-105
                v = (v >= cfg.iou_threshold) ? v : -1.0;                              // :147  never picked below thr
                ious[j] = (active && j < nd) ? v : -1.0;
            }
        }
        auto recompute = [&]() {
            best = -1.0;
            best_j = -1;
            if (DREG > 0) {
#pragma unroll
                for (int j = 0; j < (DREG > 0 ? DREG : 1); ++j) {
                    const double v = ((used >> j) & 1ull) ? -1.0 : ious[j];
                    if (v > best) best = v, best_j = j;
                }
                if (best < 0.0) best_j = -1;
                return;
            }
            for (int j = 0; j < nd; ++j) {
                if ((used >> j) & 1ull) continue;
                const double v = iou_f64(r.x1, r.y1, r.x2, r.y2, dbox[j * 4 + 0], dbox[j * 4 + 1], dbox[j * 4 + 2], dbox[j * 4 + 3]);
                if (v > best) best = v, best_j = j;
            }
            if (!(best >= cfg.iou_threshold)) best = -1.0, best_j = -1;   // :147  max < thr -> stop
        };
        if (need_loop && active && matched_j < 0 && nd > 0) recompute();
        if (need_loop && T > 0 && nd > 0) {
            const int max_iter = T < nd ? T : nd;
            for (int it = 0; it < max_iter; ++it) {
                const double key = (active && matched_j < 0 && best_j >= 0) ? best : -1.0;
                double wmax = -1.0;
                unsigned long long bal;
                if (MULTIWAVE) {
                    wmax = wave_max(key);
                    bal = __ballot(key == wmax && key >= 0.0);
                } else {
                    bal = wave_argmax_nonneg(key, key >= 0.0);
                }
                int win_row = -1, win_col = -1;
                double win_iou = -1.0;
                if (bal) {
                    const int leader = __ffsll((long long)bal) - 1;
                    win_row = (REPL ? 0 : (wid << 6)) + leader;
                    win_col = __shfl(best_j, leader, 64);
                    win_iou = wmax;
                }
                if (MULTIWAVE) {
                    if (lane == 0) sh.w_iou[wid] = win_iou, sh.w_row[wid] = win_row, sh.w_col[wid] = win_col;
                    lds_sync<MULTIWAVE>();
                    win_iou = -1.0, win_row = -1, win_col = -1;
                    for (int w = 0; w < nwaves; ++w)          // ascending wave == ascending row
                        if (sh.w_iou[w] > win_iou) win_iou = sh.w_iou[w], win_row = sh.w_row[w], win_col = sh.w_col[w];
                    lds_sync<MULTIWAVE>();
                }
                if (win_row < 0) break;
                used |= 1ull << win_col;
                if (row == win_row) matched_j = win_col;
                else if (active && matched_j < 0 && best_j == win_col) recompute();
            }
        }

        TRK_STAMP(3)                                        // association resolved
        if (PIPE && wid == 0) {                             // the frame's decisions, for the trailing wave
            rec[(f & 1) * RB + lane] = (signed char)matched_j;
            if (lane == 0) rec[(f & 1) * RB + 64] = (signed char)(unsigned char)used;
        }
        apply(matched_j, used, nd, dbox, dcls, dconf, sf);
    }
    if (PIPE && wid < REP) lds_sync<true>();               // the closing barrier: the last frame's decisions are published
#undef TRK_STAMP
    if (TIMED && s == 0 && lane == 0 && det2trk) {
        __builtin_amdgcn_s_waitcnt(0);
        for (int k = 0; k < 8; ++k) det2trk[wid * 8 + k] = (int)tacc[k];
    }

    // ---- persist ----------------------------------------------------------------------------------
    if (!full) return;
    if (row < T) {
        av_track_row g;
        g.id = r.id, g.x1 = r.x1, g.y1 = r.y1, g.x2 = r.x2, g.y2 = r.y2, g.cls = r.cls;
        g.age = r.age, g.hits = r.hits, g.misses = r.misses, g.slot = r.slot, g.hist_len = r.hlen;
        g.flags = (r.hits >= cfg.min_hits) ? 1 : 0;
        g.conf = r.conf, g.vx = r.vx, g.vy = r.vy;
        if (coherent_out) {
            unsigned long long w[sizeof(av_track_row) / 8];
            __builtin_memcpy(w, &g, sizeof(g));
            unsigned long long* d = reinterpret_cast<unsigned long long*>(rows + row);
#pragma unroll
            for (int i = 0; i < (int)(sizeof(av_track_row) / 8); ++i) __hip_atomic_store(d + i, w[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            rows[row] = g;
        }
    }
    if (row == 0) {
        if (coherent_out) {
            const int hv[4] = {T, next_id, frame_count, status};
#pragma unroll
            for (int i = 0; i < 4; ++i) __hip_atomic_store(hdr + i, hv[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            hdr[0] = T, hdr[1] = next_id, hdr[2] = frame_count, hdr[3] = status;
        }
    }
    if (publish_flag) {
        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): this wave's stores (history rings, rows, header) are acknowledged
        if (lane == 0) {
            if (publish_clock)
                __hip_atomic_store(reinterpret_cast<unsigned long long*>(publish_flag + 2), __builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(publish_flag, publish_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        emit((size_t)s * n_frames + (n_frames - 1));
    }
}

template <bool MULTIWAVE, int DREG, int REP, bool TIMED = false, int PIPE = 0>
__global__ void __launch_bounds__(1024) tracker_kernel(av_tracker_cfg cfg, int n_frames, int dcap, const int32_t* __restrict__ det_n,
                               const int32_t* __restrict__ det_box, const int32_t* __restrict__ det_cls,
                               const double* __restrict__ det_conf, int tcap, unsigned char* __restrict__ state_all,
                               av_track_row* __restrict__ snap, int32_t* __restrict__ snap_n,
                               int32_t* __restrict__ det2trk, int chunk_frames, unsigned long long wave_map) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    tracker_body<MULTIWAVE, DREG, REP, TIMED, PIPE>(cfg, n_frames, dcap, det_n, det_box, det_cls, det_conf, tcap, state_all, snap, snap_n,
                                              det2trk, chunk_frames, blockIdx.x, smem, wave_map);
}

__global__ void tracker_reset_kernel(int n_streams, size_t bytes_per_stream, unsigned char* state) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_streams) return;
    int* hdr = reinterpret_cast<int*>(state + (size_t)s * bytes_per_stream);
    for (int i = 0; i < HDR_INTS; ++i) hdr[i] = 0;
    hdr[1] = 1;     // next_id starts at 1 (multi_object_tracker.py:81)
}

// the windowed replica kernel (eight column waves, + the trailing wave with PIPE); more than 64 KB of dynamic LDS needs the attribute
template <bool TIMED, int PIPE>
int launch_replicas(int n_streams, size_t lds, hipStream_t st, const av_tracker_cfg& cfg, int n_frames, int dcap, const int32_t* det_n,
                    const int32_t* det_box, const int32_t* det_cls, const double* det_conf, int tcap, unsigned char* state,
                    av_track_row* snap, int32_t* snap_n, int32_t* det2trk, int fc) {
    // per device (the attribute belongs to the device's copy of the function): once for each device this process launches on
    static std::mutex mu;
    static unsigned long long devs_done = 0;
    int dev = 0;
    AV_HIP(hipGetDevice(&dev));
    {
        std::lock_guard<std::mutex> lk(mu);
        if (dev < 0 || dev >= 64 || !((devs_done >> dev) & 1ull)) {
            AV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(tracker_kernel<false, 8, 8, TIMED, PIPE>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            if (dev >= 0 && dev < 64) devs_done |= 1ull << dev;
        }
    }
    // hardware wave -> role.  SIMD of hardware wave h = h % 4.  With the trailing wave: SIMD 0 = column 0 + columns 6, 7; SIMD 1 = the
    // trailing wave + column 3; SIMD 2 = columns 1, 4; SIMD 3 = columns 2, 5 (0.268 -> 0.260 ms per 64 x 256 frames).
    // AVHOT_TRACKER_MAP=0: identity
    const char* map_env = getenv("AVHOT_TRACKER_MAP");
    unsigned long long wave_map = PIPE == 1 ? 0x754362180ull : 0xFEDCBA9876543210ull;
    if (map_env && atoi(map_env) == 0) wave_map = 0xFEDCBA9876543210ull;
    hipLaunchKernelGGL((tracker_kernel<false, 8, 8, TIMED, PIPE>), dim3(n_streams), dim3(64 * (8 + PIPE)), lds, st, cfg, n_frames,
                       dcap, det_n, det_box, det_cls, det_conf, tcap, state, snap, snap_n, det2trk, fc, wave_map);
    return AV_OK;
}

}  // namespace

#ifndef AVHOT_DEVICE_ONLY      // (step.hip includes this file for its device code only)

extern "C" {

size_t av_tracker_state_bytes(int tcap, int trajectory_length) {
    if (tcap <= 0 || trajectory_length <= 0) return 0;
    return state_bytes(tcap, trajectory_length);
}

int av_tracker_reset(av_ctx* ctx, av_stream_t stream, int n_streams, int tcap, int trajectory_length, void* state) {
    AV_REQUIRE(ctx && state, AV_EINVAL, "av_tracker_reset: null argument");
    AV_REQUIRE(n_streams > 0 && tcap > 0 && trajectory_length > 0, AV_EINVAL, "av_tracker_reset: bad sizes");
    hipLaunchKernelGGL(tracker_reset_kernel, dim3((n_streams + 63) / 64), dim3(64), 0, as_stream(stream), n_streams,
                       state_bytes(tcap, trajectory_length), (unsigned char*)state);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

int av_tracker_update(av_ctx* ctx, av_stream_t stream, const av_tracker_cfg* cfg, int n_streams, int n_frames,
                      int dcap, const int32_t* det_n, const int32_t* det_box, const int32_t* det_cls,
                      const double* det_conf, int tcap, void* state, av_track_row* snap, int32_t* snap_n,
                      int32_t* det2trk) {
    AV_REQUIRE(ctx && cfg && det_n && det_box && det_cls && det_conf && state, AV_EINVAL,
               "av_tracker_update: null argument");
    AV_REQUIRE(n_streams > 0 && n_frames > 0, AV_EINVAL, "av_tracker_update: n_streams/n_frames must be > 0");
    AV_REQUIRE(tcap >= 64 && tcap <= 1024 && (tcap % 64) == 0, AV_EINVAL,
               "av_tracker_update: tcap %d must be a multiple of 64 in [64,1024]", tcap);
    AV_REQUIRE(dcap >= 1 && dcap <= 64, AV_EINVAL, "av_tracker_update: dcap %d not in [1,64]", dcap);
    AV_REQUIRE(cfg->trajectory_length >= 1, AV_EINVAL, "av_tracker_update: trajectory_length must be >= 1");
    AV_REQUIRE(cfg->iou_threshold >= 0.0, AV_EINVAL,
               "av_tracker_update: iou_threshold < 0 never terminates in the reference either");
    AV_REQUIRE((snap == nullptr) == (snap_n == nullptr), AV_EINVAL, "av_tracker_update: snap and snap_n go together");
    // chunk of frames whose detections are staged in LDS at once (~16 KB)
    const size_t per_frame = 4 + (size_t)dcap * (16 + 4 + 8 + 8);
    int fc = (int)(16384 / per_frame);
    fc = fc < 1 ? 1 : (fc > 64 ? 64 : fc);
    if (fc > n_frames) fc = n_frames;
    const size_t chunk_bytes = (size_t)((fc + 3) & ~3) * 4 + (size_t)fc * dcap * 16 + (((size_t)fc * dcap + 1) & ~size_t(1)) * 4 +
                               (size_t)fc * dcap * 16 + 16;
    // replica waves (one per detection column) for the common table size; AVHOT_TRACKER_REP=1 keeps one wave
    constexpr int REPW = 8;
    const char* rep_env = getenv("AVHOT_TRACKER_REP");
    const bool rep = tcap == 64 && dcap <= REPW && cfg->iou_threshold > 0.0 && !(rep_env && atoi(rep_env) <= 1);
    const size_t rep_bytes = ((sizeof(Shared) + 63) & ~size_t(63)) + (size_t)tcap * sizeof(av_track_row);
    const size_t lds = rep_bytes * (rep ? REPW : 1) + chunk_bytes + (rep ? 2 * 576 : 0);
#define AV_TRK_LAUNCH(MW, DR, RP)                                                                                  \
    hipLaunchKernelGGL((tracker_kernel<MW, DR, RP>), dim3(n_streams), dim3(tcap * RP), lds, as_stream(stream), *cfg,   \
                       n_frames, dcap, det_n, det_box, det_cls, det_conf, tcap, (unsigned char*)state, snap, snap_n, \
                       det2trk, fc, 0xFEDCBA9876543210ull)
    if (rep) {
        // A window of frames is a long per-stream latency chain on one CU per stream; whatever else lands on that CU competes
        // with it for issue slots and for the CU's memory pipeline.  In the batched step the HBM-bound planner runs on the
        // other stream: claiming (almost) the whole LDS keeps its workgroups (38 KB each) off the tracker's 64 CUs -- config 4
        // 0.380 -> 0.352 ms per step (100 KB: no change, two planner workgroups still fit; 125 / 150 KB: 0.357 / 0.352).
        // 156 KB also keeps the Kalman kernel's one-wave workgroups (8.4 KB of LDS each) off these CUs: 0.306-0.310 against
        // 0.310-0.313 ms per step.
        // AVHOT_TRACKER_LDS_KB overrides (0 = only what the kernel needs).
        // Windows also get the trailing ninth wave (PIPE, see tracker_body); AVHOT_TRACKER_PIPE=0 / 1 overrides.
        const char* pad_env = getenv("AVHOT_TRACKER_LDS_KB");
        const char* pipe_env = getenv("AVHOT_TRACKER_PIPE");
        const int pipe = pipe_env ? (atoi(pipe_env) != 0 ? 1 : 0) : (n_frames >= 16 ? 1 : 0);
        const bool timed = getenv("AVHOT_TRACKER_TIMED") != nullptr;
        const size_t want = pad_env ? (size_t)atoi(pad_env) * 1024 : (n_frames >= 16 ? (size_t)156 * 1024 : 0);
        const size_t lds_need = pipe ? rep_bytes * (REPW + pipe) + 2 * chunk_bytes + 2 * 576 + 2 * 80 + 16 : lds;
        AV_REQUIRE(lds_need <= 160 * 1024, AV_EINVAL, "av_tracker_update: %zu bytes of LDS", lds_need);
        const size_t lds_use = want > lds_need && want <= 160 * 1024 ? want : lds_need;
        int rc;
#define AV_TRK_REP(TM, PP)                                                                                                            \
    launch_replicas<TM, PP>(n_streams, lds_use, as_stream(stream), *cfg, n_frames, dcap, det_n, det_box, det_cls, det_conf, tcap, \
                            (unsigned char*)state, snap, snap_n, det2trk, fc)
        if (pipe == 1) rc = timed ? AV_TRK_REP(true, 1) : AV_TRK_REP(false, 1);
        else rc = timed ? AV_TRK_REP(true, 0) : AV_TRK_REP(false, 0);
#undef AV_TRK_REP
        if (rc != AV_OK) return rc;
    } else if (tcap == 64) {
        if (dcap <= 8) AV_TRK_LAUNCH(false, 8, 1);
        else if (dcap <= 16) AV_TRK_LAUNCH(false, 16, 1);
        else AV_TRK_LAUNCH(false, 0, 1);
    } else {
        if (dcap <= 8) AV_TRK_LAUNCH(true, 8, 1);
        else AV_TRK_LAUNCH(true, 0, 1);
    }
#undef AV_TRK_LAUNCH
    AV_LAUNCH_CHECK();
    return AV_OK;
}

}  // extern "C"

#endif  // AVHOT_DEVICE_ONLY
