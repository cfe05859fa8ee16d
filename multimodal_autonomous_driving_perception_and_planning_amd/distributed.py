"""Multi-GPU: stream sharding + all-gather of per-frame track tables.

The hot path shards by video stream: every object in the loop is per-stream state and there is
no cross-stream term, so rank g of G owns streams [g*S/G, (g+1)*S/G) and no
data-path collective is needed for correctness.  The one exchange BASELINE.json asks for is an
all-gather of the track tables (a fleet-wide view for downstream consumers such as the reference's
InteractionDetector); the reference itself has no counterpart.

One process per GPU, torch.distributed backend "nccl" (= RCCL over xGMI) on the GPU box, "gloo"
in the CPU tests.  Tables travel in the 32-byte-per-row wire format of include/avhot.h
(av_wire_hdr / av_wire_row), packed on the device by av_pack_tracks; the gather runs on its own
stream, double-buffered, so window k's exchange overlaps window k+1's kernels.

Two modes (bytes per rank per step at S = 64 streams, tcap = 64, 2064 B per table):
  window-end  one table per stream per step: the table after the window's last frame        132 KB
  per-frame   every frame's table of the window (BASELINE config 5's wording); with W = 256  33.8 MB,
              with W = 1 (one step per time-step) the two modes coincide                     132 KB
With W = 1 and the one-launch step (av_hot_step) the step kernel writes the wire tables itself, straight into the send buffer
(TrackTableExchange.begin_step() before the step, exchange() after it): per-frame tables, no pack launch.
"""
import ctypes as C
import os

import numpy as np
import torch
import torch.distributed as dist

from . import _native as nat

WIRE_HDR_BYTES, WIRE_ROW_BYTES = 16, 32
WIRE_HDR_FIELDS = [("n_rows", "<i4"), ("stream", "<i4"), ("frame", "<i4"), ("reserved", "<i4")]
WIRE_ROW_FIELDS = [("id", "<i4"), ("x1", "<i2"), ("y1", "<i2"), ("x2", "<i2"), ("y2", "<i2"), ("age", "<i4"),
                   ("hits", "<i4"), ("misses", "<u2"), ("cls", "u1"), ("flags", "u1"), ("conf", "<f4"),
                   ("vx2", "<i2"), ("vy2", "<i2")]


def shard_streams(n_streams_total, world, rank):
    """Contiguous block partition; the first (n % world) ranks take one extra stream."""
    base, extra = divmod(n_streams_total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def wire_table_bytes(tcap):
    return WIRE_HDR_BYTES + tcap * WIRE_ROW_BYTES


def pack_wire(snap_u8, snap_n, frame_lo, n_sel, stream0=0, frame0=0, frame_count=None):
    """Host/torch statement of av_pack_tracks (the HIP kernel is checked against it bit for bit; the CPU tests and
    CPU-tensor exchanges use it directly).  snap_u8 [S, W, tcap, 64] uint8, snap_n [S, W] int32 ->
    [S, n_sel, wire_table_bytes(tcap)] uint8.  header.frame = frame0 + the stream's detector frame count at that frame when
    `frame_count` [S] (the counters after the window's last frame) is given, else frame0 + the index within the window."""
    S, W, tcap, _ = snap_u8.shape
    sel = snap_u8[:, frame_lo:frame_lo + n_sel].contiguous()
    n = snap_n[:, frame_lo:frame_lo + n_sel].to(torch.int32).clamp(max=tcap).contiguous()
    w = sel.view(torch.int32).reshape(S, n_sel, tcap, 16)
    live = (torch.arange(tcap, device=sel.device).view(1, 1, tcap) < n.unsqueeze(-1))
    conf = sel.view(torch.float64).reshape(S, n_sel, tcap, 8)[..., 6].to(torch.float32)
    vel = sel.view(torch.float32).reshape(S, n_sel, tcap, 16)[..., 14:16]
    out = torch.zeros(S, n_sel, tcap, WIRE_ROW_BYTES, dtype=torch.uint8, device=sel.device)

    def put(lo, t):
        t = torch.where(live.view(S, n_sel, tcap, *([1] * (t.dim() - 3))), t, torch.zeros_like(t)).contiguous()
        b = t.view(torch.uint8).reshape(S, n_sel, tcap, -1)
        out[..., lo:lo + b.shape[-1]] = b
    put(0, w[..., 0:1])                                             # id
    put(4, w[..., 1:5].to(torch.int16))                             # box
    put(12, w[..., 6:8])                                            # age, hits
    put(20, w[..., 8:9].clamp(max=65535).to(torch.int32).to(torch.int16))   # misses (two's complement of u16)
    put(22, w[..., 5:6].to(torch.uint8))                            # cls
    put(23, w[..., 11:12].to(torch.uint8))                          # flags
    put(24, conf.unsqueeze(-1))
    put(28, (vel * 2.0).to(torch.int16))
    hdr = torch.zeros(S, n_sel, 4, dtype=torch.int32, device=sel.device)
    hdr[..., 0] = n
    hdr[..., 1] = stream0 + torch.arange(S, device=sel.device, dtype=torch.int32).view(S, 1)
    hdr[..., 2] = frame0 + frame_lo + torch.arange(n_sel, device=sel.device, dtype=torch.int32).view(1, n_sel)
    if frame_count is not None:
        hdr[..., 2] += frame_count.to(device=sel.device, dtype=torch.int32).view(S, 1) - (W - 1)
    msg = torch.empty(S, n_sel, wire_table_bytes(tcap), dtype=torch.uint8, device=sel.device)
    msg[..., :WIRE_HDR_BYTES] = hdr.view(torch.uint8).reshape(S, n_sel, WIRE_HDR_BYTES)
    msg[..., WIRE_HDR_BYTES:] = out.reshape(S, n_sel, tcap * WIRE_ROW_BYTES)
    return msg


def unpack_wire(msg, tcap):
    """Gathered [..., wire_table_bytes(tcap)] uint8 (tensor or array) -> (hdr structured [...], rows structured [..., tcap])."""
    m = msg.cpu().numpy() if isinstance(msg, torch.Tensor) else np.asarray(msg)
    lead = m.shape[:-1]
    m = np.ascontiguousarray(m.reshape(-1, wire_table_bytes(tcap)))
    hdr = np.ascontiguousarray(m[:, :WIRE_HDR_BYTES]).view(np.dtype(WIRE_HDR_FIELDS)).reshape(lead)
    rows = np.ascontiguousarray(m[:, WIRE_HDR_BYTES:]).view(np.dtype(WIRE_ROW_FIELDS)).reshape(lead + (tcap,))
    return hdr, rows


# ---- the 64-byte-row message of round 1 (kept: the class API below no longer uses it) ------------------
def pack_tables(rows_u8, counts):
    """[S, tcap, 64] uint8 rows + [S] int32 counts -> one [S, tcap*64 + 64] uint8 message."""
    S, tcap, rb = rows_u8.shape
    msg = torch.zeros(S, tcap * rb + 64, dtype=torch.uint8, device=rows_u8.device)
    msg[:, :tcap * rb] = rows_u8.reshape(S, tcap * rb)
    msg[:, tcap * rb:tcap * rb + 4] = counts.to(torch.int32).contiguous().view(torch.uint8).reshape(S, 4)
    return msg


def unpack_tables(msg, tcap):
    """Inverse of pack_tables on a gathered [world*S, tcap*64+64] uint8 array (host)."""
    m = msg.cpu().numpy() if isinstance(msg, torch.Tensor) else np.asarray(msg)
    rows = np.ascontiguousarray(m[:, :tcap * 64]).view(np.dtype(nat.TRACK_ROW_FIELDS)).reshape(m.shape[0], tcap)
    counts = np.ascontiguousarray(m[:, tcap * 64:tcap * 64 + 4]).view(np.int32).reshape(-1)
    return rows, counts


def all_gather_tables(msg, world, group=None):
    """Blocking all-gather of a packed message; returns [world*S, ...]."""
    out = torch.empty((world * msg.shape[0],) + tuple(msg.shape[1:]), dtype=msg.dtype, device=msg.device)
    dist.all_gather_into_tensor(out, msg.contiguous(), group=group)
    return out


class TrackTableExchange:
    """Per step: pack this rank's track tables into the wire format and all-gather them over all ranks.

    `loop` provides S, W, tcap, dev, snap [S,W,tcap,64] u8, snap_n [S,W] i32 and -- on a GPU -- `stream`
    (a torch.cuda.Stream the step was enqueued on) and `ctx` (av_ctx).  With CPU tensors (gloo tests) the same
    code path runs synchronously with the torch statement of the pack kernel.
    per_frame=False: the end-of-window table of every stream; True: all W tables of the window."""

    def __init__(self, loop, world, rank, group=None, per_frame=False, native=None, bucket=1):
        """native=True (or AVHOT_NATIVE_ALLGATHER=1): the gather is the library call av_allgather_tracks on an RCCL
        communicator of its own (made from a ncclUniqueId that rank 0 broadcasts through torch.distributed) instead of
        torch.distributed.all_gather_into_tensor; GPU tensors only.
        bucket=k > 1 (window 1 with the one-launch step only): the tables of k consecutive time-steps travel in ONE all-gather --
        every frame's table is still gathered, k steps at a time.  A 13-us time-step cannot wait for a collective of its own
        (an 8-rank all-gather of 132 KB costs more than the step); k tables per stream per message amortise it.  The step
        kernel writes step t's tables into slot t % k of the send buffer; exchange() gathers when the bucket is full (and
        flush() what a run leaves in a partial one).  latest() then returns hdr [world*S, k], rows [world*S, k, tcap]."""
        self.loop, self.world, self.rank, self.group, self.per_frame = loop, world, rank, group, per_frame
        S, tcap, dev = loop.S, loop.tcap, loop.dev
        self.bucket = int(bucket)
        if self.bucket < 1:
            raise ValueError("bucket must be >= 1")
        if self.bucket > 1 and not (bool(getattr(loop, "fused_step", False)) and loop.W == 1):
            raise ValueError("bucket > 1 needs window 1 with the one-launch step (the step kernel writes the wire tables)")
        self.n_sel = (loop.W if per_frame else 1) * self.bucket
        self.frame_lo = 0 if per_frame else loop.W - 1
        tb = wire_table_bytes(tcap)
        self.bytes_per_step = S * (loop.W if per_frame else 1) * tb
        self.bytes_per_gather = S * self.n_sel * tb
        self.gpu = torch.device(dev).type == "cuda"
        # bucket 1: [S][n_sel][tb] per rank (stream-major, as av_pack_tracks writes it); bucket k: [k][S][tb] (slot-major: the step
        # kernel writes a whole [S][tb] slot per time-step) -- recv is the ranks' send buffers one after the other either way
        shape = (S, self.n_sel, tb) if self.bucket == 1 else (self.bucket, S, tb)
        self.send = [torch.zeros(*shape, dtype=torch.uint8, device=dev) for _ in range(2)]
        self.recv = [torch.zeros(world * shape[0], *shape[1:], dtype=torch.uint8, device=dev) for _ in range(2)]
        if native is None:
            native = os.environ.get("AVHOT_NATIVE_ALLGATHER", "0") == "1"
        if native and not self.gpu:
            import warnings
            warnings.warn("TrackTableExchange(native=True) needs device tensors (av_allgather_tracks is an RCCL call); "
                          "this loop's tables are on the CPU, so the gather goes through torch.distributed instead",
                          RuntimeWarning, stacklevel=2)
        self.native = bool(native) and self.gpu
        self.nccl = None
        if self.gpu:
            self.comm = torch.cuda.Stream(device=dev)
            self.ready = [torch.cuda.Event() for _ in range(2)]
        if self.native:
            L = nat.lib()
            uid = (C.c_ubyte * 128)()
            if rank == 0:
                nat.check(L.av_comm_unique_id(uid))
            if world > 1:
                box = [bytes(uid)]
                dist.broadcast_object_list(box, src=0, group=group)
                uid = (C.c_ubyte * 128).from_buffer_copy(box[0])
            h = C.c_void_p()
            nat.check(L.av_comm_create(loop.ctx.handle, uid, rank, world, C.byref(h)))
            self.nccl = h
        self.done = [None, None]
        self._last = 0                  # receive buffer of the most recent gather
        self.k = 0                      # steps exchanged so far; step k uses buffer (k // bucket) & 1, slot k % bucket
        # window 1 with the one-launch step (HotLoop.fused_step): the step kernel itself writes the wire tables, straight into
        # the send buffer handed to it by begin_step() -- no pack launch between the step and the gather
        self.prepacked = bool(getattr(loop, "fused_step", False)) and loop.W == 1
        self._handed = None             # the send buffer begin_step() handed to the loop for the step being enqueued

    def _loop_streams(self):
        """The torch stream(s) the loop's steps run on (HotLoop(overlap=2) alternates between two)."""
        return list(getattr(self.loop, "_pstreams", None) or [self.loop.stream])

    def step_bucket(self, z_steps=None):
        """HotLoop(overlap=2) only: ONE bucket -- `bucket` consecutive time-steps enqueued by one library call
        (HotLoop.enqueue_steps), every step's wire tables written by the step kernels straight into the send buffer -- and its
        all-gather.  Replaces `bucket` rounds of begin_step() / enqueue_step() / exchange()."""
        loop = self.loop
        if getattr(loop, "overlap", 1) < 2 or not self.prepacked:
            raise RuntimeError("step_bucket needs HotLoop(window=1, overlap=2..)")
        if self.k % self.bucket:
            raise RuntimeError("step_bucket: a bucket is being filled step by step (%d of %d)" % (self.k % self.bucket, self.bucket))
        b = (self.k // self.bucket) & 1
        if self.gpu and self.done[b] is not None:          # the gather that last read this send buffer
            for st in self._loop_streams():
                st.wait_event(self.done[b])
        loop.set_wire(None, stream0=self.rank * loop.S, frame0=0)
        loop.enqueue_steps(self.bucket, z_steps=z_steps, wire_steps=self.send[b])
        self.k += self.bucket
        return self._gather(b)

    def begin_step(self):
        """Call BEFORE enqueuing step k when `prepacked`: hands send buffer k & 1 to the loop (after the gather that last read it).
        A step enqueued without it is still exchanged correctly -- exchange() then packs the tables itself."""
        if not self.prepacked:
            return
        b, slot = (self.k // self.bucket) & 1, self.k % self.bucket
        if self.gpu and self.done[b] is not None and slot == 0:
            for st in self._loop_streams():
                st.wait_event(self.done[b])
        wire = self.send[b].view(self.loop.S, -1) if self.bucket == 1 else self.send[b][slot]
        self.loop.set_wire(wire, stream0=self.rank * self.loop.S, frame0=0)
        self._handed = (b, slot)

    def exchange(self):
        """Enqueue pack + all-gather of the step just enqueued on loop.stream; returns the receive buffer
        ([world*S, n_sel, table bytes], complete once synchronize() / latest() returns)."""
        b, slot = (self.k // self.bucket) & 1, self.k % self.bucket
        loop = self.loop
        # header.frame: the stream's detector frame count at that frame when the loop keeps the counters (HotLoop does -- the value
        # the one-launch step stamps as well), else the number of frames exchanged before it
        fcount = getattr(loop, "frame_count", None)
        frame0 = 0 if fcount is not None else self.k * loop.W
        # prepacked: the step just enqueued wrote its wire tables into send[b] -- if begin_step() handed it that buffer.  If the
        # caller skipped begin_step() the tables are packed here like in any other mode (never a stale or empty buffer).
        want_ptr = self.send[b].data_ptr() if self.bucket == 1 else self.send[b][slot].data_ptr()
        prepacked = self.prepacked and self._handed == (b, slot) and getattr(loop, "wire", None) is not None \
            and loop.wire.data_ptr() == want_ptr
        self._handed = None
        if self.prepacked:
            loop.set_wire(None)                    # a later step without begin_step() must not write into a buffer being gathered
        if self.bucket > 1:
            if not prepacked:
                raise RuntimeError("TrackTableExchange(bucket=%d): call begin_step() before every step" % self.bucket)
            self.k += 1
            if slot + 1 < self.bucket:
                return self.recv[(b + 1) & 1]      # the bucket is still filling: the last complete gather
            return self._gather(b)
        if not self.gpu:
            if not prepacked:
                self.send[b].copy_(pack_wire(loop.snap, loop.snap_n, self.frame_lo, self.n_sel, self.rank * loop.S, frame0, fcount))
            self.k += 1
            return self._gather(b)
        if not prepacked:
            if self.done[b] is not None:           # buffer b is still being sent from two steps ago
                loop.stream.wait_event(self.done[b])
            nat.check(nat.lib().av_pack_tracks(loop.ctx.handle, nat.stream_handle(loop.stream), loop.S, loop.W, loop.tcap,
                                               self.frame_lo, self.n_sel, self.rank * loop.S, frame0, nat.ptr(loop.snap),
                                               nat.ptr(loop.snap_n), nat.ptr(fcount), nat.ptr(self.send[b])))
        self.k += 1
        return self._gather(b)

    def _gather(self, b):
        """Enqueue the all-gather of send buffer b (complete on loop.stream) on the communication stream."""
        if not self.gpu:
            dist.all_gather_into_tensor(self.recv[b], self.send[b], group=self.group)
            self._last = b
            return self.recv[b]
        streams = self._loop_streams()
        self.ready[b].record(streams[0])
        with torch.cuda.stream(self.comm):
            self.comm.wait_event(self.ready[b])
            for st in streams[1:]:                 # (overlap=2: the bucket's steps ran on both streams)
                ev2 = torch.cuda.Event()
                ev2.record(st)
                self.comm.wait_event(ev2)
            if self.native:
                nat.check(nat.lib().av_allgather_tracks(self.loop.ctx.handle, self.nccl, nat.stream_handle(self.comm),
                                                        nat.ptr(self.send[b]), nat.ptr(self.recv[b]), self.bytes_per_gather))
            else:
                dist.all_gather_into_tensor(self.recv[b], self.send[b], group=self.group)
            ev = torch.cuda.Event()
            ev.record(self.comm)
            self.done[b] = ev
        self._last = b
        return self.recv[b]

    def flush(self):
        """bucket > 1: gather a partially filled bucket (its unfilled slots still hold the tables of two buckets ago)."""
        slot = self.k % self.bucket
        if self.bucket > 1 and slot != 0:
            b = (self.k // self.bucket) & 1
            self.k += self.bucket - slot
            return self._gather(b)
        return None

    def latest(self):
        """Host view (hdr [world*S, n_sel], rows [world*S, n_sel, tcap]) of the most recent gather."""
        self.synchronize()
        b = self._last
        hdr, rows = unpack_wire(self.recv[b], self.loop.tcap)
        if self.bucket > 1:                    # [world, bucket, S] -> [world * S, bucket]
            S, k, w = self.loop.S, self.bucket, self.world
            hdr = np.ascontiguousarray(hdr.reshape(w, k, S).transpose(0, 2, 1)).reshape(w * S, k)
            rows = np.ascontiguousarray(rows.reshape(w, k, S, -1).transpose(0, 2, 1, 3)).reshape(w * S, k, -1)
        return hdr, rows

    def synchronize(self):
        if self.gpu:
            self.comm.synchronize()

    def close(self):
        """Destroys the native communicator (if any); the object must not exchange afterwards."""
        if self.nccl is not None:
            self.synchronize()
            nat.check(nat.lib().av_comm_destroy(self.nccl))
            self.nccl = None
