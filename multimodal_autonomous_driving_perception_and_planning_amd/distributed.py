"""Multi-GPU: stream sharding + all-gather of per-frame track tables.

The hot path shards by video stream: every object in the loop is per-stream state and there is
no cross-stream term, so rank g of G owns streams [g*S/G, (g+1)*S/G) and no
data-path collective is needed for correctness.  The one exchange BASELINE.json asks for is an
all-gather of the track tables (a fleet-wide view for downstream consumers such as the reference's
InteractionDetector); the reference itself has no counterpart.

One process per GPU, torch.distributed backend "nccl" (= RCCL over xGMI) on the GPU box, "gloo"
in the CPU tests.  The gather runs on its own stream, double-buffered, so window k's exchange
overlaps window k+1's kernels.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import _native as nat


def shard_streams(n_streams_total, world, rank):
    """Contiguous block partition; the first (n % world) ranks take one extra stream."""
    base, extra = divmod(n_streams_total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


WIRE_ROW_BYTES = nat.TRACK_ROW_BYTES


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
    """Blocking all-gather of a packed message; returns [world*S, bytes]."""
    out = torch.empty(world * msg.shape[0], msg.shape[1], dtype=msg.dtype, device=msg.device)
    dist.all_gather_into_tensor(out, msg.contiguous(), group=group)
    return out


class TrackTableExchange:
    """Per window: gather every rank's end-of-window track tables (one per local stream)."""

    def __init__(self, loop, world, rank, group=None):
        self.loop, self.world, self.rank, self.group = loop, world, rank, group
        S, tcap = loop.S, loop.tcap
        nbytes = tcap * WIRE_ROW_BYTES + 64
        dev = loop.dev
        self.send = [torch.zeros(S, nbytes, dtype=torch.uint8, device=dev) for _ in range(2)]
        self.recv = [torch.zeros(world * S, nbytes, dtype=torch.uint8, device=dev) for _ in range(2)]
        self.comm = torch.cuda.Stream(device=dev)
        self.ready = [torch.cuda.Event() for _ in range(2)]
        self.done = [None, None]
        self.k = 0

    def exchange(self):
        b = self.k & 1
        loop = self.loop
        if self.done[b] is not None:           # buffer b is still being sent from two windows ago
            loop.stream.wait_event(self.done[b])
        tb = loop.tcap * WIRE_ROW_BYTES
        with torch.cuda.stream(loop.stream):
            self.send[b][:, :tb] = loop.snap[:, loop.W - 1].reshape(loop.S, tb)
            self.send[b][:, tb:tb + 4] = loop.snap_n[:, loop.W - 1].contiguous().view(torch.uint8).reshape(loop.S, 4)
            self.ready[b].record(loop.stream)
        with torch.cuda.stream(self.comm):
            self.comm.wait_event(self.ready[b])
            dist.all_gather_into_tensor(self.recv[b], self.send[b], group=self.group)
            ev = torch.cuda.Event()
            ev.record(self.comm)
            self.done[b] = ev
        self.k += 1
        return self.recv[b]

    def latest(self):
        """Host view (rows [world*S, tcap], counts [world*S]) of the most recent completed gather."""
        self.synchronize()
        b = (self.k - 1) & 1
        return unpack_tables(self.recv[b], self.loop.tcap)

    def synchronize(self):
        self.comm.synchronize()
