"""Small host<->device helpers for the per-frame (S=1, W=1) class surfaces."""
import numpy as np
import torch

from . import _native as nat


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("no HIP device visible: the avhot classes run on MI355X only (there is no CPU fallback)")


class Dev:
    """Device + context + stream shared by the objects of one process."""

    def __init__(self, device=0):
        require_gpu()
        self.index = device
        self.device = torch.device("cuda", device)
        self.ctx = nat.default_context(device)
        self.lib = nat.lib()

    @property
    def stream(self):
        return nat.stream_handle(torch.cuda.current_stream(self.device))

    def empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def zeros(self, shape, dtype):
        return torch.zeros(shape, dtype=dtype, device=self.device)

    def upload(self, arr, dtype):
        return torch.as_tensor(np.ascontiguousarray(arr, dtype)).to(self.device)

    def sync(self):
        torch.cuda.current_stream(self.device).synchronize()


class Packed:
    """One device buffer and one pinned host mirror carved into the same named, typed fields.

    The per-frame class surfaces (one frame per call, results read on the host after every call) pay for every
    separate copy and allocation; with this, a call is: write the inputs into the host views, ONE upload, the kernel,
    ONE download (+ stream sync), read the host views.  fields: [(name, numpy dtype, shape)], each 64-byte aligned.
    `h[name]` is a NumPy view of the pinned host copy, `ptr(name)` the device address of the field."""

    def __init__(self, dev, fields, mapped=True):
        """mapped=True: no device copy at all -- kernels read the inputs from and write the results to the pinned host
        buffer itself (hipHostMalloc memory is mapped into the device's address space; a few hundred bytes to a few
        KB per call cross PCIe inside the kernel instead of as two extra copy commands, each ~8 us of host time);
        upload() is then a no-op and download() only waits for the stream."""
        import ctypes as C
        self._dev, self._lib, self.mapped = dev, dev.lib, mapped
        off, self._off, self._shape, self._dtype = 0, {}, {}, {}
        for name, dt, shape in fields:
            dt = np.dtype(dt)
            n = int(np.prod(shape)) * dt.itemsize
            self._off[name], self._shape[name], self._dtype[name] = off, tuple(shape), dt
            off += (n + 63) & ~63
        self.nbytes = max(off, 64)
        hp = C.c_void_p()
        nat.check(self._lib.av_host_alloc(C.byref(hp), self.nbytes))
        self._hp = hp
        if mapped:
            self.dev, self._base = None, hp.value
        else:
            self.dev = torch.zeros(self.nbytes, dtype=torch.uint8, device=dev.device)
            self._base = self.dev.data_ptr()
        raw = np.ctypeslib.as_array((C.c_uint8 * self.nbytes).from_address(hp.value))
        raw[:] = 0
        self._raw = raw
        self.h = {}
        for name in self._off:
            o, dt, sh = self._off[name], self._dtype[name], self._shape[name]
            self.h[name] = raw[o:o + int(np.prod(sh)) * dt.itemsize].view(dt).reshape(sh)

    def ptr(self, name):
        import ctypes as C
        return C.c_void_p(self._base + self._off[name])

    def tensor(self, name, torch_dtype):
        """Device tensor view of a field (copy mode only)."""
        if self.mapped:
            raise RuntimeError("a mapped Packed buffer has no device tensor")
        o, sh = self._off[name], self._shape[name]
        n = int(np.prod(sh)) * self._dtype[name].itemsize
        return self.dev[o:o + n].view(torch_dtype).view(*sh)

    def upload(self, upto=None):
        """Host -> device, the whole buffer or the leading fields up to and including `upto`."""
        if self.mapped:
            return
        n = self.nbytes
        if upto is not None:
            n = self._off[upto] + ((int(np.prod(self._shape[upto])) * self._dtype[upto].itemsize + 63) & ~63)
        nat.check(self._lib.av_copy_h2d(self._base, self._hp, n, self._dev.stream))

    def upload_from(self, name, src, chunks=4):
        """Copy `src` (an array of the field's shape) into the pinned field and upload it, in `chunks` pieces along the
        first non-unit axis: the DMA of piece i runs while the CPU copies piece i+1 into the pinned buffer (a 2.8-MB frame:
        host copy 65 us + DMA 73 us one after the other, ~85 us pipelined)."""
        dst = self.h[name]
        if self.mapped or chunks <= 1 or dst.size == 0:
            np.copyto(dst, src)
            self.upload()
            return
        d2 = dst.reshape(-1, dst.shape[-1]) if dst.ndim > 1 else dst.reshape(-1, 1)
        s2 = np.asarray(src).reshape(d2.shape)
        rows, rowb = d2.shape[0], d2.shape[1] * dst.itemsize
        step = (rows + chunks - 1) // chunks
        base_off = self._off[name]
        for r0 in range(0, rows, step):
            r1 = min(rows, r0 + step)
            np.copyto(d2[r0:r1], s2[r0:r1])
            off = base_off + r0 * rowb
            nat.check(self._lib.av_copy_h2d(self._base + off, self._hp.value + off, (r1 - r0) * rowb, self._dev.stream))

    def download(self, first=None, sync=True):
        """Device -> host (from field `first` to the end), then wait for the stream."""
        if self.mapped:
            if sync:
                nat.check(self._lib.av_stream_sync_spin(self._dev.stream))
            return
        o = self._off[first] if first is not None else 0
        nat.check(self._lib.av_copy_d2h(self._hp.value + o, self._base + o, self.nbytes - o, self._dev.stream, 1 if sync else 0))

    def close(self):
        if getattr(self, "_hp", None) is not None and self._hp:
            self.h, self._raw = {}, None
            self._lib.av_host_free(self._hp)
            self._hp = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
