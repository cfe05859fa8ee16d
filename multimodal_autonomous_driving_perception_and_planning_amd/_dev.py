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
