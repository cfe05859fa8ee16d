"""SyntheticDataGenerator: device-side stand-in for the reference's lost data/generators/synthetic_data.py.

Only the reference class's API survives (the source is gone, a 3.12 .pyc remains):
`SyntheticDataGenerator(width, height, fps)` with `generate_frame_with_vehicles`, `generate_video_stream`,
`generate_ego_motion`, `generate_agent_trajectories`, `reset`.  Frames are produced in HBM by
`av_synth_frames` (csrc/synth.hip) from integer formulas, so a frame is a pure function of
(height, width, stream, frame index) and feeds the detector / lane kernels without crossing PCIe.
The vehicle boxes below restate the kernel's rectangle formulas on the host (metadata only).
"""
import ctypes as C

import numpy as np

from . import _native as nat
from ._dev import Dev
from .harness import generate_ego_motion as _ego_motion


def vehicle_boxes(height, width, stream, frame):
    """-> list of (x1, y1, x2, y2, (b, g, r)) in paint order (later boxes cover earlier ones); csrc/synth.hip."""
    hz = (height * 9) // 20
    out = []
    k = 3 + ((stream * 5 + frame // 8) % 4)
    for i in range(k):
        s = (stream * 131 + i * 977 + (frame // 8) * 31) & 0xFFFF
        bw, bh = 50 + (s % 90), 36 + ((s >> 3) % 60)
        bx = (s * 37 + i * 211 + frame * (3 + i)) % max(width - bw, 1)
        by = hz + 10 + ((s >> 5) % max(height - hz - bh - 10, 1))
        col = (40 + (s % 160), 40 + ((s >> 4) % 160), 40 + ((s >> 8) % 160))
        out.append((bx, by, min(bx + bw, width), min(by + bh, height), col))
    return out


class SyntheticDataGenerator:
    """One camera stream per `stream` id; `n_streams` consecutive ids are rendered per call on the device."""

    def __init__(self, width=1280, height=720, fps=30.0, stream=0, n_streams=1, device=0):
        import torch
        self.width, self.height, self.fps = int(width), int(height), float(fps)
        self.stream, self.n_streams = int(stream), int(n_streams)
        self._dev = Dev(device)
        self._frames = torch.empty(self.n_streams, self.height, self.width, 3, dtype=torch.uint8,
                                   device=self._dev.device)
        self.frame_idx = 0

    # -- device side ------------------------------------------------------------------------------------
    def generate_device_frames(self):
        """Render the next frame of every stream into HBM; returns the [n_streams, H, W, 3] uint8 BGR tensor
        (re-used by the next call) without synchronising."""
        nat.check(self._dev.lib.av_synth_frames(self._dev.ctx.handle, self._dev.stream, self.n_streams, self.height,
                                                self.width, self.stream, self.frame_idx, nat.ptr(self._frames)))
        self.frame_idx += 1
        return self._frames

    # -- the reference generator's API ----------------------------------------------------------------------
    def generate_frame_with_vehicles(self):
        """-> (frame uint8[H, W, 3] BGR on the host, vehicles) for stream `self.stream`; vehicles is a list of
        dicts with the painted box and colour."""
        f = self.frame_idx
        frames = self.generate_device_frames()
        self._dev.sync()
        veh = [dict(bbox=(x1, y1, x2, y2), color=col) for x1, y1, x2, y2, col in
               vehicle_boxes(self.height, self.width, self.stream, f)]
        return frames[0].cpu().numpy(), veh

    def generate_video_stream(self, num_frames):
        for _ in range(int(num_frames)):
            yield self.generate_frame_with_vehicles()[0]

    def generate_ego_motion(self, num_steps):
        """[x, y, vx, vy] measurements, VideoDataLoader.generate_ego_motion's formula (video_loader.py:166-205)
        seeded with the stream id."""
        return _ego_motion(int(num_steps), fps=self.fps, seed=self.stream)

    def generate_agent_trajectories(self, num_agents, num_steps=100):
        """Centre tracks of the first `num_agents` painted vehicles over the next `num_steps` frames:
        float64[num_agents, num_steps, 2] (NaN where a vehicle slot is not painted in that frame)."""
        out = np.full((int(num_agents), int(num_steps), 2), np.nan)
        for t in range(int(num_steps)):
            for i, (x1, y1, x2, y2, _) in enumerate(vehicle_boxes(self.height, self.width, self.stream,
                                                                 self.frame_idx + t)[:int(num_agents)]):
                out[i, t] = ((x1 + x2) / 2.0, (y1 + y2) / 2.0)
        return out

    def reset(self):
        self.frame_idx = 0
