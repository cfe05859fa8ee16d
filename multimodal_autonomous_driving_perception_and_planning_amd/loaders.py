"""VideoDataLoader -- drop-in surface of data/loaders/video_loader.py for UNCOMPRESSED video.

The reference opens any container OpenCV can decode (cv2.VideoCapture, video_loader.py:43-57) and seeks with
CAP_PROP_POS_FRAMES per frame (:120).  Neither OpenCV nor any other decoder exists in this image, so this loader reads the
two formats that need none and does the pixel work on the device:
  * `.y4m`  YUV4MPEG2, 4:2:0 planar (what `ffmpeg -pix_fmt yuv420p out.y4m` writes): frames are memory-mapped, a seek is
            an offset, I420 -> BGR (av_i420_to_bgr) and the resize to target_size (av_resize_into) run on the GPU;
  * `.npy`  uint8 array [frames, height, width, 3] (BGR), memory-mapped; resize on the GPU.
Compressed containers (mp4 / avi / mov / mkv) raise ValueError naming the missing decoder.  Same properties and methods
as the reference class, plus read_frames_device() which leaves a batch of frames in HBM for the batched pipeline.
"""
from pathlib import Path
from typing import Generator, Optional, Tuple

import numpy as np
import torch

from . import _native as nat
from ._dev import Dev


class VideoDataLoader:
    def __init__(self, video_path: str, target_size: Optional[Tuple[int, int]] = None, device: int = 0):
        self.video_path = Path(video_path)
        self.target_size = target_size
        self.cap = None
        self.frame_count = 0
        if not self.video_path.exists():
            raise FileNotFoundError("Video file not found: %s" % video_path)
        self._dev = Dev(device)
        self._open_video()

    def _open_video(self):
        suf = self.video_path.suffix.lower()
        if suf == ".npy":
            arr = np.load(str(self.video_path), mmap_mode="r")
            if arr.ndim != 4 or arr.shape[3] != 3 or arr.dtype != np.uint8:
                raise ValueError("Could not open video file: %s (expected uint8 [frames, h, w, 3])" % self.video_path)
            self._kind, self._arr, self._fps = "npy", arr, 30.0
            self._total_frames, self._height, self._width = int(arr.shape[0]), int(arr.shape[1]), int(arr.shape[2])
        elif suf == ".y4m":
            with open(self.video_path, "rb") as f:
                head = f.readline(256)
            if not head.startswith(b"YUV4MPEG2 "):
                raise ValueError("Could not open video file: %s (not a YUV4MPEG2 stream)" % self.video_path)
            w = h = 0
            fps, chroma = 30.0, "420"
            for tok in head.split()[1:]:
                t = tok.decode("ascii", "replace")
                if t[0] == "W":
                    w = int(t[1:])
                elif t[0] == "H":
                    h = int(t[1:])
                elif t[0] == "F" and ":" in t:
                    a, b = t[1:].split(":")
                    fps = float(a) / float(b) if float(b) else 30.0
                elif t[0] == "C":
                    chroma = t[1:]
            if not chroma.startswith("420") or w <= 0 or h <= 0 or w % 2 or h % 2:
                raise ValueError("Could not open video file: %s (only even-sized 4:2:0 Y4M is supported, got C%s %dx%d)" % (
                    self.video_path, chroma, w, h))
            self._kind, self._fps, self._width, self._height = "y4m", fps, w, h
            self._hdr, self._fsz = len(head), w * h * 3 // 2
            raw = np.memmap(str(self.video_path), np.uint8, mode="r")
            self._raw = raw
            self._total_frames = (len(raw) - self._hdr) // (6 + self._fsz)        # every frame: b"FRAME\n" + planes
            if self._total_frames <= 0 or bytes(raw[self._hdr:self._hdr + 6]) != b"FRAME\n":
                raise ValueError("Could not open video file: %s (no plain FRAME records)" % self.video_path)
        else:
            raise ValueError("Could not open video file: %s -- %s needs a bitstream decoder (OpenCV / FFmpeg / VCN), none is "
                             "available here; convert to .y4m (4:2:0) or .npy" % (self.video_path, suf or "this format"))
        self.cap = self
        self._duration = self._total_frames / self._fps if self._fps > 0 else 0

    # ---- properties (video_loader.py:59-87) ------------------------------------------------------------------------
    total_frames = property(lambda self: self._total_frames)
    fps = property(lambda self: self._fps)
    width = property(lambda self: self.target_size[0] if self.target_size else self._width)
    height = property(lambda self: self.target_size[1] if self.target_size else self._height)
    duration = property(lambda self: self._duration)
    dt = property(lambda self: 1.0 / self._fps if self._fps > 0 else 0.033)

    # ---- frames ---------------------------------------------------------------------------------------------------------
    def read_frames_device(self, first: int, count: int) -> Optional[torch.Tensor]:
        """Frames [first, first + count) as a uint8 tensor [count, height, width, 3] (BGR) in HBM: planes / pixels go up
        as they lie in the file, colour conversion and resize happen on the device."""
        if self.cap is None or first < 0 or count <= 0 or first + count > self._total_frames:
            return None
        d, h, w = self._dev, self._height, self._width
        if self._kind == "npy":
            src = d.upload(np.array(self._arr[first:first + count]), np.uint8)            # copy out of the read-only map
        else:
            stride = 6 + self._fsz
            planes = np.stack([self._raw[self._hdr + k * stride + 6:self._hdr + (k + 1) * stride] for k in range(first, first + count)])
            yuv = d.upload(planes, np.uint8)
            src = d.empty((count, h, w, 3), torch.uint8)
            nat.check(d.lib.av_i420_to_bgr(d.ctx.handle, d.stream, count, h, w, nat.ptr(yuv), nat.ptr(src)))
        if self.target_size is None or tuple(self.target_size) == (w, h):
            return src
        tw, th = int(self.target_size[0]), int(self.target_size[1])
        out = d.empty((count, th, tw, 3), torch.uint8)
        for k in range(count):
            nat.check(d.lib.av_resize_into(d.ctx.handle, d.stream, nat.ptr(src[k]), h, w, nat.ptr(out[k]), th, tw, tw, 0))
        return out

    def read_frame(self) -> Optional[np.ndarray]:
        t = self.read_frames_device(self.frame_count, 1)
        if t is None:
            return None
        self.frame_count += 1
        return t[0].cpu().numpy()

    def read_frame_at(self, frame_idx: int) -> Optional[np.ndarray]:
        t = self.read_frames_device(frame_idx, 1)
        if t is None:
            return None
        self.frame_count = frame_idx + 1
        return t[0].cpu().numpy()

    def generate_frame_with_vehicles(self) -> Optional[np.ndarray]:
        return self.read_frame()

    def generate_video_stream(self, num_frames: Optional[int] = None) -> Generator[np.ndarray, None, None]:
        self.reset()
        n = num_frames if num_frames else self._total_frames
        for _ in range(n):
            frame = self.read_frame()
            if frame is None:
                break
            yield frame

    def generate_ego_motion(self, num_steps: Optional[int] = None) -> list:
        """Placeholder ego measurements (video_loader.py:166-205), drawn from the global NumPy stream like the reference."""
        if num_steps is None:
            num_steps = self._total_frames
        out, x, y, speed, dt = [], 0.0, 0.0, 10.0, self.dt
        for i in range(num_steps):
            heading = 0.05 * np.sin(i * dt * 0.5)
            vx, vy = speed * np.cos(heading), speed * np.sin(heading)
            x += vx * dt
            y += vy * dt
            out.append((x + np.random.normal(0, 0.1), y + np.random.normal(0, 0.1), vx + np.random.normal(0, 0.05),
                        vy + np.random.normal(0, 0.05)))
        return out

    def reset(self):
        self.frame_count = 0

    def release(self):
        self.cap = None

    def __len__(self) -> int:
        return self._total_frames

    def __iter__(self):
        self.reset()
        return self

    def __next__(self) -> np.ndarray:
        frame = self.read_frame()
        if frame is None:
            raise StopIteration
        return frame

    def get_info(self) -> dict:
        return {"path": str(self.video_path), "total_frames": self._total_frames, "fps": self._fps, "width": self._width,
                "height": self._height, "duration": self._duration, "target_size": self.target_size}

    def __repr__(self) -> str:
        return "VideoDataLoader(path='%s', frames=%d, fps=%.1f, size=%dx%d)" % (self.video_path.name, self._total_frames, self._fps,
                                                                             self._width, self._height)
