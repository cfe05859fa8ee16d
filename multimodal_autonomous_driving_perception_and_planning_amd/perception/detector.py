"""ObjectDetector -- drop-in surface of src/perception/detector.py over libavhot.so.

mode="simulated": the detections of frame k are produced on the GPU by av_simdet_generate, which
re-derives NumPy's legacy MT19937 stream the reference seeds with frame_count % 1000
(detector.py:125-169).  Unlike the reference, the process-global np.random state is left alone.
mode="yolo": a YOLOv8n-topology network run by the library's MFMA convolution path (see yolo.py);
like the reference (detector.py:79-84) it silently falls back to "simulated" when the model cannot
be set up.
"""
import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np
import torch

from .. import _native as nat
from .._dev import Dev, Packed


@dataclass
class Detection:
    """One detected object (detector.py:14-26)."""
    bbox: Tuple[int, int, int, int]
    class_id: int
    class_name: str
    confidence: float
    center: Tuple[float, float] = None

    def __post_init__(self):
        if self.center is None:
            x1, y1, x2, y2 = self.bbox
            self.center = ((x1 + x2) / 2, (y1 + y2) / 2)


class ObjectDetector:
    CLASSES = {0: "car", 1: "truck", 2: "pedestrian", 3: "cyclist", 4: "motorcycle", 5: "bus",
               6: "traffic_light", 7: "stop_sign"}
    CLASS_COLORS = {0: (0, 255, 0), 1: (0, 165, 255), 2: (0, 0, 255), 3: (255, 255, 0), 4: (255, 0, 255),
                    5: (0, 255, 255), 6: (128, 0, 128), 7: (0, 128, 255)}
    DCAP = 8

    def __init__(self, mode: str = "simulated", model_path: Optional[str] = None, device: int = 0):
        self.mode = mode
        self.model = None
        self.frame_count = 0
        self._dev = Dev(device)
        # one device buffer + pinned host mirror: counter (uploaded only when the host's frame_count was changed behind
        # the device's back), then the outputs, fetched with one copy per frame
        self._io = Packed(self._dev, [("fc", np.int32, (1,)), ("n", np.int32, (1, 1)), ("box", np.int32, (1, 1, self.DCAP, 4)),
                                      ("cls", np.int32, (1, 1, self.DCAP)), ("conf", np.float64, (1, 1, self.DCAP)),
                                      ("status", np.int32, (1,))])
        self._dev_fc = None                      # what the device counter holds (None: unknown)
        if mode == "yolo" and model_path:
            self._load_yolo_model(model_path)

    def _load_yolo_model(self, model_path: str):
        try:
            from .yolo import YoloV8n
            self.model = YoloV8n(model_path, device=self._dev.index)
        except (ImportError, FileNotFoundError) as e:
            print("YOLO path unavailable (%s). Falling back to simulated mode." % e)
            self.mode = "simulated"

    def detect(self, frame: np.ndarray) -> List[Detection]:
        self.frame_count += 1
        if self.mode == "yolo" and self.model is not None:
            return self._detect_yolo(frame)
        return self._detect_simulated(frame)

    def _detect_yolo(self, frame):
        boxes, conf, cls = self.model.detect(frame)
        names = self.model.names
        # int() of every coordinate (detector.py:111: truncation toward zero), float() of every confidence, the box centre of
        # Detection.__post_init__ -- for up to 300 boxes: as array operations, then plain attribute stores (a random-init network keeps
        # max_det = 300 boxes per frame; 300 dataclass constructions over NumPy scalars were 0.4 ms of a 1.1-ms call)
        bi = boxes.astype(np.int64)
        cx, cy = ((bi[:, 0] + bi[:, 2]) / 2).tolist(), ((bi[:, 1] + bi[:, 3]) / 2).tolist()
        out, new = [], object.__new__
        for b, k, c, x, y in zip(bi.tolist(), cls.tolist(), conf.astype(np.float64).tolist(), cx, cy):
            d = new(Detection)
            d.bbox, d.class_id, d.class_name, d.confidence, d.center = (b[0], b[1], b[2], b[3]), k, names.get(k, "unknown"), c, (x, y)
            out.append(d)
        return out

    def _detect_simulated(self, frame) -> List[Detection]:
        h, w = frame.shape[:2]
        d, io = self._dev, self._io
        if self._dev_fc != self.frame_count - 1:    # the kernel generates frame counter + 1 and advances the counter
            io.h["fc"][0] = self.frame_count - 1
            io.upload(upto="fc")
        nat.check(d.lib.av_simdet_generate(d.ctx.handle, d.stream, 1, 1, int(h), int(w), self.DCAP, io.ptr("fc"), io.ptr("n"),
                                           io.ptr("box"), io.ptr("cls"), io.ptr("conf"), io.ptr("status")))
        io.download(first="n")
        self._dev_fc = self.frame_count
        n = int(io.h["n"][0, 0])
        box, cls, conf = io.h["box"][0, 0, :n].tolist(), io.h["cls"][0, 0, :n].tolist(), io.h["conf"][0, 0, :n].tolist()
        names = self.CLASSES
        return [Detection(bbox=tuple(box[i]), class_id=cls[i], class_name=names[cls[i]], confidence=conf[i])
                for i in range(n)]

    def draw_detections(self, frame: np.ndarray, detections: List[Detection], show_labels: bool = True,
                        show_confidence: bool = True) -> np.ndarray:
        """Boxes + labels (detector.py:171-222), drawn by the device rasteriser (visualization/_prims.py)."""
        from ..visualization._prims import PrimList, paint
        pl = PrimList()
        for det in detections:
            x1, y1, x2, y2 = det.bbox
            color = self.CLASS_COLORS.get(det.class_id, (255, 255, 255))
            pl.rectangle((x1, y1), (x2, y2), color, 2)
            if show_labels:
                text = det.class_name + (" %.2f" % det.confidence if show_confidence else "")
                tw, th = 6 * len(text), 9                          # extent of the label in the 5x7 font
                pl.rectangle((x1, y1 - th - 10), (x1 + tw + 5, y1), color, -1)
                pl.put_text(text, (x1 + 2, y1 - 5), 0.5, (0, 0, 0), 1)
        return paint(frame, pl, self._dev.index)

    def reset(self):
        self.frame_count = 0
