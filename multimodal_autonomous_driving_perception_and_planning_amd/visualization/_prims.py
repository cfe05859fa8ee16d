"""Host side of the device rasteriser (csrc/raster.hip): an ordered primitive list per picture and the call that paints it.

The reference draws with cv2 calls; each of them maps onto one or a few av_prim records (include/avhot.h) in the same
order, so a picture built here has the reference's layout and layering.  Pixel rules are this project's own (parity
unpinned: OpenCV is absent), see raster.hip.
"""
import ctypes as C
import math

import numpy as np
import torch

from .. import _native as nat
from .._dev import Dev

_PRIM = np.dtype(nat.PRIM_FIELDS)
assert _PRIM.itemsize == nat.PRIM_BYTES


def cv_round(v):
    """cvRound: nearest integer, halves to even."""
    return int(np.rint(v))


class PrimList:
    """Ordered list of drawing primitives; method names follow the cv2 call each one stands for.  Colours are BGR."""

    def __init__(self):
        self.rows = []
        self.verts = []

    def _add(self, t, x0=0, y0=0, x1=0, y1=0, x2=0, y2=0, x3=0, y3=0, p=0, color=(0, 0, 0)):
        self.rows.append((t, int(x0), int(y0), int(x1), int(y1), int(x2), int(y2), int(x3), int(y3), int(p),
                          int(color[0]) & 255, int(color[1]) & 255, int(color[2]) & 255, 0, 0))

    def rectangle(self, pt1, pt2, color, thickness=1):
        """cv2.rectangle: filled for thickness < 0, else the outline."""
        (x0, y0), (x1, y1) = pt1, pt2
        if thickness < 0:
            self._add(nat.PRIM_RECT, x0, y0, x1, y1, color=color)
        else:
            for a, b in (((x0, y0), (x1, y0)), ((x1, y0), (x1, y1)), ((x1, y1), (x0, y1)), ((x0, y1), (x0, y0))):
                self.line(a, b, color, thickness)

    def line(self, pt1, pt2, color, thickness=1):
        self._add(nat.PRIM_SEG, pt1[0], pt1[1], pt2[0], pt2[1], p=max(1, int(thickness)), color=color)

    def arrowed_line(self, pt1, pt2, color, thickness=1, tip_length=0.1):
        """cv2.arrowedLine: the shaft plus two tip strokes at +-45 degrees, tip size = length * tip_length."""
        self.line(pt1, pt2, color, thickness)
        tip = math.hypot(pt1[0] - pt2[0], pt1[1] - pt2[1]) * tip_length
        ang = math.atan2(pt1[1] - pt2[1], pt1[0] - pt2[0])
        for da in (math.pi / 4, -math.pi / 4):
            p = (cv_round(pt2[0] + tip * math.cos(ang + da)), cv_round(pt2[1] + tip * math.sin(ang + da)))
            self.line(p, pt2, color, thickness)

    def polylines(self, pts, closed, color, thickness=1):
        pts = [tuple(int(v) for v in p) for p in np.asarray(pts).reshape(-1, 2)]
        for a, b in zip(pts[:-1], pts[1:]):
            self.line(a, b, color, thickness)
        if closed and len(pts) > 2:
            self.line(pts[-1], pts[0], color, thickness)

    def fill_convex_quad(self, pts, color):
        """cv2.fillPoly of a convex quadrilateral (vehicle footprints)."""
        (x0, y0), (x1, y1), (x2, y2), (x3, y3) = [tuple(int(v) for v in p) for p in np.asarray(pts).reshape(4, 2)]
        self._add(nat.PRIM_QUAD, x0, y0, x1, y1, x2, y2, x3, y3, color=color)

    def circle(self, center, radius, color, thickness=1):
        self._add(nat.PRIM_DISC if thickness < 0 else nat.PRIM_RING, center[0], center[1], p=int(radius), color=color)

    def put_text(self, text, org, font_scale, color, thickness=1):
        """cv2.putText(FONT_HERSHEY_SIMPLEX): org is the bottom-left of the text; drawn in the 5x7 bitmap font, doubled
        for the larger labels (Hershey at scale 0.3-0.5 has 7-11 pixel capitals, at 0.6 thirteen)."""
        sc = 2 if font_scale >= 0.55 else 1
        x, y = int(org[0]), int(org[1]) - 7 * sc
        for ch in str(text):
            code = ord(ch)
            if 32 < code <= 126:
                self._add(nat.PRIM_GLYPH, x, y, sc, p=code, color=color)
            x += 6 * sc

    def blend_rectangle(self, pt1, pt2, color):
        """overlay = frame.copy(); cv2.rectangle(overlay, ..., -1); frame = cv2.addWeighted(frame, 0.7, overlay, 0.3, 0)."""
        self._add(nat.PRIM_BLEND_RECT, pt1[0], pt1[1], pt2[0], pt2[1], color=color)

    def blend_polygon(self, pts, color):
        """The same with cv2.fillPoly of an arbitrary polygon (the lane area)."""
        pts = np.asarray(pts, np.int64).reshape(-1, 2)
        v0 = len(self.verts)
        self.verts += [(int(x), int(y)) for x, y in pts]
        self._add(nat.PRIM_POLY_BLEND, v0, len(pts), 0, 0, int(pts[:, 0].min()), int(pts[:, 1].min()), int(pts[:, 0].max()),
                  int(pts[:, 1].max()), color=color)

    def array(self):
        return np.array(self.rows, _PRIM) if self.rows else np.zeros(0, _PRIM)

    def vert_array(self):
        return np.asarray(self.verts, np.int32).reshape(-1, 2)


class Raster:
    """Device buffers for one picture size; paint(img, prims) -> painted copy (uint8 [H, W, 3])."""

    def __init__(self, h, w, device=0, prim_cap=4096, vert_cap=512):
        self._dev = Dev(device)
        self.h, self.w, self.prim_cap, self.vert_cap = h, w, prim_cap, vert_cap
        d = self._dev
        self.img = d.empty((1, h, w, 3), torch.uint8)
        self.prims = d.zeros((1, prim_cap, nat.PRIM_BYTES), torch.uint8)
        self.n = d.zeros(1, torch.int32)
        self.verts = d.zeros((1, vert_cap, 2), torch.int32)

    def _grow(self, np_, nv):
        """Capacities grow in powers of two up to the kernel's limit of 65535 primitives; nothing is changed when the request
        cannot be met (the cached Raster stays usable)."""
        d = self._dev
        if np_ > 65535:
            raise ValueError("too many primitives in one picture (%d > 65535)" % np_)
        if np_ > self.prim_cap:
            cap = self.prim_cap
            while cap < np_:
                cap *= 2
            cap = min(cap, 65535)
            self.prims = d.zeros((1, cap, nat.PRIM_BYTES), torch.uint8)
            self.prim_cap = cap
        if nv > self.vert_cap:
            cap = self.vert_cap
            while cap < nv:
                cap *= 2
            self.verts = d.zeros((1, cap, 2), torch.int32)
            self.vert_cap = cap

    def upload(self, img):
        self.img.copy_(torch.as_tensor(np.ascontiguousarray(img, np.uint8)).view(1, self.h, self.w, 3))

    def draw(self, plist):
        """Paints the list onto the picture currently in self.img (device), in place."""
        d = self._dev
        arr, va = plist.array(), plist.vert_array()
        if len(arr) == 0:
            return
        self._grow(len(arr), len(va))
        self.prims[0, :len(arr)].copy_(torch.as_tensor(arr.view(np.uint8).reshape(len(arr), nat.PRIM_BYTES)))
        if len(va):
            self.verts[0, :len(va)].copy_(torch.as_tensor(va))
        self.n.fill_(len(arr))
        nat.check(d.lib.av_raster_draw(d.ctx.handle, d.stream, 1, self.h, self.w, nat.ptr(self.img), nat.ptr(self.prims),
                                       self.prim_cap, nat.ptr(self.n), nat.ptr(self.verts), self.vert_cap))

    def download(self):
        return self.img.cpu().numpy()[0]

    def paint(self, img, plist):
        self.upload(img)
        self.draw(plist)
        return self.download()


_rasters = {}


def raster_for(h, w, device=0):
    r = _rasters.get((h, w, device))
    if r is None:
        r = _rasters[(h, w, device)] = Raster(h, w, device)
    return r


def paint(img, plist, device=0):
    """Paint a primitive list onto a host image through the device rasteriser; returns the painted copy."""
    img = np.ascontiguousarray(img, np.uint8)
    if img.ndim != 3 or img.shape[2] != 3:
        raise ValueError("expected an HxWx3 uint8 image, got shape %s" % (img.shape,))
    return raster_for(img.shape[0], img.shape[1], device).paint(img, plist)
