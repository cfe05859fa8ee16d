"""MotionPlanner -- drop-in surface of src/planning/motion_planner.py over libavhot.so.

plan() is one fused launch (av_planner_plan): all 3*num_samples candidates are generated, costed and
stably ranked on the GPU; the host only wraps the result in Trajectory/Waypoint objects.
"""
import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np
import torch

from .. import _native as nat
from .._dev import Dev, Packed


@dataclass
class Waypoint:
    """One trajectory sample (motion_planner.py:14-22)."""
    x: float
    y: float
    heading: float
    velocity: float
    timestamp: float
    curvature: float = 0.0


class Trajectory:
    """A planned trajectory (motion_planner.py:25-54): same constructor, fields, value equality and helpers as
    the reference's dataclass.  Trajectories coming out of the planner carry their waypoints as one float64
    [n, 6] array and build the `List[Waypoint]` on first access to `.waypoints` (21 x 51 Python objects per
    plan() were most of its per-call cost); `get_positions`, `length`, `duration` read the array directly."""
    __slots__ = ("_wps", "_arr", "cost", "is_feasible", "trajectory_type")

    def __init__(self, waypoints: List[Waypoint], cost: float = 0.0, is_feasible: bool = True,
                 trajectory_type: str = "nominal"):
        self._wps, self._arr = waypoints, None
        self.cost, self.is_feasible, self.trajectory_type = cost, is_feasible, trajectory_type

    @classmethod
    def _from_array(cls, arr, cost=0.0, trajectory_type="nominal"):
        t = cls(None, cost=cost, trajectory_type=trajectory_type)
        t._arr = arr
        return t

    @property
    def waypoints(self) -> List[Waypoint]:
        if self._wps is None:
            self._wps = [Waypoint(*row) for row in self._arr.tolist()]
        return self._wps

    @waypoints.setter
    def waypoints(self, value):
        self._wps, self._arr = value, None

    def __eq__(self, other):
        if other.__class__ is not self.__class__:
            return NotImplemented
        return (self.waypoints, self.cost, self.is_feasible, self.trajectory_type) == \
               (other.waypoints, other.cost, other.is_feasible, other.trajectory_type)

    __hash__ = None          # like an eq dataclass

    def __repr__(self):
        return "Trajectory(waypoints=%r, cost=%r, is_feasible=%r, trajectory_type=%r)" % (
            self.waypoints, self.cost, self.is_feasible, self.trajectory_type)

    @property
    def length(self) -> float:
        if self._wps is None:
            if len(self._arr) < 2:
                return 0.0
            d = np.diff(self._arr[:, :2], axis=0)
            total = 0.0
            for v in np.sqrt(d[:, 0] ** 2 + d[:, 1] ** 2):      # same left-to-right sum as the reference's loop
                total += v
            return total
        if len(self.waypoints) < 2:
            return 0.0
        total = 0.0
        for a, b in zip(self.waypoints[:-1], self.waypoints[1:]):
            total += np.sqrt((b.x - a.x) ** 2 + (b.y - a.y) ** 2)
        return total

    @property
    def duration(self) -> float:
        if self._wps is None:
            return float(self._arr[-1, 4] - self._arr[0, 4]) if len(self._arr) else 0.0
        if not self.waypoints:
            return 0.0
        return self.waypoints[-1].timestamp - self.waypoints[0].timestamp

    def get_positions(self) -> np.ndarray:
        if self._wps is None:
            return np.array(self._arr[:, :2])
        return np.array([[wp.x, wp.y] for wp in self.waypoints])


def _to_waypoints(arr):
    return [Waypoint(*row) for row in arr.tolist()]


class MotionPlanner:
    def __init__(self, planning_horizon: float = 5.0, dt: float = 0.1, num_samples: int = 7, device: int = 0):
        self.planning_horizon = planning_horizon
        self.dt = dt
        self.num_samples = num_samples
        self.w_lateral = 1.0
        self.w_velocity = 0.5
        self.w_acceleration = 0.3
        self.w_jerk = 0.2            # unused by the reference's cost as well (motion_planner.py:88)
        self.w_curvature = 0.4
        self.reference_trajectory: Optional[Trajectory] = None
        self._dev = Dev(device)
        self._sig = None
        self._ref_dev = None

    # ---- device plumbing --------------------------------------------------------------------------
    def _configure(self):
        """(Re)builds the constant tables when a public attribute changed since the last call."""
        sig = (float(self.planning_horizon), float(self.dt), int(self.num_samples), float(self.w_lateral),
               float(self.w_velocity), float(self.w_acceleration), float(self.w_curvature))
        d = self._dev
        cfg = nat.PlannerCfg(sig[0], sig[1], sig[2], 0, sig[3], sig[4], sig[5], sig[6])
        # the context's tables are shared by every planner on this device: always re-assert ours
        if getattr(d.ctx, "_planner_sig", None) != sig:
            nat.check(d.lib.av_planner_configure(d.ctx.handle, C.byref(cfg)))
            d.ctx._planner_sig = sig
        if self._sig != sig:
            n, c = C.c_int(), C.c_int()
            nat.check(d.lib.av_planner_dims(d.ctx.handle, C.byref(n), C.byref(c)))
            self._n, self._c = n.value, c.value
            self._sig = sig
            # start state up, every candidate's waypoints + costs + ranking down: one copy each way per plan()
            self._io = Packed(d, [("st", np.float64, (1, 4)), ("wp", np.float64, (1, self._c, self._n, nat.WP_DOUBLES)),
                                  ("cost", np.float64, (1, self._c)), ("order", np.int32, (1, self._c))])
            self._lat = np.linspace(-3.5, 3.5, int(self.num_samples))
            self._kinds = ["lane_keep" if abs(self._lat[c // 3]) < 0.5 else
                           ("lane_change_left" if self._lat[c // 3] < 0 else "lane_change_right") for c in range(self._c)]

    def _ref_arrays(self):
        if not self.reference_trajectory:
            return None, 0
        pts = self.reference_trajectory.get_positions()
        return self._dev.upload(pts, np.float64), len(pts)

    @staticmethod
    def _obs_arrays(dev, obstacles):
        if not obstacles:
            return None, 0
        o = np.asarray([tuple(x) for x in obstacles], np.float64).reshape(-1, 3)
        return dev.upload(o, np.float64), len(o)

    # ---- reference surface ----------------------------------------------------------------------------
    def set_reference_path(self, waypoints: List[Tuple[float, float]]):
        if len(waypoints) < 2:
            return
        pts = [(float(x), float(y)) for x, y in waypoints]
        wps = []
        for i, (x, y) in enumerate(pts):
            if i < len(pts) - 1:
                hd = np.arctan2(pts[i + 1][1] - y, pts[i + 1][0] - x)
            else:
                hd = np.arctan2(y - pts[i - 1][1], x - pts[i - 1][0])
            wps.append(Waypoint(x=x, y=y, heading=hd, velocity=10.0, timestamp=i * 0.5))
        self.reference_trajectory = Trajectory(waypoints=wps, trajectory_type="reference")

    def generate_polynomial_trajectory(self, start_state: Tuple[float, float, float, float],
                                       end_lateral_offset: float, target_velocity: float) -> Trajectory:
        self._configure()
        d = self._dev
        st = d.upload(np.asarray(start_state, np.float64).reshape(1, 4), np.float64)
        df = d.upload([float(end_lateral_offset)], np.float64)
        vt = d.upload([float(target_velocity)], np.float64)
        wp = d.empty((1, self._n, nat.WP_DOUBLES), torch.float64)
        nat.check(d.lib.av_planner_generate(d.ctx.handle, d.stream, 1, nat.ptr(st), nat.ptr(df), nat.ptr(vt), nat.ptr(wp)))
        return Trajectory._from_array(wp.cpu().numpy()[0])

    def evaluate_trajectory_cost(self, trajectory: Trajectory,
                                 obstacles: Optional[List[Tuple[float, float, float]]] = None) -> float:
        if (trajectory._wps is None and len(trajectory._arr) == 0) or (trajectory._wps is not None and not trajectory._wps):
            return float("inf")
        self._configure()
        d = self._dev
        if trajectory._wps is None:
            arr = np.ascontiguousarray(trajectory._arr, np.float64)
        else:
            arr = np.array([[w.x, w.y, w.heading, w.velocity, w.timestamp, w.curvature] for w in trajectory.waypoints],
                           np.float64)
        wp = d.upload(arr.reshape(1, -1, 6), np.float64)
        ref, nr = self._ref_arrays()
        obs, no = self._obs_arrays(d, obstacles)
        out = d.empty(1, torch.float64)
        nat.check(d.lib.av_planner_evaluate(d.ctx.handle, d.stream, 1, len(arr), nat.ptr(wp), nat.ptr(ref), nr,
                                            nat.ptr(obs), no, nat.ptr(out)))
        trajectory.cost = float(out.item())
        return trajectory.cost

    def plan(self, current_state: Tuple[float, float, float, float],
             obstacles: Optional[List[Tuple[float, float, float]]] = None) -> Tuple[Trajectory, List[Trajectory]]:
        self._configure()
        d = self._dev
        io = self._io
        io.h["st"][0] = np.asarray(current_state, np.float64).reshape(4)
        io.upload(upto="st")
        ref, nr = self._ref_arrays()
        obs, no = self._obs_arrays(d, obstacles)
        nat.check(d.lib.av_planner_plan(d.ctx.handle, d.stream, 1, io.ptr("st"), nat.ptr(ref), nr, nat.ptr(obs), no,
                                        io.ptr("wp"), io.ptr("cost"), io.ptr("order")))
        io.download(first="wp")
        wph = io.h["wp"][0].copy()                 # the staging buffer is reused by the next call
        costh, orderh = io.h["cost"][0].tolist(), io.h["order"][0].tolist()
        kinds = self._kinds
        gen = [Trajectory._from_array(wph[c], cost=costh[c], trajectory_type=kinds[c]) for c in range(self._c)]
        candidates = [gen[c] for c in orderh]
        return (candidates[0] if candidates else None), candidates

    def draw_trajectories(self, frame: np.ndarray, optimal: Optional[Trajectory], candidates: List[Trajectory],
                          transform_func=None, draw_all: bool = True) -> np.ndarray:
        """Candidate and optimal paths over a camera frame (motion_planner.py:305-370), drawn by the device rasteriser."""
        from ..visualization._prims import PrimList, paint
        if transform_func is None:
            h, w = frame.shape[:2]

            def transform_func(x, y):
                return int(w / 2 + x * 10), int(h - y * 10 - 50)
        pl = PrimList()
        if draw_all:
            worst = max((t.cost for t in candidates), default=0.0) + 1
            for t in candidates:
                if t == optimal or len(t.waypoints) < 2:
                    continue
                ratio = t.cost / worst
                pl.polylines([transform_func(x, y) for x, y in t.get_positions()], False, (0, int(255 * (1 - ratio)), int(255 * ratio)), 1)
        if optimal and len(optimal.waypoints) >= 2:
            pos = optimal.get_positions()
            pts = [transform_func(x, y) for x, y in pos]
            pl.polylines(pts, False, (0, 255, 0), 4)
            pl.polylines(pts, False, (100, 255, 100), 2)
            for x, y in pos[::5]:
                pl.circle(transform_func(x, y), 3, (255, 255, 0), -1)
        return paint(frame, pl, self._dev.index)

    def reset(self):
        self.reference_trajectory = None
