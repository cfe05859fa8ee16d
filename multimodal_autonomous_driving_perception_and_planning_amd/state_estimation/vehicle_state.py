"""VehicleStateEstimator -- drop-in surface of src/state_estimation/vehicle_state.py over libavhot.so.

The 6-state filter (x, P, prev_heading, prev_speed, time) lives in one 48-double device record
advanced by av_kf_step; `self.kf` is a thin view exposing the arrays filterpy's object exposes.
"""
import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np
import torch

from .. import _native as nat
from .._dev import Dev, Packed


@dataclass
class VehicleState:
    """Estimated ego state (vehicle_state.py:14-30)."""
    x: float
    y: float
    vx: float
    vy: float
    heading: float
    speed: float
    acceleration: float
    yaw_rate: float
    timestamp: float
    pos_uncertainty: float = 0.0
    vel_uncertainty: float = 0.0
    heading_uncertainty: float = 0.0


class _FilterView:
    """Host view of the device filter record with filterpy-like attribute names."""

    def __init__(self, owner):
        self._o = owner
        self.dim_x, self.dim_z = 6, 4

    def _rec(self):
        return self._o._state.cpu().numpy()[0]

    @property
    def x(self):
        return self._rec()[:6].copy()

    @x.setter
    def x(self, v):
        rec = self._rec()
        rec[:6] = np.asarray(v, np.float64).reshape(6)
        self._o._state.copy_(torch.as_tensor(rec).view(1, -1))

    @property
    def P(self):
        return self._rec()[6:42].reshape(6, 6).copy()

    @P.setter
    def P(self, v):
        rec = self._rec()
        rec[6:42] = np.asarray(v, np.float64).reshape(36)
        self._o._state.copy_(torch.as_tensor(rec).view(1, -1))

    @property
    def F(self):
        dt = self._o.dt
        F = np.eye(6)
        F[0, 2] = F[1, 3] = F[2, 4] = F[3, 5] = dt
        F[0, 4] = F[1, 5] = 0.5 * dt ** 2
        return F

    @property
    def H(self):
        H = np.zeros((4, 6))
        H[:4, :4] = np.eye(4)
        return H

    @property
    def R(self):
        return np.eye(4) * self._o._r

    @property
    def Q(self):
        Q = np.eye(6) * self._o._q
        Q[4, 4] *= 10
        Q[5, 5] *= 10
        return Q


class VehicleStateEstimator:
    def __init__(self, dt: float = 0.033, process_noise: float = 0.1, measurement_noise: float = 1.0,
                 device: int = 0):
        self.dt = dt
        self._q, self._r = process_noise, measurement_noise
        self._dev = Dev(device)
        d = self._dev
        self._state = d.zeros((1, nat.KF_STATE_DOUBLES), torch.float64)
        # measurement + mode go up in one copy, the derived state comes back in one
        self._io = Packed(d, [("z", np.float64, (1, 1, 4)), ("mode", np.uint8, (1, 1)), ("out", np.float64, (1, 1, nat.VSTATE_DOUBLES))])
        self.kf = _FilterView(self)
        self.state_history: List[VehicleState] = []
        self.reset()

    # ---- device plumbing --------------------------------------------------------------------------
    def _run(self, mode, z=None) -> VehicleState:
        d = self._dev
        if z is not None:
            zz = np.asarray(z, np.float64).reshape(-1)
            if zz.size != 4:
                raise ValueError("measurement must have 4 elements [x, y, vx, vy], got shape %s" % (np.shape(z),))
            self._io.h["z"][0, 0] = zz
        io = self._io
        io.h["mode"][0, 0] = mode
        io.upload(upto="mode")
        cfg = nat.KfCfg(float(self.dt), float(self._q), float(self._r))
        nat.check(d.lib.av_kf_step(d.ctx.handle, d.stream, C.byref(cfg), 1, 1, io.ptr("z"), io.ptr("mode"),
                                   nat.ptr(self._state), io.ptr("out"), None))
        io.download(first="out")
        return VehicleState(*io.h["out"][0, 0, :11].tolist())

    @property
    def time(self):
        return float(self._state[0, 44].item())

    @time.setter
    def time(self, v):
        self._state[0, 44] = float(v)

    @property
    def prev_heading(self):
        return float(self._state[0, 42].item())

    @property
    def prev_speed(self):
        return float(self._state[0, 43].item())

    # ---- reference surface ----------------------------------------------------------------------------
    def predict(self) -> VehicleState:
        return self._run(0)

    def update(self, measurement: np.ndarray) -> VehicleState:
        st = self._run(3, measurement)
        self._remember(st)
        return st

    def step(self, measurement: Optional[np.ndarray] = None) -> VehicleState:
        st = self._run(1, measurement) if measurement is not None else self._run(2)
        self._remember(st)
        return st

    def _remember(self, st):
        self.state_history.append(st)
        if measurement_history_cap < len(self.state_history):
            self.state_history = self.state_history[-measurement_history_cap:]

    def get_state_history(self, n: Optional[int] = None) -> List[VehicleState]:
        return self.state_history.copy() if n is None else self.state_history[-n:]

    def get_trajectory(self) -> np.ndarray:
        return np.array([[s.x, s.y] for s in self.state_history]) if self.state_history else np.array([])

    def get_velocity_history(self) -> np.ndarray:
        return np.array([[s.vx, s.vy] for s in self.state_history]) if self.state_history else np.array([])

    def get_speed_history(self) -> Tuple[np.ndarray, np.ndarray]:
        if not self.state_history:
            return np.array([]), np.array([])
        return (np.array([s.timestamp for s in self.state_history]), np.array([s.speed for s in self.state_history]))

    def get_heading_history(self) -> Tuple[np.ndarray, np.ndarray]:
        if not self.state_history:
            return np.array([]), np.array([])
        return (np.array([s.timestamp for s in self.state_history]), np.array([s.heading for s in self.state_history]))

    def set_initial_state(self, x: float, y: float, vx: float = 0, vy: float = 0, ax: float = 0, ay: float = 0):
        rec = self._state.cpu().numpy()[0]
        rec[:6] = (x, y, vx, vy, ax, ay)
        rec[42] = np.arctan2(vy, vx)
        rec[43] = np.sqrt(vx ** 2 + vy ** 2)
        self._state.copy_(torch.as_tensor(rec).view(1, -1))

    def reset(self):
        d = self._dev
        nat.check(d.lib.av_kf_reset(d.ctx.handle, d.stream, 1, nat.ptr(self._state)))
        self.state_history.clear()


measurement_history_cap = 1000      # vehicle_state.py:134-135
