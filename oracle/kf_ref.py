"""Oracle E1-E3: 6-state constant-acceleration Kalman estimator (TEST INFRASTRUCTURE).

PARITY UNPINNED: filterpy (requirements.txt:5, >=1.4.5, no lock) is not installed
in the build container and the reference ships no tests, so nothing pins these
numbers to the real dependency.  Restates filterpy 1.4.5 KalmanFilter as used by
src/state_estimation/vehicle_state.py:

  _init_kalman_filter :68-106   F (CA model), H=[I4 0], R=r*I4, Q=diag(q,q,q,q,10q,10q), P0=10*I6, x0=0
  predict  (filterpy)           x = F x ; P = alpha^2 F P F^T + Q   (alpha = 1)
  update   (filterpy)           y = z - H x ; PHT = P H^T ; S = H PHT + R ; SI = inv(S)
                                K = PHT SI ; x = x + K y ; I_KH = I - K H
                                P = I_KH P I_KH^T + K R K^T      (Joseph form)
  step :139-156                 predict() then update(z) (or extract+append when z is None)
  _extract_state :158-198       called in predict() AND update(): prev_heading/prev_speed
                                are overwritten twice per step
"""
import numpy as np

STATE_FIELDS = ("x", "y", "vx", "vy", "heading", "speed", "acceleration", "yaw_rate",
                "timestamp", "pos_uncertainty", "vel_uncertainty", "heading_uncertainty")


def make_F(dt):
    return np.array([[1, 0, dt, 0, 0.5 * dt ** 2, 0],
                     [0, 1, 0, dt, 0, 0.5 * dt ** 2],
                     [0, 0, 1, 0, dt, 0],
                     [0, 0, 0, 1, 0, dt],
                     [0, 0, 0, 0, 1, 0],
                     [0, 0, 0, 0, 0, 1]], np.float64)


class KalmanRef:
    def __init__(self, dt=0.033, process_noise=0.1, measurement_noise=1.0):
        self.dt = dt
        self.F = make_F(dt)
        self.Hm = np.zeros((4, 6))
        self.Hm[:4, :4] = np.eye(4)
        self.R = np.eye(4) * measurement_noise
        self.Q = np.eye(6) * process_noise
        self.Q[4, 4] *= 10
        self.Q[5, 5] *= 10
        self.reset()

    def reset(self):                                             # vehicle_state.py:250-257
        self.x = np.zeros(6)
        self.P = np.eye(6) * 10
        self.time = 0.0
        self.prev_heading = 0.0
        self.prev_speed = 0.0
        self.history = []

    def set_initial_state(self, x, y, vx=0, vy=0, ax=0, ay=0):   # :242-248
        self.x = np.array([x, y, vx, vy, ax, ay], np.float64)
        self.prev_heading = float(np.arctan2(vy, vx))
        self.prev_speed = float(np.sqrt(vx ** 2 + vy ** 2))

    def _extract(self):                                          # :158-198
        x, y, vx, vy, ax, ay = self.x
        speed = np.sqrt(vx ** 2 + vy ** 2)
        heading = np.arctan2(vy, vx) if speed > 0.1 else self.prev_heading
        acc = (speed - self.prev_speed) / self.dt if self.dt > 0 else 0
        dh = heading - self.prev_heading
        if dh > np.pi:
            dh -= 2 * np.pi
        elif dh < -np.pi:
            dh += 2 * np.pi
        yaw = dh / self.dt if self.dt > 0 else 0
        pu = np.sqrt(self.P[0, 0] + self.P[1, 1])
        vu = np.sqrt(self.P[2, 2] + self.P[3, 3])
        self.prev_heading = heading
        self.prev_speed = speed
        return np.array([x, y, vx, vy, heading, speed, acc, yaw, self.time, pu, vu, 0.0])

    def predict(self):                                           # :108-117
        self.x = self.F @ self.x
        self.P = self.F @ self.P @ self.F.T + self.Q
        self.time += self.dt
        return self._extract()

    def update(self, z):                                         # :119-137
        z = np.asarray(z, np.float64).reshape(4)
        H, R = self.Hm, self.R
        y = z - H @ self.x
        PHT = self.P @ H.T
        S = H @ PHT + R
        SI = np.linalg.inv(S)
        K = PHT @ SI
        self.x = self.x + K @ y
        I_KH = np.eye(6) - K @ H
        self.P = I_KH @ self.P @ I_KH.T + K @ R @ K.T
        st = self._extract()
        self.history.append(st)
        if len(self.history) > 1000:
            self.history = self.history[-1000:]
        return st

    def step(self, z=None):                                      # :139-156
        self.predict()
        if z is not None:
            return self.update(z)
        st = self._extract()
        self.history.append(st)
        return st
