"""Oracle P1-P3: quintic-lateral candidate planner (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Array restatement of MotionPlanner (src/planning/motion_planner.py):

  generate_polynomial_trajectory :126-204
      v_i = v0 + (vt - v0)(1 - exp(-t_i));  s_i = s_{i-1} + v_i*dt  (sequential, :151-157)
      d_i = df*(10 tau^3 - 15 tau^4 + 6 tau^5), tau = clip(t_i/H, 0, 1)        (:164-169)
      x = x0 + s cos h0 + d cos(h0+pi/2),  y likewise with sin                  (:175-180)
      heading_i = atan2(y_{i+1}-y_i, x_{i+1}-x_i), last = previous              (:183-190)
      curv_i = (heading_i - heading_{i-1}) / (v_i*dt + 1e-6) for 0<i<n-1        (:193-196)
  evaluate_trajectory_cost :206-262   left-to-right float64 accumulation:
      [ref-path min-distance^2] -> velocity -> acceleration -> curvature -> obstacles
  plan :264-303   lateral offsets (outer) x speeds {8,10,12} (inner), stable ascending sort.

Everything is evaluated with the same elementary operations, in the same order,
as the reference's scalar loops, so the result is bit-identical on the same
machine (verified against tests/golden/planner.npz).
"""
import numpy as np

SPEEDS = (8.0, 10.0, 12.0)                       # motion_planner.py:280
WP_FIELDS = ("x", "y", "heading", "velocity", "timestamp", "curvature")
TYPE_CODE = {"lane_keep": 0, "lane_change_left": 1, "lane_change_right": 2}


class PlannerRef:
    def __init__(self, planning_horizon=5.0, dt=0.1, num_samples=7):
        self.H = planning_horizon
        self.dt = dt
        self.num_samples = num_samples
        self.w_lateral, self.w_velocity, self.w_acceleration = 1.0, 0.5, 0.3   # :85-87
        self.w_jerk, self.w_curvature = 0.2, 0.4                               # :88-89 (w_jerk unused)
        self.ref = None
        self.n = int(self.H / self.dt) + 1                                     # :143
        self.t = np.linspace(0, self.H, self.n)                                # :144
        self.alpha = np.array([1 - np.exp(-np.float64(ti)) for ti in self.t])  # :153
        q = np.zeros(self.n)
        for i, ti in enumerate(self.t):                                        # :166-169
            tau = np.clip(ti / self.H, 0, 1)
            q[i] = 10 * tau ** 3 - 15 * tau ** 4 + 6 * tau ** 5
        self.q = q
        self.lat = np.linspace(-3.5, 3.5, self.num_samples)                    # :279

    def set_reference_path(self, pts):
        pts = np.asarray(pts, np.float64).reshape(-1, 2)
        if len(pts) < 2:                                                       # :100-101
            return
        self.ref = pts

    def reset(self):
        self.ref = None

    # ---- P1 -------------------------------------------------------------
    def generate(self, state, df, vt):
        """-> float64[n,6] rows (x, y, heading, velocity, timestamp, curvature)."""
        x0, y0, h0, v0 = (np.float64(v) for v in state)
        n = self.n
        v = v0 + (np.float64(vt) - v0) * self.alpha
        inc = v * self.dt
        inc[0] = 0.0
        s = np.cumsum(inc)                                   # sequential prefix sum == :156-157
        d = np.float64(df) * self.q
        c, sn = np.cos(h0), np.sin(h0)
        c2, sn2 = np.cos(h0 + np.pi / 2), np.sin(h0 + np.pi / 2)
        x = x0 + s * c
        y = y0 + s * sn
        x = x + d * c2
        y = y + d * sn2
        hd = np.empty(n)
        hd[:-1] = np.arctan2(y[1:] - y[:-1], x[1:] - x[:-1])
        hd[-1] = hd[-2] if n > 1 else h0
        cv = np.zeros(n)
        if n > 2:
            cv[1:-1] = (hd[1:-1] - hd[:-2]) / (v[1:-1] * self.dt + 1e-6)
        return np.stack([x, y, hd, v, self.t, cv], axis=1)

    # ---- P2 -------------------------------------------------------------
    def cost(self, wp, obstacles=None):
        terms = []
        if self.ref is not None:                             # :224-231
            dx = self.ref[None, :, 0] - wp[:, None, 0]
            dy = self.ref[None, :, 1] - wp[:, None, 1]
            md = np.sqrt(dx * dx + dy * dy).min(axis=1)
            terms.append(self.w_lateral * md ** 2)
        terms.append(self.w_velocity * (wp[:, 3] - 10.0) ** 2)                 # :234-236
        dts = wp[1:, 4] - wp[:-1, 4]                                           # :239-244
        acc = (wp[1:, 3] - wp[:-1, 3]) / np.where(dts > 0, dts, 1.0)
        terms.append(np.where(dts > 0, self.w_acceleration * acc ** 2, 0.0))
        terms.append(self.w_curvature * wp[:, 5] ** 2)                         # :247-248
        if obstacles is not None and len(obstacles):                           # :251-259
            for ox, oy, r in obstacles:
                dist = np.sqrt((wp[:, 0] - ox) ** 2 + (wp[:, 1] - oy) ** 2)
                hard = 1000 * (r * 2 - dist)
                with np.errstate(divide="ignore"):
                    soft = 10 / (dist - r + 0.1)
                terms.append(np.where(dist < r * 2, hard, np.where(dist < r * 4, soft, 0.0)))
        allt = np.concatenate([[0.0]] + terms)
        return float(np.cumsum(allt)[-1])                    # strict left-to-right accumulation

    # ---- P3 -------------------------------------------------------------
    def plan(self, state, obstacles=None):
        """-> dict(wp [C,n,6] generation order, cost [C], order [C] sorted->generation, types [C])."""
        C = self.num_samples * len(SPEEDS)
        wp = np.zeros((C, self.n, 6))
        cost = np.zeros(C)
        types = np.zeros(C, np.int32)
        k = 0
        for df in self.lat:
            for vt in SPEEDS:
                wp[k] = self.generate(state, df, vt)
                cost[k] = self.cost(wp[k], obstacles)
                types[k] = 0 if abs(df) < 0.5 else (1 if df < 0 else 2)        # :289-294
                k += 1
        order = np.argsort(cost, kind="stable").astype(np.int32)               # :300
        return dict(wp=wp, cost=cost, order=order, types=types)


def trajectory_length(wp):
    """Trajectory.length (motion_planner.py:33-43)."""
    dx = wp[1:, 0] - wp[:-1, 0]
    dy = wp[1:, 1] - wp[:-1, 1]
    seg = np.sqrt(dx ** 2 + dy ** 2)
    return float(np.cumsum(np.concatenate([[0.0], seg]))[-1])
