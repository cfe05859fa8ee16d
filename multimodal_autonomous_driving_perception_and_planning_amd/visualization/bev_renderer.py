"""BEVRenderer -- drop-in surface of src/visualization/bev_renderer.py, drawn by the device rasteriser.

Same constructor, colours, world<->pixel mapping, layering and method names as the reference (bev_renderer.py:29-364);
every cv2 call becomes primitives of an ordered list (see _prims.py) and render() paints the whole panel in ONE launch
over the cached road image.  Pixel rules and the label font are this project's own (parity unpinned, OpenCV absent).
"""
from typing import List, Tuple

import numpy as np

from ._prims import PrimList, paint, raster_for


class BEVRenderer:
    def __init__(self, width: int = 600, height: int = 600, pixels_per_meter: float = 10.0,
                 x_range: Tuple[float, float] = (-30, 30), y_range: Tuple[float, float] = (-10, 50), device: int = 0):
        self.width, self.height = width, height
        self.pixels_per_meter = pixels_per_meter
        self.x_range, self.y_range = x_range, y_range
        self.x_scale = width / (x_range[1] - x_range[0])
        self.y_scale = height / (y_range[1] - y_range[0])
        self.bg_color = (40, 40, 40)
        self.road_color = (60, 60, 60)
        self.lane_color = (200, 200, 200)
        self.ego_color = (0, 200, 255)
        self.agent_colors = [(0, 255, 0), (255, 0, 0), (0, 0, 255), (255, 255, 0), (255, 0, 255), (0, 255, 255)]
        self._device = device
        self._base = None

    # ---- coordinates (bev_renderer.py:70-90) -------------------------------------------------------------------
    def world_to_pixel(self, x: float, y: float) -> Tuple[int, int]:
        return int((x - self.x_range[0]) * self.x_scale), int(self.height - (y - self.y_range[0]) * self.y_scale)

    def pixel_to_world(self, px: int, py: int) -> Tuple[float, float]:
        return px / self.x_scale + self.x_range[0], (self.height - py) / self.y_scale + self.y_range[0]

    # ---- primitive builders: one per reference drawing method ------------------------------------------------------
    def _base_prims(self, pl: PrimList):
        h = self.height
        road_left, road_right = self.world_to_pixel(-7, self.y_range[0])[0], self.world_to_pixel(7, self.y_range[0])[0]
        pl.rectangle((road_left, 0), (road_right, h), self.road_color, -1)
        for lane_x in (-3.5, 0, 3.5):
            px, _ = self.world_to_pixel(lane_x, 0)
            if lane_x == 0:
                for y in range(0, h, 30):
                    pl.line((px, y), (px, min(y + 15, h)), (0, 200, 200), 2)
            else:
                for y in range(0, h, 40):
                    pl.line((px, y), (px, min(y + 20, h)), self.lane_color, 2)
        for edge_x in (-7, 7):
            px, _ = self.world_to_pixel(edge_x, 0)
            pl.line((px, 0), (px, h), (255, 255, 255), 2)

    def _vehicle_prims(self, pl, x, y, heading, color, length=4.5, width=2.0, label=None):
        cos_h, sin_h = np.cos(heading), np.sin(heading)
        hl, hw = length / 2, width / 2
        corners = [(x + hl * cos_h - hw * sin_h, y + hl * sin_h + hw * cos_h), (x + hl * cos_h + hw * sin_h, y + hl * sin_h - hw * cos_h),
                   (x - hl * cos_h + hw * sin_h, y - hl * sin_h - hw * cos_h), (x - hl * cos_h - hw * sin_h, y - hl * sin_h + hw * cos_h)]
        pts = [self.world_to_pixel(cx, cy) for cx, cy in corners]
        pl.fill_convex_quad(pts, color)
        pl.polylines(pts, True, (255, 255, 255), 1)
        front = self.world_to_pixel(x + hl * cos_h, y + hl * sin_h)
        center = self.world_to_pixel(x, y)
        pl.arrowed_line(center, front, (255, 255, 255), 2, tip_length=0.5)
        if label:
            pl.put_text(label, (center[0] - 20, center[1] - 15), 0.4, (255, 255, 255), 1)

    def _agent_prims(self, pl, tracks, draw_trajectories=True):
        for track in tracks:
            color = self.agent_colors[track.track_id % len(self.agent_colors)]
            cx, cy = track.center
            self._vehicle_prims(pl, (cx - 320) * 0.03, 50 - cy * 0.1, 0, color, length=3.0, width=1.5, label="ID:%d" % track.track_id)
            n = len(track.trajectory)
            if draw_trajectories and n > 1:
                for j in range(1, n):
                    (pcx, pcy), (ccx, ccy) = track.trajectory[j - 1], track.trajectory[j]
                    a = self.world_to_pixel((pcx - 320) * 0.03, 50 - pcy * 0.1)
                    b = self.world_to_pixel((ccx - 320) * 0.03, 50 - ccy * 0.1)
                    pl.line(a, b, color, max(1, int(2 * (j / n))))

    def _trajectory_prims(self, pl, trajectory, color=(0, 255, 0), thickness=2, draw_waypoints=True):
        if not trajectory or not trajectory.waypoints:
            return
        positions = trajectory.get_positions()
        if len(positions) < 2:
            return
        pl.polylines([self.world_to_pixel(p[0], p[1]) for p in positions], False, color, thickness)
        if draw_waypoints:
            for p in positions[::3]:
                pl.circle(self.world_to_pixel(p[0], p[1]), 3, color, -1)

    def _uncertainty_prims(self, pl, x, y, uncertainty, color=(0, 255, 255)):
        radius = int(uncertainty * self.pixels_per_meter)
        if radius > 0:
            pl.circle(self.world_to_pixel(x, y), radius, color, 1)

    def _legend_prims(self, pl):
        y = 20
        for label, color in (("EGO", self.ego_color), ("Planned", (0, 255, 0)), ("Agents", self.agent_colors[0])):
            pl.rectangle((10, y - 10), (25, y + 5), color, -1)
            pl.put_text(label, (30, y), 0.4, (255, 255, 255), 1)
            y += 20

    def _grid_prims(self, pl):
        for x in range(-30, 31, 10):
            px, _ = self.world_to_pixel(x, 0)
            pl.line((px, 0), (px, self.height), (50, 50, 50), 1)
            pl.put_text("%dm" % x, (px, self.height - 5), 0.3, (100, 100, 100), 1)
        for y in range(-10, 51, 10):
            _, py = self.world_to_pixel(0, y)
            pl.line((0, py), (self.width, py), (50, 50, 50), 1)
            pl.put_text("%dm" % y, (5, py), 0.3, (100, 100, 100), 1)

    # ---- reference surface -----------------------------------------------------------------------------------------
    def create_base_image(self) -> np.ndarray:
        if self._base is None:
            pl = PrimList()
            self._base_prims(pl)
            self._base = paint(np.full((self.height, self.width, 3), self.bg_color, np.uint8), pl, self._device)
        return self._base.copy()

    def draw_vehicle(self, img, x, y, heading, color, length: float = 4.5, width: float = 2.0, label: str = None) -> np.ndarray:
        pl = PrimList()
        self._vehicle_prims(pl, x, y, heading, color, length, width, label)
        img[...] = paint(img, pl, self._device)
        return img

    def draw_ego_vehicle(self, img, state) -> np.ndarray:
        return self.draw_vehicle(img, state.x, state.y, state.heading, self.ego_color, label="EGO")

    def draw_agents(self, img, tracks: List, draw_trajectories: bool = True) -> np.ndarray:
        pl = PrimList()
        self._agent_prims(pl, tracks, draw_trajectories)
        img[...] = paint(img, pl, self._device)
        return img

    def draw_trajectory(self, img, trajectory, color=(0, 255, 0), thickness: int = 2, draw_waypoints: bool = True) -> np.ndarray:
        pl = PrimList()
        self._trajectory_prims(pl, trajectory, color, thickness, draw_waypoints)
        img[...] = paint(img, pl, self._device)
        return img

    def draw_uncertainty_ellipse(self, img, x, y, uncertainty, color=(0, 255, 255)) -> np.ndarray:
        pl = PrimList()
        self._uncertainty_prims(pl, x, y, uncertainty, color)
        img[...] = paint(img, pl, self._device)
        return img

    def scene_prims(self, ego_state=None, tracks=None, planned_trajectory=None, candidate_trajectories=None, show_grid=False):
        """The whole panel of render() above the road image, as one ordered list (bev_renderer.py:304-348)."""
        pl = PrimList()
        if show_grid:
            self._grid_prims(pl)
        if candidate_trajectories:
            for traj in candidate_trajectories:
                if traj != planned_trajectory:
                    self._trajectory_prims(pl, traj, (80, 80, 80), 1, False)
        if planned_trajectory:
            self._trajectory_prims(pl, planned_trajectory, (0, 255, 0), 3, True)
        if tracks:
            self._agent_prims(pl, tracks)
        if ego_state:
            self._vehicle_prims(pl, ego_state.x, ego_state.y, ego_state.heading, self.ego_color, label="EGO")
            if hasattr(ego_state, "pos_uncertainty"):
                self._uncertainty_prims(pl, ego_state.x, ego_state.y, ego_state.pos_uncertainty)
        self._legend_prims(pl)
        return pl

    def render(self, ego_state=None, tracks: List = None, planned_trajectory=None, candidate_trajectories: List = None,
               show_grid: bool = False) -> np.ndarray:
        base = self.create_base_image()
        pl = self.scene_prims(ego_state, tracks, planned_trajectory, candidate_trajectories, show_grid)
        return raster_for(self.height, self.width, self._device).paint(base, pl)

    def _draw_legend(self, img):
        pl = PrimList()
        self._legend_prims(pl)
        img[...] = paint(img, pl, self._device)
