from .motion_planner import MotionPlanner, Trajectory, Waypoint

__all__ = ["MotionPlanner", "Trajectory", "Waypoint"]
