from multimodal_autonomous_driving_perception_and_planning_amd.planning import (  # noqa: F401
    MotionPlanner, Trajectory, Waypoint)

__all__ = ["MotionPlanner", "Trajectory", "Waypoint"]
