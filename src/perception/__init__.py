from multimodal_autonomous_driving_perception_and_planning_amd.perception import (  # noqa: F401
    Detection, LaneDetector, LaneLine, ObjectDetector)

__all__ = ["ObjectDetector", "LaneDetector"]
