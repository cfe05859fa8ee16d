from .detector import Detection, ObjectDetector
from .lane_detector import LaneDetector, LaneLine

__all__ = ["ObjectDetector", "LaneDetector", "Detection", "LaneLine"]
