from multimodal_autonomous_driving_perception_and_planning_amd.visualization import BEVRenderer, OverlayRenderer  # noqa: F401

__all__ = ["BEVRenderer", "OverlayRenderer"]
