"""Import-path shim: `from data.loaders.video_loader import VideoDataLoader` (demo.py:21, app.py:30)."""
from multimodal_autonomous_driving_perception_and_planning_amd.loaders import VideoDataLoader  # noqa: F401

__all__ = ["VideoDataLoader"]
