from .video_loader import VideoDataLoader  # noqa: F401

__all__ = ["VideoDataLoader"]
