"""Import-path shim for the reference's `data` package (data/__init__.py:3 exports VideoDataLoader)."""
from .loaders import VideoDataLoader  # noqa: F401

__all__ = ["VideoDataLoader"]
