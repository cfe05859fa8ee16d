from .multi_object_tracker import MultiObjectTracker, Track

__all__ = ["MultiObjectTracker", "Track"]
