from .bev_renderer import BEVRenderer
from .overlays import OverlayRenderer

__all__ = ["BEVRenderer", "OverlayRenderer"]
