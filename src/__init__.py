"""Import-path shim: `from src.perception import ObjectDetector, LaneDetector` etc. resolve to the
MI355X-native classes, so the reference's demo.py / app.py import lines work unchanged
(demo.py:17-20, app.py:26-29)."""
