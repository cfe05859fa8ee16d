from multimodal_autonomous_driving_perception_and_planning_amd.tagging import (  # noqa: F401
    LateralManeuver, LongitudinalManeuver, ManeuverDetector, ManeuverTags, TurningManeuver)

__all__ = ["ManeuverDetector"]
