from multimodal_autonomous_driving_perception_and_planning_amd.tagging.maneuver_detector import *  # noqa: F401,F403
from multimodal_autonomous_driving_perception_and_planning_amd.tagging.maneuver_detector import (  # noqa: F401
    LateralManeuver, LongitudinalManeuver, ManeuverDetector, ManeuverTags, TurningManeuver)
