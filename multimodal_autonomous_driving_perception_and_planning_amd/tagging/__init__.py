from .maneuver_detector import (LateralManeuver, LongitudinalManeuver, ManeuverDetector, ManeuverTags,  # noqa: F401
                                TurningManeuver)

__all__ = ["ManeuverDetector", "ManeuverTags", "LateralManeuver", "LongitudinalManeuver", "TurningManeuver"]
