from .interaction_detector import (Interaction, InteractionDetector, InteractionTags, InteractionType,  # noqa: F401
                                   RiskLevel)
from .maneuver_detector import (LateralManeuver, LongitudinalManeuver, ManeuverDetector, ManeuverTags,  # noqa: F401
                                TurningManeuver)

__all__ = ["ManeuverDetector", "ManeuverTags", "LateralManeuver", "LongitudinalManeuver", "TurningManeuver",
           "InteractionDetector", "InteractionTags", "Interaction", "InteractionType", "RiskLevel"]
