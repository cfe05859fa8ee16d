from multimodal_autonomous_driving_perception_and_planning_amd.tagging import (  # noqa: F401
    Interaction, InteractionDetector, InteractionTags, InteractionType, LateralManeuver, LongitudinalManeuver,
    ManeuverDetector, ManeuverTags, RiskLevel, TurningManeuver)

__all__ = ["ManeuverDetector", "InteractionDetector"]
