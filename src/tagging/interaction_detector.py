from multimodal_autonomous_driving_perception_and_planning_amd.tagging.interaction_detector import (  # noqa: F401
    Interaction, InteractionDetector, InteractionTags, InteractionType, RiskLevel)
