from multimodal_autonomous_driving_perception_and_planning_amd.state_estimation import (  # noqa: F401
    VehicleState, VehicleStateEstimator)

__all__ = ["VehicleStateEstimator", "VehicleState"]
