from .vehicle_state import VehicleState, VehicleStateEstimator

__all__ = ["VehicleStateEstimator", "VehicleState"]
