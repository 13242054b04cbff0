"""Second argument of v2 drivers.  Same 8 field names as the reference's dataclass (ft_grandprix/vehicle.py:3-12)."""
from dataclasses import dataclass


@dataclass
class VehicleStateSnapshot:
    laps: int
    velocity: list
    yaw: float
    pitch: float
    roll: float
    lap_completion: int
    absolute_completion: int
    time: float
