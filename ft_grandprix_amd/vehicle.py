"""The object handed to two-argument drivers: ``process_lidar(ranges, state)``.

Drivers written for the reference read eight attributes from ``state``; those names (and their meaning) are the
plugin contract and are kept exactly -- see ft_grandprix/vehicle.py:3-12 and the call site ft_grandprix/custom.py:1397-1399
of the reference.  Everything else here is this framework's own: the values come from one packed row of
``ftgp_get_snapshot`` (include/ftgp.h, FTGP_SNAPSHOT_DOUBLES) instead of live MuJoCo views.
"""
from __future__ import annotations

import dataclasses
from typing import Sequence

import numpy as np

# order of the doubles in one ftgp_get_snapshot row
_ROW = ("laps", "vx", "vy", "vz", "yaw", "pitch", "roll", "lap_completion", "absolute_completion", "time")


@dataclasses.dataclass
class VehicleStateSnapshot:
    laps: int                   #: completed laps (negative after crossing the line backwards)
    velocity: Sequence[float]   #: world-frame linear velocity (vx, vy, vz) -- qvel[:3] in the reference
    yaw: float                  #: heading, radians (ZYX Euler angles of the body quaternion)
    pitch: float
    roll: float
    lap_completion: int         #: percent of the current lap, negative while running a lap entered backwards
    absolute_completion: int    #: laps * 100 + lap_completion
    time: float                 #: the reference passes steps / timestep here (custom.py:1397), reproduced as is

    @classmethod
    def from_row(cls, row: Sequence[float]) -> "VehicleStateSnapshot":
        """Build a snapshot from one row of ``capi.Env.snapshot()``."""
        r = np.asarray(row, dtype=np.float64)
        if r.shape != (len(_ROW),):
            raise ValueError(f"expected {len(_ROW)} doubles, got shape {r.shape}")
        return cls(laps=int(r[0]), velocity=r[1:4].copy(), yaw=float(r[4]), pitch=float(r[5]), roll=float(r[6]),
                   lap_completion=int(r[7]), absolute_completion=int(r[8]), time=float(r[9]))

    @property
    def speed(self) -> float:
        """Ground speed, a convenience the reference does not offer."""
        return float(np.hypot(self.velocity[0], self.velocity[1]))
