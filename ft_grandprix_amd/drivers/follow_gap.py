"""A small driver of this package's own for ``python -m ft_grandprix_amd.sim``: head for the widest opening.

Written against the plugin contract only (drivers/template.py of the reference): ``ranges`` is a 1-D float array, index 0 =
rear, counter-clockwise, world units, all 0 right after a reset; the return value is (speed, steering_angle)."""
import numpy as np


class Driver:
    FIELD_OF_VIEW = np.pi            # radians, centred on the heading
    SMOOTH = 9                       # samples of the moving average that hides single-ray gaps
    CRUISE, SLOW = 1.6, 0.7

    def process_lidar(self, ranges):
        r = np.asarray(ranges, dtype=np.float64)
        n = r.size
        if n < 4 or not np.any(r > 0):
            return 0.0, 0.0                              # nothing seen yet (first step after a reset)
        r = np.where(r < 0, r.max(), r)                  # "no hit" counts as far away
        half = max(1, int(n * self.FIELD_OF_VIEW / (4 * np.pi)))
        front = n // 2                                   # index 0 looks backwards, so n / 2 looks ahead
        window = r[front - half: front + half + 1]
        k = min(self.SMOOTH, window.size)
        smooth = np.convolve(window, np.ones(k) / k, mode="same")
        target = int(np.argmax(smooth)) - half
        steering = float(np.clip(target * (2 * np.pi / n), -1.0, 1.0))
        ahead = float(smooth[half])
        speed = self.CRUISE if abs(steering) < 0.25 and ahead > 1.0 else self.SLOW
        return speed, steering
