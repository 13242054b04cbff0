"""Host-side mirror of the reference's driver plugin surface, on top of the C-ABI.

What is mirrored (reference paths):
  * roster entries {"driver": "file://a/b.py" | "a.b", "name": ...} -> module path -> ``module.Driver()``
    with a null-driver fallback on import failure -- ft_grandprix/custom.py:1096-1109, template/cars/cars.json
  * arity sniffing: ``process_lidar(ranges)`` or ``process_lidar(ranges, snapshot)`` -- custom.py:103,1398-1399
  * per step and per car: ranges (1-D float64, index 0 = rear, CCW), snapshot (VehicleStateSnapshot),
    ``speed, steering_angle = driver.process_lidar(...)``; an exception prints a message and leaves that
    car's previous controls in place -- custom.py:1395-1411,1421-1423
  * finished cars are handed the null driver -- custom.py:1367-1371,1446
  * ``winners``: {vehicle id: place} of every finisher, in the order cars reach ``lap_target`` -- custom.py:1125,1367-1369 --
    derived from the device's finish step, so it is the same after one ``rollout()`` to the end of the race as after stepping
    through it.  With one world (``n_envs == 1``, the reference's case) it is exactly the reference's dict; with several it
    holds every world's finishers under their global vehicle ids (places count per world) and ``winners_by_env[e]`` is world
    e's own dict
  * ``python -m ft_grandprix_amd.sim --cars template/cars/cars.json --track track --steps N``: the headless form of
    ``python -m ft_grandprix.drive`` (drive.py:69-115: roster + track in, physics loop out)
  * ``reset()`` = Mujoco.reload(): drivers re-instantiated, race state cleared, cars re-spawned -- custom.py:1089-1128
  * the options that change a step (plain constructor arguments / attributes here):
    ``detach_control`` (custom.py:952,1421-1423: drivers run, ``data.ctrl`` is not written), ``manual_control`` + ``watching`` +
    ``manual_speed`` / ``manual_steering_angle`` (custom.py:954-957,1413-1416: the watched car takes the keyboard's controls, coasting
    at 0.99 of its throttle when no key is held), ``always_invoke_driver`` (custom.py:956,1403: with manual control on, whether the
    drivers are still called), ``rangefinder_tilt`` (custom.py:986,1387: FAKELIDAR mode only, see ``tilted_fan``)
The physics / LiDAR / lap logic themselves run on the GPU behind ``capi.Env``.
"""
from __future__ import annotations

import importlib
import inspect
import json
import math
from typing import List, Optional, Sequence

import numpy as np

from . import capi
from .track import Track
from .vehicle import VehicleStateSnapshot


class LobotomyDriver:
    """Null driver: (0, 0).  Fallback on import failure and for finished cars (lobotomy.py:1-3)."""

    def process_lidar(self, ranges):
        return 0, 0


def lap_completion(completion: int, good_start: bool) -> int:
    """custom.py:132-140: negative while running a lap that was entered backwards."""
    return completion if good_start else -(100 - completion)


def absolute_completion(laps: int, completion: int, good_start: bool) -> int:
    """custom.py:142-143."""
    return laps * 100 + lap_completion(completion, good_start)


def quaternion_to_euler(w, x, y, z):
    """[yaw, pitch, roll] with the asin clamp of custom.py:62-76."""
    roll = math.atan2(2.0 * (w * x + y * z), 1.0 - 2.0 * (x * x + y * y))
    t2 = 2.0 * (w * y - z * x)
    t2 = 1.0 if t2 > 1.0 else t2
    t2 = -1.0 if t2 < -1.0 else t2
    pitch = math.asin(t2)
    yaw = math.atan2(2.0 * (w * z + x * y), 1.0 - 2.0 * (y * y + z * z))
    return [yaw, pitch, roll]


def ordinal(n: int) -> str:
    """1 -> '1st', 2 -> '2nd', 11 -> '11th', 0 -> '0th' ... the dashboard's position label (custom.py:47-55, used at custom.py:358)."""
    text = str(n)
    teen = len(text) > 1 and text[-2] == "1"
    suffix = "th" if (text == "0" or teen) else {"1": "st", "2": "nd", "3": "rd"}.get(text[-1], "th")
    return text + suffix


def tilted_fan(n_rays: int, tilt: float) -> np.ndarray:
    """The FAKELIDAR fan under the option ``rangefinder_tilt`` (custom.py:986,1387): the reference lays its scan angles out as
    ``linspace(tilt + yaw + pi, yaw - pi, r, endpoint=False)`` -- the tilt moves the START of the sweep and not its end, so scan
    i is turned by ``tilt * (1 - i / r)``: the rear ray by the whole tilt, the last one by tilt / r.  Those angles live in the
    y-down pixel frame, where a positive turn is a clockwise one in the world; the rangefinders' own order (index 0 = rear,
    counter-clockwise, include/ftgp.h) is kept, as SURVEY.md 8a-3 prescribes for this never-exercised branch.  Returns the
    body-frame directions [n_rays, 2] (binary64) for ``FtgpConfig.fan_dirs``; tilt = 0 gives the default fan bit for bit."""
    j = np.arange(n_rays, dtype=np.float64)
    phi = np.array([math.radians(360.0 / n_rays * k - 90.0) for k in range(n_rays)])       # mushr.em.xml:112-117
    if tilt == 0.0:
        return np.stack([np.array([math.sin(p) for p in phi]), np.array([-math.cos(p) for p in phi])], axis=1)
    turned = phi - tilt * (1.0 - j / n_rays)
    return np.stack([np.sin(turned), -np.cos(turned)], axis=1)


def resolve_driver_path(spec: str) -> Optional[str]:
    """Roster 'driver' string -> importable module path (custom.py:1097-1104)."""
    if spec.startswith("file://"):
        return spec[7:-3].replace("/", ".")
    if "//" not in spec:
        return spec
    print("Unsupported schema: supported (file://)")
    return None


def load_driver(path: Optional[str]):
    try:
        return importlib.import_module(path).Driver()
    except Exception:
        return LobotomyDriver()


class VehicleState:
    """Per-car race state + driver handle (the fields of custom.py:91-126 that drivers and dashboards read)."""

    def __init__(self, id: int, offset: int, driver, label: str, driver_path: Optional[str]):
        self.id, self.offset, self.driver, self.label, self.driver_path = id, offset, driver, label, driver_path
        self.v2 = len(inspect.signature(self.driver.process_lidar).parameters) >= 2
        self.completion, self.laps, self.start, self.delta = 0, 0, 0, 0
        self.good_start, self.finished, self.off_track = True, False, False
        self.speed, self.steering_angle = 0.0, 0.0
        self.times: List[float] = []
        self.n_times = 0

    def lap_completion(self):
        return lap_completion(self.completion, self.good_start)

    def absolute_completion(self):
        return absolute_completion(self.laps, self.completion, self.good_start)

    def reload_code(self):
        self.driver = load_driver(self.driver_path)
        self.v2 = len(inspect.signature(self.driver.process_lidar).parameters) >= 2


class Simulator:
    """Batched worlds with Python drivers: one ``Driver`` instance per car, called in car order every step."""

    def __init__(self, track: Track, cars: Sequence[dict], n_envs: int = 1, n_rays: int = 90, lap_target: int = 10,
                 lib: Optional[capi.CLib] = None, spawn_mode: int = 0, seed: int = 1234, device_id: int = 0,
                 lidar_mode="rangefinder", detach_control: bool = False, manual_control: bool = False,
                 always_invoke_driver: bool = True, rangefinder_tilt: float = 0.0):
        self.lib = lib if lib is not None else capi.load()
        self.cars = list(cars)
        # options that change a step (custom.py:952-957,986); plain attributes: a caller may flip them between steps, as the GUI does
        self.detach_control, self.manual_control, self.always_invoke_driver = detach_control, manual_control, always_invoke_driver
        self.watching: Optional[int] = 0                             # custom.py:931: the car the manual controls go to
        self.manual_speed, self.manual_steering_angle = 0.0, 0.0     # ModelAndView.speed / .steering_angle (custom.py:465-482): what the keys hold
        self.rangefinder_tilt = float(rangefinder_tilt)
        fan = tilted_fan(n_rays, self.rangefinder_tilt) if (lidar_mode == "fakelidar" and self.rangefinder_tilt != 0.0) else None
        self.env = capi.Env(self.lib, track, n_envs=n_envs, cars_per_env=len(self.cars), n_rays=n_rays,
                            lap_target=lap_target, spawn_mode=spawn_mode, seed=seed, device_id=device_id, lidar_mode=lidar_mode, fan_dirs=fan)
        self.n_envs, self.cars_per_env, self.n_rays = n_envs, len(self.cars), n_rays
        self.timestep = self.env.dt
        self.steps = 0
        self.vehicle_states: List[VehicleState] = []
        self.winners: dict = {}                                      # {vehicle id: place}, custom.py:1125,1368-1369
        self.winners_by_env: List[dict] = [{} for _ in range(n_envs)]
        self.reset()

    @staticmethod
    def load_roster(path: str) -> list:
        with open(path) as f:
            return json.load(f)

    def close(self):
        self.env.close()

    # reload(): custom.py:1089-1128
    def reset(self):
        self.env.reset()
        states = []
        for e in range(self.n_envs):
            for i, car in enumerate(self.cars):
                path = resolve_driver_path(car["driver"])
                states.append(VehicleState(id=e * self.cars_per_env + i, offset=(i + 5) * 2, driver=load_driver(path),
                                           label=car.get("name", f"car #{i}"), driver_path=path))
        self.vehicle_states = states
        self.steps = 0
        self.winners = {}
        self.winners_by_env = [{} for _ in range(self.n_envs)]
        self._sync_race_state()

    def _sync_race_state(self):
        prog = self.env.progress()
        counts, times = self.env.lap_times()
        places = self.env.winners().reshape(-1)
        for vs in self.vehicle_states:
            p = prog[vs.id]
            vs.laps, vs.completion, vs.start, vs.delta = int(p[0]), int(p[1]), int(p[6]), int(p[8])
            vs.good_start, vs.off_track = bool(p[7]), bool(p[5])
            vs.times = capi.lap_time_list(counts[vs.id], times[vs.id])     # the newest MAX_LAP_TIMES of them, oldest first
            vs.n_times = int(counts[vs.id])                                # len(times) in the reference (unbounded there)
            if p[4] and not vs.finished:           # custom.py:1367-1371 + 1446
                self.winners[vs.id] = int(places[vs.id])
                self.winners_by_env[vs.id // self.cars_per_env][vs.id] = int(places[vs.id])
                vs.finished = True
                vs.driver = LobotomyDriver()
                vs.v2 = False

    def snapshots(self) -> List[VehicleStateSnapshot]:
        snap = self.env.snapshot()
        return [VehicleStateSnapshot.from_row(row) for row in snap]

    def step(self):
        """One iteration of the reference's physics loop body (custom.py:1337-1426)."""
        ranges = self.env.lidar().astype(np.float64)
        snaps = self.snapshots() if any(vs.v2 for vs in self.vehicle_states) else None
        ctrl = np.zeros((self.env.n_cars, 2))
        mask = np.ones(self.env.n_cars, dtype=np.uint8)
        current = None
        for vs in self.vehicle_states:
            args = [ranges[vs.id], snaps[vs.id]] if vs.v2 else [ranges[vs.id]]
            speed, steering_angle = 0.0, 0.0                             # custom.py:1401
            if self.always_invoke_driver or not self.manual_control:     # custom.py:1403
                try:
                    speed, steering_angle = vs.driver.process_lidar(*args)
                except Exception as e:
                    print(f"Error in vehicle `{vs.label}`: `{e}`")
                    mask[vs.id] = 0                                      # `continue`: this car's ctrl stays what it was (custom.py:1409-1411)
                    continue
            if self.manual_control and self.watching == vs.id:           # custom.py:1413-1416
                speed, steering_angle = self.manual_speed, self.manual_steering_angle
                if current is None:
                    current = self.env.ctrl()
                if speed == 0.0 and current[vs.id, 0] > 0.0:
                    speed = current[vs.id, 0] * 0.99
            vs.speed, vs.steering_angle = speed, steering_angle
            ctrl[vs.id] = (speed, steering_angle)
        if self.detach_control:                                          # custom.py:1421-1423: nobody's data.ctrl is written
            mask[:] = 0
        self.env.set_ctrl(ctrl, mask)
        self.env.step(1)
        self.steps += 1
        self._sync_race_state()

    def drive(self, n_steps: int):
        for _ in range(n_steps):
            self.step()

    # the reference's bundled drivers, restated on the device (K5): roster module path -> device policy
    BUNDLED = {"ft_grandprix.nidc": "nidc", "ft_grandprix.fast": "fast", "ft_grandprix.lobotomy": "lobotomy"}

    def roster_policies(self) -> Optional[List[str]]:
        """The device policy of every roster entry, or None when some entry is not one of the reference's bundled drivers
        (``ft_grandprix.nidc`` / ``.fast`` / ``.lobotomy``, as module path or file:// string, custom.py:1097-1104)."""
        names = [self.BUNDLED.get(resolve_driver_path(car["driver"]) or "") for car in self.cars]
        return None if any(n is None for n in names) else names

    def rollout(self, policy: Optional[str], n_steps: int):
        """n_steps on the device in ONE launch: with one of the on-device drivers ("nidc", "fast", "lobotomy", "random") for every
        car, or -- policy "roster" / None -- with every car's own bundled driver as the roster names them (template/cars/cars.json:
        nidc, fast, nidc).  The race state -- laps, lap times, finished, winners -- is the same as after n_steps single steps."""
        if policy in (None, "roster", "per_car"):
            names = self.roster_policies()
            if names is None:
                raise ValueError("the roster has drivers that exist only in Python: use step() / drive(), or name a device policy")
            self.env.set_car_policies(names)
            policy = "per_car"
        self.env.rollout(policy, n_steps)
        self.steps += n_steps
        self._sync_race_state()

    def podium(self, env: int = 0) -> List[int]:
        """Vehicle ids of env's finishers, winner first (the order Mujoco.winners was filled in, custom.py:1368-1369)."""
        w = self.winners_by_env[env]
        return sorted(w, key=w.get)

    def ranking(self) -> List[int]:
        """Car ids by absolute completion, best first (the dashboard order of custom.py:335)."""
        return [vs.id for vs in sorted(self.vehicle_states, key=lambda v: -v.absolute_completion())]


def main(argv=None):
    """Headless runner in the shape of the reference's legacy loop (``python -m ft_grandprix.drive``, drive.py:69-115): a roster
    (template/cars/cars.json layout) and a track in, the physics loop out -- sense -> Driver.process_lidar -> ctrl -> step --
    without the viewer and without the real-time sleep.  Prints one line per car at the end (and every --report steps)."""
    import argparse
    import os
    import sys
    from .track import load_track, load_track_from_template
    ap = argparse.ArgumentParser(prog="python -m ft_grandprix_amd.sim", description=main.__doc__)
    ap.add_argument("--cars", default=None, help="roster JSON: [{\"driver\": \"file://pkg/mod.py\" | \"pkg.mod\", \"name\": ...}, ...] "
                                                 "(template/cars/cars.json); default: one car driven by --driver")
    ap.add_argument("--driver", default="ft_grandprix_amd.drivers.follow_gap", help="driver module of the default one-car roster")
    ap.add_argument("--track", default="track", help="name of a shipped track blob, or of <template-dir>/<name>.png + <name>-path.svg")
    ap.add_argument("--template-dir", default=None, help="directory holding <track>.png and <track>-path.svg (the reference's template/)")
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--rays", type=int, default=90, help="rangefinders per car (custom.py:1158 uses 90)")
    ap.add_argument("--envs", type=int, default=1, help="independent copies of the world (every copy runs the whole roster)")
    ap.add_argument("--lap-target", type=int, default=10)
    ap.add_argument("--device-policy", default=None, choices=("nidc", "fast", "lobotomy", "random", "roster"),
                    help="run the whole loop on the device (ONE launch) instead of calling the roster's Python drivers: with this built-in driver "
                         "for every car, or -- roster -- with each car's own, when the roster names only the reference's bundled drivers")
    ap.add_argument("--lidar", default="rangefinder", choices=("rangefinder", "fakelidar"))
    ap.add_argument("--report", type=int, default=0, help="print the standings every this many steps")
    args = ap.parse_args(argv)
    sys.path.insert(0, os.getcwd())                      # roster entries are module paths relative to the working directory, as in the reference
    roster = Simulator.load_roster(args.cars) if args.cars else [{"driver": args.driver, "name": "car #0"}]
    track = load_track_from_template(args.template_dir, args.track) if args.template_dir else load_track(args.track)
    sim = Simulator(track, roster, n_envs=args.envs, n_rays=args.rays, lap_target=args.lap_target, lidar_mode=args.lidar)

    def standings():
        for place, i in enumerate(sim.ranking(), 1):
            vs = sim.vehicle_states[i]
            fin = f" finished {ordinal(sim.winners[vs.id])}" if vs.finished else ""
            print(f"{ordinal(place):>5}  {vs.label:<24} laps {vs.laps:3d}  completion {vs.lap_completion():4d} %  "
                  f"lap times {[round(t, 3) for t in vs.times[-3:]]}{fin}")

    try:
        if args.device_policy:
            sim.rollout(args.device_policy, args.steps)
        else:
            for k in range(args.steps):
                sim.step()
                if args.report and (k + 1) % args.report == 0:
                    print(f"-- step {k + 1}")
                    standings()
        print(f"-- after {sim.steps} steps ({sim.steps * sim.timestep:.3f} s of simulated time)")
        standings()
    finally:
        sim.close()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
