"""ctypes binding of the C-ABI in include/ftgp.h (no PyTorch, numpy buffers only).

``load()`` opens the HIP library built by ``__graft_entry__.build()`` and fails loudly
when it is missing -- the product has no CPU fallback.  ``CLib`` is generic over the
symbol prefix so that the test-suite can bind the CPU oracle (same signatures, prefix
``oracle_``) through the same wrapper; nothing in this package loads the oracle.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from .track import Track

ABI_VERSION = 5
PATH_POINTS = 100
MAX_LAP_TIMES = 32
SNAPSHOT_DOUBLES = 10
POSE_DOUBLES = 13
PROGRESS_INTS = 10
METRIC_DOUBLES = 8

POLICY_HOST, POLICY_LOBOTOMY, POLICY_NIDC, POLICY_FAST, POLICY_RANDOM = 0, 1, 2, 3, 4
LIDAR_RANGEFINDER, LIDAR_FAKELIDAR = 0, 1
LIDAR_BY_NAME = {"rangefinder": LIDAR_RANGEFINDER, "fakelidar": LIDAR_FAKELIDAR}
POLICY_PER_CAR = 5
POLICY_BY_NAME = {"host": POLICY_HOST, "lobotomy": POLICY_LOBOTOMY, "nidc": POLICY_NIDC,
                  "fast": POLICY_FAST, "random": POLICY_RANDOM, "per_car": POLICY_PER_CAR}

PROGRESS_FIELDS = ("laps", "completion", "lap_completion", "absolute_completion", "finished",
                   "off_track", "start", "good_start", "delta", "finish_step")
METRIC_FIELDS = ("steps", "n_cars", "sum_laps", "sum_absolute_completion", "n_finished",
                 "n_off_track", "min_lap_time", "max_lap_time")


class FtgpTrack(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("words_per_row", C.c_int32),
                ("reserved0", C.c_int32), ("bits", C.c_void_p),
                ("px_size_x", C.c_double), ("px_size_y", C.c_double),
                ("origin_x", C.c_double), ("origin_y", C.c_double), ("path", C.c_void_p)]


class FtgpVehicle(C.Structure):
    _fields_ = [("mass", C.c_double), ("izz", C.c_double),
                ("wheel_x", C.c_double * 4), ("wheel_y", C.c_double * 4),
                ("wheel_radius", C.c_double), ("wheel_inertia", C.c_double), ("wheel_damping", C.c_double),
                ("throttle_kv", C.c_double), ("throttle_gear", C.c_double), ("throttle_force_limit", C.c_double),
                ("steer_kp", C.c_double), ("steer_damping", C.c_double), ("steer_inertia", C.c_double),
                ("steer_limit", C.c_double),
                ("friction", C.c_double), ("gravity", C.c_double), ("tire_damping", C.c_double),
                ("contact_x", C.c_double * 3), ("contact_radius", C.c_double),
                ("contact_stiffness", C.c_double), ("contact_damping", C.c_double),
                ("lidar_x", C.c_double), ("lidar_y", C.c_double), ("lidar_ring_radius", C.c_double),
                ("body_z", C.c_double),
                ("box_xmin", C.c_double), ("box_xmax", C.c_double), ("box_ymin", C.c_double), ("box_ymax", C.c_double),
                ("softener_radius", C.c_double), ("motor_forward_limit", C.c_double), ("motor_turn_limit", C.c_double),
                ("kind", C.c_int32), ("reserved1", C.c_int32)]


class FtgpConfig(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("n_envs", C.c_int32), ("cars_per_env", C.c_int32),
                ("n_rays", C.c_int32), ("lap_target", C.c_int32), ("device_id", C.c_int32),
                ("spawn_mode", C.c_int32), ("env_base", C.c_int32), ("seed", C.c_uint64),
                ("dt", C.c_double), ("bubble_wrap", C.c_int32), ("naive_flatten", C.c_int32),
                ("lidar_mode", C.c_int32), ("reserved2", C.c_int32), ("map_size", C.c_double), ("fan_dirs", C.c_void_p),
                ("track", FtgpTrack), ("vehicle", FtgpVehicle)]


# every symbol include/ftgp.h declares (checked by tests/test_capi.py)
API_SYMBOLS = (
    "default_vehicle", "tricycle_vehicle", "last_error", "device_count", "create", "destroy", "reset", "set_ctrl", "step",
    "rollout", "set_car_policies", "get_lidar", "get_snapshot", "get_pose", "get_progress", "get_winners", "get_lap_times", "get_ctrl",
    "get_steps", "set_pose", "policy_eval", "eval_progress", "metrics_local", "comm_unique_id", "comm_init", "metrics_allgather",
    "metrics_allgather_begin", "metrics_allgather_end", "get_distance_field",
    "last_kernel_ms", "kernel_name", "fakelidar", "selftest", "build_info", "get_race_steps",
)


class FtgpError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"ftgp error {code}: {msg}")
        self.code = code


def product_library_path() -> str:
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libftgp.so")


class CLib:
    """One loaded shared library exposing the ftgp C-ABI under ``prefix``."""

    def __init__(self, path: str, prefix: str = "ftgp_"):
        if not os.path.exists(path):
            raise FileNotFoundError(
                f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"(the HIP library is required; there is no CPU fallback)")
        self.path, self.prefix = path, prefix
        self.dll = C.CDLL(path)
        self._sig()

    def fn(self, name: str):
        return getattr(self.dll, self.prefix + name)

    def has(self, name: str) -> bool:
        return hasattr(self.dll, self.prefix + name)

    def _sig(self):
        vp, i32, dp = C.c_void_p, C.c_int, C.c_void_p
        sigs = {
            "default_vehicle": (None, [C.POINTER(FtgpVehicle)]),
            "tricycle_vehicle": (None, [C.POINTER(FtgpVehicle)]),
            "last_error": (C.c_char_p, []),
            "create": (i32, [C.POINTER(FtgpConfig), C.POINTER(vp)]),
            "destroy": (i32, [vp]),
            "reset": (i32, [vp, dp]),
            "set_ctrl": (i32, [vp, dp, dp]),
            "step": (i32, [vp, i32]),
            "rollout": (i32, [vp, i32, i32]),
            "set_car_policies": (i32, [vp, dp]),
            "get_lidar": (i32, [vp, dp]),
            "get_snapshot": (i32, [vp, dp]),
            "get_pose": (i32, [vp, dp]),
            "get_progress": (i32, [vp, dp]),
            "get_winners": (i32, [vp, dp]),
            "get_lap_times": (i32, [vp, dp, dp]),
            "get_ctrl": (i32, [vp, dp]),
            "get_race_steps": (i32, [vp, dp]),
            "get_steps": (i32, [vp, dp]),
            "set_pose": (i32, [vp, dp]),
            "policy_eval": (i32, [vp, i32, dp, dp]),
            "eval_progress": (i32, [vp]),
            "metrics_local": (i32, [vp, dp]),
            "device_count": (i32, []),
            "comm_unique_id": (i32, [dp]),
            "comm_init": (i32, [vp, dp, i32, i32]),
            "metrics_allgather": (i32, [vp, dp]),
            "metrics_allgather_begin": (i32, [vp]),
            "metrics_allgather_end": (i32, [vp, dp]),
            "get_distance_field": (i32, [vp, dp]),
            "fakelidar": (i32, [i32, dp, i32, i32, i32, dp, i32, dp, dp, C.c_double, dp, dp]),
            "last_kernel_ms": (i32, [vp, C.POINTER(C.c_float)]),
            "kernel_name": (C.c_char_p, [vp]),
            "selftest": (i32, [i32, C.POINTER(C.c_int64)]),
            "build_info": (C.c_char_p, []),
        }
        for name, (res, args) in sigs.items():
            if name == "fakelidar" and self.prefix != "ftgp_":
                continue        # the oracle's single-origin form is typed by tests/helpers.py
            if self.has(name):
                f = self.fn(name)
                f.restype, f.argtypes = res, args

    def build_info(self) -> dict:
        """What the library says it was built from and with (ftgp_build_info): {"abi", "sources", "diag", ...}."""
        text = self.fn("build_info")().decode()
        return dict(kv.split("=", 1) for kv in text.split() if "=" in kv) | {"diag": text.split("diag=", 1)[1].split(" fair_shift=")[0]}

    def last_error(self) -> str:
        s = self.fn("last_error")()
        return s.decode() if s else ""

    def check(self, code: int):
        if code != 0:
            raise FtgpError(code, self.last_error())

    def default_vehicle(self) -> FtgpVehicle:
        v = FtgpVehicle()
        self.fn("default_vehicle")(C.byref(v))
        return v

    def tricycle_vehicle(self) -> FtgpVehicle:
        """The legacy differential-drive car of template/car.em.xml (pair it with dt = 0.0075)."""
        v = FtgpVehicle()
        self.fn("tricycle_vehicle")(C.byref(v))
        return v


_product: Optional[CLib] = None


def load() -> CLib:
    """The product library (HIP).  Raises if it has not been built."""
    global _product
    if _product is None:
        _product = CLib(product_library_path(), "ftgp_")
    return _product


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Env:
    """A batch of worlds behind one opaque C handle (see include/ftgp.h for the contract of each call)."""

    def __init__(self, lib: CLib, track: Track, n_envs: int = 1, cars_per_env: int = 1, n_rays: int = 90,
                 lap_target: int = 10, dt: float = 0.004, spawn_mode: int = 0, seed: int = 1234,
                 device_id: int = 0, vehicle: Optional[FtgpVehicle] = None, env_base: int = 0,
                 bubble_wrap: bool = False, naive_flatten: bool = False, lidar_mode="rangefinder", map_size: float = 0.0,
                 fan_dirs: Optional[np.ndarray] = None):
        self.lib, self.track = lib, track
        self.n_envs, self.cars_per_env, self.n_rays = int(n_envs), int(cars_per_env), int(n_rays)
        self.n_cars = self.n_envs * self.cars_per_env
        self.dt, self.lap_target = float(dt), int(lap_target)
        cfg = FtgpConfig()
        cfg.abi_version = ABI_VERSION
        cfg.n_envs, cfg.cars_per_env, cfg.n_rays = self.n_envs, self.cars_per_env, self.n_rays
        cfg.lap_target, cfg.device_id, cfg.spawn_mode, cfg.seed, cfg.dt = lap_target, device_id, spawn_mode, seed, dt
        cfg.env_base = env_base
        cfg.bubble_wrap, cfg.naive_flatten = int(bool(bubble_wrap)), int(bool(naive_flatten))
        cfg.lidar_mode = LIDAR_BY_NAME[lidar_mode] if isinstance(lidar_mode, str) else int(lidar_mode)
        cfg.map_size = float(map_size)
        self._fan = None if fan_dirs is None else np.ascontiguousarray(fan_dirs, dtype=np.float64).reshape(self.n_rays, 2)
        cfg.fan_dirs = None if self._fan is None else self._fan.ctypes.data
        self.env_base = int(env_base)
        self._bits = np.ascontiguousarray(track.bits, dtype=np.uint32)
        self._path = np.ascontiguousarray(track.path, dtype=np.float64)
        assert self._path.shape == (PATH_POINTS, 2)
        t = cfg.track
        t.width, t.height, t.words_per_row = track.width, track.height, self._bits.shape[1]
        t.bits, t.path = self._bits.ctypes.data, self._path.ctypes.data
        t.px_size_x, t.px_size_y, t.origin_x, t.origin_y = track.px_size_x, track.px_size_y, track.origin_x, track.origin_y
        cfg.vehicle = vehicle if vehicle is not None else lib.default_vehicle()
        self.cfg = cfg
        self.h = C.c_void_p()
        lib.check(lib.fn("create")(C.byref(cfg), C.byref(self.h)))

    # -- lifecycle
    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.lib.fn("destroy")(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _call(self, name, *args):
        self.lib.check(self.lib.fn(name)(self.h, *args))

    # -- control
    def reset(self, mask: Optional[np.ndarray] = None):
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        if m is not None:
            assert m.shape == (self.n_envs,)
        self._call("reset", _ptr(m))

    def set_ctrl(self, ctrl: np.ndarray, car_mask: Optional[np.ndarray] = None):
        c = np.ascontiguousarray(ctrl, dtype=np.float64).reshape(self.n_cars, 2)
        m = None if car_mask is None else np.ascontiguousarray(car_mask, dtype=np.uint8).reshape(self.n_cars)
        self._call("set_ctrl", _ptr(c), _ptr(m))

    def step(self, n_steps: int = 1):
        self._call("step", int(n_steps))

    def rollout(self, policy, n_steps: int):
        p = POLICY_BY_NAME[policy] if isinstance(policy, str) else int(policy)
        self._call("rollout", p, int(n_steps))

    def set_car_policies(self, policies):
        """The bundled driver of every car slot of an env (names or numbers, one per car of the roster): what
        ``rollout("per_car", n)`` evaluates.  Replaces the per-vehicle Driver() instances of custom.py:1097-1104."""
        p = np.array([POLICY_BY_NAME[x] if isinstance(x, str) else int(x) for x in policies], dtype=np.int32)
        if p.shape != (self.cars_per_env,):
            raise ValueError(f"one policy per car of an env: expected {self.cars_per_env}, got {p.shape}")
        self._call("set_car_policies", _ptr(p))

    # -- read-backs
    def lidar(self) -> np.ndarray:
        out = np.empty((self.n_cars, self.n_rays), dtype=np.float32)
        self._call("get_lidar", _ptr(out))
        return out

    def snapshot(self) -> np.ndarray:
        out = np.empty((self.n_cars, SNAPSHOT_DOUBLES), dtype=np.float64)
        self._call("get_snapshot", _ptr(out))
        return out

    def pose(self) -> np.ndarray:
        out = np.empty((self.n_cars, POSE_DOUBLES), dtype=np.float64)
        self._call("get_pose", _ptr(out))
        return out

    def set_pose(self, pose: np.ndarray):
        p = np.ascontiguousarray(pose, dtype=np.float64).reshape(self.n_cars, POSE_DOUBLES)
        self._call("set_pose", _ptr(p))

    def policy_eval(self, policy, ranges: np.ndarray) -> np.ndarray:
        p = POLICY_BY_NAME[policy] if isinstance(policy, str) else int(policy)
        r = np.ascontiguousarray(ranges, dtype=np.float32).reshape(self.n_cars, self.n_rays)
        out = np.empty((self.n_cars, 2), dtype=np.float64)
        self._call("policy_eval", p, _ptr(r), _ptr(out))
        return out

    def eval_progress(self):
        self._call("eval_progress")

    def progress(self) -> np.ndarray:
        out = np.empty((self.n_cars, PROGRESS_INTS), dtype=np.int32)
        self._call("get_progress", _ptr(out))
        return out

    def winners(self) -> np.ndarray:
        """Place of each car among the finishers of its env (1 = winner, 0 = still racing): Mujoco.winners, custom.py:1367-1369."""
        out = np.empty(self.n_cars, dtype=np.int32)
        self._call("get_winners", _ptr(out))
        return out.reshape(self.n_envs, self.cars_per_env)

    def lap_times(self):
        """(counts, ring): counts[i] = len(VehicleState.times) of car i (the true count), ring[i] = the ring of its newest
        MAX_LAP_TIMES lap times, lap time k in slot k % MAX_LAP_TIMES (see ``lap_time_list``)."""
        counts = np.empty(self.n_cars, dtype=np.int32)
        times = np.empty((self.n_cars, MAX_LAP_TIMES), dtype=np.float64)
        self._call("get_lap_times", _ptr(counts), _ptr(times))
        return counts, times

    def race_steps(self) -> np.ndarray:
        """int64 [n_cars, 2] = (start, finish_step): vehicle_state.start (custom.py:1362) and the step at which `finished` was set (-1 while
        racing), with all 64 bits of self.steps (columns 6 and 9 of ``progress()`` saturate at 2**31 - 1)."""
        out = np.empty((self.n_cars, 2), dtype=np.int64)
        self._call("get_race_steps", _ptr(out))
        return out

    def ctrl(self) -> np.ndarray:
        out = np.empty((self.n_cars, 2), dtype=np.float64)
        self._call("get_ctrl", _ptr(out))
        return out

    def steps(self) -> np.ndarray:
        out = np.empty(self.n_envs, dtype=np.int64)
        self._call("get_steps", _ptr(out))
        return out

    def metrics_local(self) -> np.ndarray:
        out = np.empty(METRIC_DOUBLES, dtype=np.float64)
        self._call("metrics_local", _ptr(out))
        return out

    # -- multi-GPU
    def comm_init(self, unique_id: bytes, rank: int, world_size: int):
        buf = np.frombuffer(unique_id, dtype=np.uint8).copy()
        assert buf.size == 128
        self.world_size = world_size
        self._call("comm_init", _ptr(buf), int(rank), int(world_size))

    def metrics_allgather(self) -> np.ndarray:
        out = np.empty((getattr(self, "world_size", 1), METRIC_DOUBLES), dtype=np.float64)
        self._call("metrics_allgather", _ptr(out))
        return out

    def metrics_allgather_begin(self):
        """Enqueue the exchange of the latest launch's record on the side stream and return at once."""
        self._call("metrics_allgather_begin")

    def metrics_allgather_end(self) -> np.ndarray:
        """Wait for the exchange begun last (not for any later launch) and return the [world, 8] records."""
        out = np.empty((getattr(self, "world_size", 1), METRIC_DOUBLES), dtype=np.float64)
        self._call("metrics_allgather_end", _ptr(out))
        return out

    def distance_field(self) -> np.ndarray:
        """FAKELIDAR mode: the Euclidean distance transform built at create, float64 [H, W] in pixels (self.dt of custom.py:1152-1153)."""
        out = np.empty((self.track.height, self.track.width), dtype=np.float64)
        self._call("get_distance_field", _ptr(out))
        return out

    def last_kernel_ms(self) -> float:
        ms = C.c_float()
        self._call("last_kernel_ms", C.byref(ms))
        return float(ms.value)

    def kernel_name(self) -> str:
        s = self.lib.fn("kernel_name")(self.h)
        return s.decode() if s else ""


def lap_time_list(count: int, ring: np.ndarray) -> list:
    """The tail of VehicleState.times (custom.py:124) that the ring still holds, oldest first: all ``count`` lap times while
    count <= MAX_LAP_TIMES, the newest MAX_LAP_TIMES after that."""
    count = int(count)
    first = max(0, count - MAX_LAP_TIMES)
    # (a slot that reads NaN holds no entry: a backward crossing popped the lap time that had overwritten it, see include/ftgp.h)
    return [float(ring[k % MAX_LAP_TIMES]) for k in range(first, count) if ring[k % MAX_LAP_TIMES] == ring[k % MAX_LAP_TIMES]]


def fakelidar(lib: CLib, dt: np.ndarray, origins: np.ndarray, cosines: np.ndarray, sines: np.ndarray, eps: float = 2.0,
              device_id: int = 0):
    """Batched ``ft_grandprix.raycast.fakelidar``: origins [n, 2] px, cosines / sines [n, R] -> (scan [n, R], points [n, R, 2])."""
    dt = np.ascontiguousarray(dt, dtype=np.float64)
    o = np.ascontiguousarray(origins, dtype=np.float64).reshape(-1, 2)
    c = np.ascontiguousarray(cosines, dtype=np.float64).reshape(len(o), -1)
    s = np.ascontiguousarray(sines, dtype=np.float64).reshape(len(o), -1)
    scan = np.empty_like(c)
    pts = np.empty(c.shape + (2,), dtype=np.float64)
    if lib.prefix == "ftgp_":
        lib.check(lib.fn("fakelidar")(device_id, _ptr(dt), dt.shape[0], dt.shape[1], len(o), _ptr(o), c.shape[1], _ptr(c), _ptr(s),
                                      float(eps), _ptr(scan), _ptr(pts)))
    else:   # the oracle exposes the single-origin form
        for k in range(len(o)):
            lib.check(lib.fn("fakelidar")(float(o[k, 0]), float(o[k, 1]), _ptr(dt), dt.shape[0], dt.shape[1], c.shape[1],
                                          c[k].ctypes.data_as(C.c_void_p), s[k].ctypes.data_as(C.c_void_p), float(eps),
                                          scan[k].ctypes.data_as(C.c_void_p), pts[k].ctypes.data_as(C.c_void_p)))
    return scan, pts


def selftest(lib: CLib, device_id: int = 0) -> int:
    """Mismatches of the device self-test (``ftgp_selftest``); 0 is the only acceptable answer."""
    n = C.c_int64(-1)
    lib.check(lib.fn("selftest")(device_id, C.byref(n)))
    return int(n.value)


def comm_unique_id(lib: CLib) -> bytes:
    buf = np.zeros(128, dtype=np.uint8)
    lib.check(lib.fn("comm_unique_id")(_ptr(buf)))
    return buf.tobytes()
