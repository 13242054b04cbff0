"""Test-side helpers.  The only place (besides bench.py's cpu_baseline leg and smoke()) that loads the oracle."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libftgp_oracle.so")


def load_oracle():
    from ft_grandprix_amd.capi import CLib
    src = os.path.join(ORACLE_DIR, "ftgp_oracle.c")
    if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    lib = CLib(ORACLE_SO, "oracle_")
    import ctypes as C
    d = lib.dll
    d.oracle_policy_eval1.restype = C.c_int
    d.oracle_policy_eval1.argtypes = [C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    d.oracle_fakelidar.restype = C.c_int
    d.oracle_fakelidar.argtypes = [C.c_double, C.c_double, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_double, C.c_void_p, C.c_void_p]
    d.oracle_quaternion_to_euler.argtypes = [C.c_double] * 4 + [C.c_void_p]
    d.oracle_euler_to_quaternion.argtypes = [C.c_void_p, C.c_void_p]
    d.oracle_progress_trace.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    d.oracle_set_lidar_mode.argtypes = [C.c_void_p, C.c_int]
    d.oracle_set_threads.argtypes = [C.c_void_p, C.c_int]
    d.oracle_get_field.argtypes = [C.c_void_p, C.c_void_p]
    d.oracle_get_distance_field.argtypes = [C.c_void_p, C.c_void_p]
    return lib


def golden(name):
    return os.path.join(GOLDEN, name)


def oracle_policy(lib, policy, scan_f32, last_steer=0.0):
    import ctypes as C
    s = np.ascontiguousarray(scan_f32, dtype=np.float32)
    ls, sp, st = C.c_double(last_steer), C.c_double(), C.c_double()
    rc = lib.dll.oracle_policy_eval1(policy, s.size, s.ctypes.data, C.byref(ls), C.byref(sp), C.byref(st))
    assert rc == 0
    return sp.value, st.value, ls.value


def write_template_track(template_dir, name="generated", width=640, height=480, half_width_px=30.0):
    """A user-style track in the reference's input layout (<name>.png + <name>-path.svg): the centre-line is an SVG path that
    uses H, V, L/l, A, Q, C and S commands; the walls are the pixels 30 +- 1.2 px away from it (white on black, plus
    off-white pixels that must not count)."""
    from PIL import Image
    from scipy.ndimage import distance_transform_edt
    from ft_grandprix_amd import track as tr
    d = ("M 120,80 H 520 A 60,60 0 0 1 580,140 V 340 Q 580,400 520,400 L 320,400 l -200,0 "
         "C 90,400 60,370 60,340 C 60,300 60,280 60,260 S 60,180 60,140 C 60,110 90,80 120,80 Z")
    seg = tr.parse_svg_path(d)
    pts = tr.sample_path_points(seg, 6000)
    line = np.ones((height, width), dtype=bool)
    line[np.clip(pts[:, 1].round().astype(int), 0, height - 1), np.clip(pts[:, 0].round().astype(int), 0, width - 1)] = False
    dist = distance_transform_edt(line)
    wall = np.abs(dist - half_width_px) <= 1.2
    rgb = np.zeros((height, width, 3), dtype=np.uint8)
    rgb[wall] = 255
    rgb[5:9, 5:30] = (255, 255, 254)                       # almost white: not a wall (chunk.py:41)
    os.makedirs(template_dir, exist_ok=True)
    Image.fromarray(rgb).save(os.path.join(template_dir, f"{name}.png"))
    with open(os.path.join(template_dir, f"{name}-path.svg"), "w") as f:
        f.write(f'<?xml version="1.0"?>\n<svg xmlns="http://www.w3.org/2000/svg" width="{width}" height="{height}">'
                f'<g transform="translate(3,4)"><path d="{d}"/></g></svg>\n')
    return wall


# ---------------------------------------------------------------------------------------------------------------------
# K2 (the rangefinder sweep) against the only LiDAR arithmetic the reference owns: raycast.fakelidar (fixture G2).
def lidar_from_pixel_origins(lib, track, origins_px, yaw_world, n_rays):
    """One env per origin: the car is posed so that its LiDAR centre sits on the pixel position origins_px[k] (wall frame of
    `track`) with heading yaw_world[k]; returns (ranges [n, R] in world units, the vehicle constants)."""
    from ft_grandprix_amd import capi
    n = len(origins_px)
    with capi.Env(lib, track, n_envs=n, n_rays=n_rays) as e:
        v = e.cfg.vehicle
        pose = e.pose()
        cx = track.origin_x + origins_px[:, 0] * track.px_size_x
        cy = track.origin_y - origins_px[:, 1] * track.px_size_y
        c, s = np.cos(yaw_world), np.sin(yaw_world)
        pose[:, 0] = cx - (c * v.lidar_x - s * v.lidar_y)
        pose[:, 1] = cy - (s * v.lidar_x + c * v.lidar_y)
        pose[:, 3], pose[:, 6] = np.cos(yaw_world / 2), np.sin(yaw_world / 2)
        pose[:, 7:] = 0.0
        e.set_pose(pose)
        e.step(1)                                  # sensors are evaluated at the pose the step starts from
        return e.lidar().astype(np.float64), float(v.lidar_ring_radius)


def k2_minus_fakelidar_square_pixels(lib, name, R):
    """Differences (pixels) between K2's range measured from the LiDAR centre and the reference's own fakelidar scans of fixture
    G2, ray by ray, on the bitmap of track `name` taken with SQUARE pixels (fakelidar works in pixel space, so this is the frame
    in which its uniform fan and K2's coincide).  G2 ray k has the image-frame angle yaw_g + pi - 2 pi k / R (custom.py:1387);
    K2 ray j of a car with world yaw psi has -(psi + pi + 2 pi j / R) (mushr.em.xml:112-117, y up): the same ray for psi = -yaw_g."""
    import dataclasses
    from ft_grandprix_amd.track import load_track
    g = np.load(golden("g2_fakelidar.npz"))
    t = load_track(name)
    s = 40.0 / t.width
    sq = dataclasses.replace(t, px_size_x=s, px_size_y=s)
    origins, ang, scan = g[f"{name}_origins"], g[f"{name}_{R}_angles"], g[f"{name}_{R}_scan"]
    yaw_g = ang[:, 0] - np.pi
    rng, r0 = lidar_from_pixel_origins(lib, sq, origins, -yaw_g, R)
    assert (rng > 0).all()                          # every origin is enclosed by walls
    return (rng - r0) / s - scan                    # K2 starts its rays r0 behind the centre


def k2_minus_fakelidar_along_k2_rays(lib, name, R, seed=11):
    """The same comparison in the track's REAL wall frame (non-square pixels on the 2133-px tracks): K2's world-space fan is not
    uniform in pixel space there, so fakelidar -- the library's restatement, itself pinned bit for bit to G2 -- is evaluated
    along K2's own pixel-space directions.  Returns the differences in pixels along each ray."""
    from scipy.ndimage import distance_transform_edt
    from ft_grandprix_amd import capi
    from ft_grandprix_amd.track import load_track
    g = np.load(golden("g2_fakelidar.npz"))
    t = load_track(name)
    origins = g[f"{name}_origins"]
    yaw = np.random.default_rng(seed).uniform(-np.pi, np.pi, len(origins))
    rng, r0 = lidar_from_pixel_origins(lib, t, origins, yaw, R)
    a = yaw[:, None] + np.pi + 2 * np.pi * np.arange(R)[None, :] / R            # world angle of ray j
    du, dv = np.cos(a) / t.px_size_x, -np.sin(a) / t.px_size_y                   # pixels per world unit of range
    scale = np.hypot(du, dv)
    dt = np.ascontiguousarray(distance_transform_edt(~t.wall_mask()), dtype=np.float64)
    scan, _ = capi.fakelidar(lib, dt, origins, du / scale, dv / scale)
    assert (rng > 0).all()
    return (rng - r0) * scale - scan
