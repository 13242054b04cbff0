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
