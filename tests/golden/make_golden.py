#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING the reference's own Python.

Run once in the build container (needs /root/reference; never runs on the GPU box):
    python tests/golden/make_golden.py [/root/reference]

Fixtures (data only -- inputs and the reference's outputs):
  g1_drivers.npz   nidc / fast / lobotomy / drivers.template process_lidar on seeded scans, R in {36, 90, 1080}
  g2_fakelidar.npz ft_grandprix.raycast.fakelidar on each track's EDT from on-track origins, R in {36, 1080}
  g3_chunk.json    ft_grandprix.chunk.chunk() metadata.json for the 4 tracks + SHA-256 of the chunk pixels
  g4_math.npz      custom.quaternion_to_euler / euler_to_quaternion / quaternion_to_angle, VehicleState
                   lap_completion / absolute_completion truth table, ordinal()
  g5_progress.npz  the lap-progress block custom.py:1340-1372 executed verbatim (sliced from the source text
                   and exec'd against stub objects) on hand-built position traces
  g6_chunk_cli.json  ft_grandprix.chunk.chunk() on a generated 130 x 95 image with -W 20 -H 20 and 32 x 24 tiles:
                   metadata.json, the list of files written and the SHA-256 of every tile's decoded RGB pixels
  g7_bracket.json  ft_grandprix.bracket.Hasher(10).hash on sample strings; compute_driver_files() on a generated drivers
                   directory (the palette it used is stored as the function's input)

  g8_fakelidar_step.npz  the reference's 2-D LiDAR as the step loop would call it (custom.py:1381-1393): ft_grandprix.raycast.fakelidar
                   on scipy's distance transform (recipe of custom.py:1149-1153 / raycast.py:24-27) from car poses mapped to pixels with the
                   expressions of custom.py:1382-1384, fan = the rangefinders' (SURVEY.md 8a-3), ranges scaled as custom.py:1392-1393;
                   plus the SHA-256 of each track's distance transform

custom.py imports mujoco / dearpygui / empy / svg.path, which are not installed; they are replaced by
MagicMock entries in sys.modules for the import only -- none of the functions
exercised here touches them.
"""
import hashlib
import json
import os
import sys
import tempfile
import textwrap
import types
from unittest import mock

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(HERE, "..", ".."))

from ft_grandprix import nidc, fast, lobotomy, raycast, chunk as ref_chunk  # noqa: E402
from drivers import template as ref_template  # noqa: E402


# ----------------------------------------------------------------------------- G1
def scans(rng, R, n):
    """Seeded scans, binary32-representable (the device LiDAR is f32), returned as float64 arrays."""
    out = []
    ang = np.linspace(0, 2 * np.pi, R, endpoint=False)
    for k in range(n):
        kind = k % 8
        if kind == 0:
            r = rng.uniform(0.2, 8.0, R)
        elif kind == 1:   # smooth corridor-like profile with a few step disparities
            r = 1.5 + 1.0 * np.cos(ang * rng.integers(1, 4) + rng.uniform(0, 6.28)) + 0.3 * rng.uniform(0, 1, R)
            for _ in range(rng.integers(1, 6)):
                a, b = sorted(rng.integers(0, R, 2))
                r[a:b] += rng.uniform(0.7, 4.0)
        elif kind == 2:   # no-hit markers
            r = rng.uniform(0.2, 8.0, R)
            r[rng.uniform(0, 1, R) < 0.05] = -1.0
        elif kind == 3:   # zeros (first step after reset) and near-zeros
            r = rng.uniform(0.2, 8.0, R) if k % 16 == 3 else np.zeros(R)
            if k % 16 == 3:
                r[rng.integers(0, R, max(1, R // 20))] = 0.0
        elif kind == 4:   # ranges[0] straddling fast's 0.5 threshold, nearly straight-ahead maximum
            r = 2.0 + 0.5 * np.cos(ang - np.pi) + 0.01 * rng.uniform(0, 1, R)
            r[R // 2 + rng.integers(-2, 3)] = 9.0
            r[0] = rng.choice([0.49, 0.5, 0.51, 0.2, 3.0])
        elif kind == 5:   # disparities close to the array ends (cover loops hit the bounds)
            r = rng.uniform(2.0, 2.5, R)
            e = R // 8
            r[e + rng.integers(0, 3)] += 3.0
            r[R - e - 1 - rng.integers(0, 3)] += 3.0
            r[e + 5: e + 8] = 0.25
        elif kind == 6:   # differences right at the 0.6 threshold (binary32 neighbours of 0.6)
            r = np.full(R, 1.0)
            idx = rng.integers(R // 8 + 2, R - R // 8 - 2, 6)
            r[idx] = 1.0 + np.array([0.6, 0.6000001, 0.5999999, 0.61, 0.59, 2.0])
        else:             # plateaus: ties for the argmax
            r = np.round(rng.uniform(0.2, 4.0, R), 1)
        out.append(np.float32(r).astype(np.float64))
    return np.stack(out)


def gen_g1():
    rng = np.random.default_rng(20240601)
    data = {}
    for R in (36, 90, 1080):
        S = scans(rng, R, 64)
        res = {k: np.zeros((len(S), 2)) for k in ("nidc", "fast", "lobotomy", "template")}
        d_fast = fast.Driver()  # one instance, called in sequence (keeps last_steering_angle)
        for i, r in enumerate(S):
            with np.errstate(all="ignore"):
                res["nidc"][i] = nidc.Driver().process_lidar(r.copy())
                res["fast"][i] = d_fast.process_lidar(r.copy())
            res["lobotomy"][i] = lobotomy.Driver().process_lidar(r.copy())
            res["template"][i] = ref_template.Driver().process_lidar(r.copy(), None)
        data[f"scans_{R}"] = S.astype(np.float32)
        for k, v in res.items():
            data[f"{k}_{R}"] = v
    np.savez_compressed(os.path.join(HERE, "g1_drivers.npz"), **data)
    print("g1: ok", {k: v.shape for k, v in data.items() if k.startswith("scans")})


# ----------------------------------------------------------------------------- G2
def gen_g2():
    from scipy.ndimage import distance_transform_edt
    from ft_grandprix_amd.track import load_track_from_template
    rng = np.random.default_rng(7)
    data = {}
    for name in ("track", "circle", "small-circle", "inkscape"):
        t = load_track_from_template(os.path.join(REF, "template"), name)
        dt = distance_transform_edt(~t.wall_mask())
        # on-track origins: the centre-line samples mapped back to pixels, plus jitter
        px = t.path[:, 0] / 40.0 * t.width
        py = -t.path[:, 1] / 40.0 * t.height
        idx = rng.choice(100, 32, replace=False)
        ox = px[idx] + rng.uniform(-4, 4, 32)
        oy = py[idx] + rng.uniform(-4, 4, 32)
        for R in (36, 1080):
            yaw = rng.uniform(-np.pi, np.pi, 32)
            scan = np.zeros((32, R)); pts = np.zeros((32, R, 2)); angs = np.zeros((32, R))
            for k in range(32):
                a = np.linspace(yaw[k] + np.pi, yaw[k] - np.pi, R, endpoint=False)  # custom.py:1387
                angs[k] = a
                scan[k], pts[k] = raycast.fakelidar(ox[k], oy[k], dt, R, np.cos(a), np.sin(a))
            data[f"{name}_{R}_angles"] = angs
            data[f"{name}_{R}_scan"] = scan
            data[f"{name}_{R}_points"] = pts
        data[f"{name}_origins"] = np.stack([ox, oy], axis=1)
    np.savez_compressed(os.path.join(HERE, "g2_fakelidar.npz"), **data)
    print("g2: ok")


# ----------------------------------------------------------------------------- G3
def gen_g3():
    from PIL import Image
    out = {}
    cwd = os.getcwd()
    for name in ("track", "circle", "small-circle", "inkscape"):
        with tempfile.TemporaryDirectory() as td:
            os.chdir(td)
            try:
                ref_chunk.chunk(os.path.join(REF, "template", f"{name}.png"), verbose=False, force=True, scale=2.0)
                with open("rendered/chunks/metadata.json") as f:
                    meta = json.load(f)
                h = hashlib.sha256()
                white = 0
                for i, j in meta["chunks"]:
                    a = np.asarray(Image.open(f"rendered/chunks/{i:03}x{j:03}.png").convert("RGB"))
                    h.update(a.tobytes())
                    white += int((a.astype(int).sum(2) == 765).sum())
                out[name] = {"metadata": meta, "chunk_pixels_sha256": h.hexdigest(), "white_pixels": white}
            finally:
                os.chdir(cwd)
    with open(os.path.join(HERE, "g3_chunk.json"), "w") as f:
        json.dump(out, f)
    print("g3: ok", {k: (len(v["metadata"]["chunks"]), v["white_pixels"]) for k, v in out.items()})


# ----------------------------------------------------------------------------- G4 / G5 need custom.py
def import_custom():
    for m in ("mujoco", "mujoco.viewer", "dearpygui", "dearpygui.dearpygui", "em", "svg", "svg.path",
              "OpenGL", "OpenGL.GL", "glfw"):
        sys.modules.setdefault(m, mock.MagicMock())
    import importlib
    return importlib.import_module("ft_grandprix.custom")


def gen_g4(custom):
    rng = np.random.default_rng(11)
    q = rng.normal(size=(256, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    # planar (yaw-only) quaternions and gimbal-lock clamps
    yaw = rng.uniform(-np.pi, np.pi, 64)
    q[:64] = np.stack([np.cos(yaw / 2), np.zeros(64), np.zeros(64), np.sin(yaw / 2)], axis=1)
    q[64] = [np.sqrt(0.5), 0, np.sqrt(0.5), 0]; q[65] = [np.sqrt(0.5), 0, -np.sqrt(0.5), 0]
    eul = np.array([custom.quaternion_to_euler(*row) for row in q])
    ang = np.array([custom.quaternion_to_angle(*row) for row in q])
    e_in = rng.uniform(-np.pi, np.pi, size=(256, 3)); e_in[:64, 1:] = 0.0
    quat = np.array([custom.euler_to_quaternion(list(row)) for row in e_in], dtype=np.float64)
    # VehicleState accessors: truth table over (completion, laps, good_start)
    rows = []
    vs = custom.VehicleState.__new__(custom.VehicleState)
    for laps in (-2, -1, 0, 1, 9):
        for completion in (0, 1, 37, 50, 99):
            for good in (True, False):
                vs.laps, vs.completion, vs.good_start = laps, completion, good
                rows.append([laps, completion, int(good), vs.lap_completion(), vs.absolute_completion()])
    ords = [custom.ordinal(n) for n in range(0, 125)]
    np.savez_compressed(os.path.join(HERE, "g4_math.npz"), quats=q, eulers=eul, angles=ang,
                        euler_in=e_in, quat_out=quat, vehicle_state=np.array(rows, dtype=np.int64),
                        ordinals=np.array(ords))
    print("g4: ok")


def progress_block_source():
    """The text of custom.py's per-car progress block (custom.py:1340-1372), located by its first/last statements."""
    lines = open(os.path.join(REF, "ft_grandprix", "custom.py")).read().split("\n")
    a = next(i for i, l in enumerate(lines) if "xpos = vehicle_state.joint.qpos[0:2]" in l)
    b = next(i for i, l in enumerate(lines) if "vehicle_state.completion = completion" in l and i > a)
    return textwrap.dedent("\n".join(lines[a:b + 1]))


def gen_g5():
    src = progress_block_source()
    code = compile(src, "custom.py:progress-block", "exec")
    rng = np.random.default_rng(5)
    from ft_grandprix_amd.track import load_track_from_template
    t = load_track_from_template(os.path.join(REF, "template"), "track")
    path = t.path
    dt = 0.004

    def run(xy, offset, lap_target):
        vs = types.SimpleNamespace(id=0, offset=offset, completion=0, good_start=True, finished=False, delta=0,
                                   distance_from_track=0.0, start=0, laps=0, times=[], off_track=False,
                                   joint=types.SimpleNamespace(qpos=np.zeros(7)))
        shadowed = []
        slf = types.SimpleNamespace(path=path, steps=0, winners={}, shadow=lambda i: shadowed.append(i),
                                    option=lambda k: {"lap_target": lap_target}[k],
                                    model=types.SimpleNamespace(opt=types.SimpleNamespace(timestep=dt)))
        out = np.zeros((len(xy), 7), dtype=np.int64); d2 = np.zeros(len(xy)); tt = []
        for s, p in enumerate(xy):
            slf.steps = s
            vs.joint.qpos[0:2] = p
            exec(code, {"np": np, "abs": abs, "len": len}, {"self": slf, "vehicle_state": vs})
            out[s] = [vs.laps, vs.completion, int(vs.good_start), len(vs.times), int(vs.finished), vs.delta, int(vs.off_track)]
            d2[s] = vs.distance_from_track
            tt.append(list(vs.times))
        return out, d2, tt[-1]

    def along(indices, jitter=0.05):
        p = path[np.asarray(indices) % 100]
        return p + rng.uniform(-jitter, jitter, p.shape)

    traces = {
        # forward: two and a half laps from the spawn offset, several steps per point
        "forward": (np.repeat(np.arange(10, 10 + 260), 3), 10, 2),
        # backward from the start (crosses 0 -> 99 immediately: reverse start), then forward again past the line twice
        "reverse_start": (np.concatenate([np.arange(10, -15, -1), np.arange(-15, 130)]), 10, 10),
        # forward one lap, back across the line, forward again (pop / good_start handling)
        "back_and_forth": (np.concatenate([np.arange(12, 115), np.arange(115, 105, -1), np.arange(105, 230),
                                           np.arange(230, 205, -1), np.arange(205, 320)]), 12, 10),
        # coarse jumps (fast car): 7 points per step
        "fast_jumps": (np.arange(14, 14 + 7 * 120, 7), 14, 5),
    }
    data = {}
    for name, (idx, offset, target) in traces.items():
        xy = along(idx)
        if name == "back_and_forth":   # off-track excursion in the middle: progress frozen while d^2 > 1
            xy[60:75] += np.array([3.0, 3.0])
        out, d2, times = run(xy, offset, target)
        data[f"{name}_xy"] = xy
        data[f"{name}_out"] = out
        data[f"{name}_d2"] = d2
        data[f"{name}_times"] = np.array(times, dtype=np.float64)
        data[f"{name}_params"] = np.array([offset, target], dtype=np.int64)
        print("g5:", name, "final laps/completion/good/ntimes/finished/delta/off =", out[-1], "times", np.round(times, 3))
    data["path"] = path
    data["dt"] = np.array(dt)
    np.savez_compressed(os.path.join(HERE, "g5_progress.npz"), **data)


# ----------------------------------------------------------------------------- G6
def g6_image():
    """Deterministic test image: white curves and blobs on black, plus off-white and coloured pixels that must NOT count as wall."""
    rng = np.random.default_rng(6)
    h, w = 95, 130
    img = np.zeros((h, w, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    ring = np.abs(np.hypot(xx - 60, yy - 45) - 30) < 1.2
    img[ring] = 255
    img[10:14, 100:128] = 255
    img[rng.random((h, w)) < 0.01] = (255, 255, 254)          # almost white: not a wall (chunk.py:41)
    img[rng.random((h, w)) < 0.01] = (255, 0, 0)
    img[80:95, 0:3] = 255                                       # touches the ragged last row of tiles
    return img


def gen_g6():
    from PIL import Image
    out = {"cases": []}
    with tempfile.TemporaryDirectory() as td:
        cwd = os.getcwd(); os.chdir(td)
        try:
            Image.fromarray(g6_image()).save("g6.png")
            Image.fromarray(g6_image()).save(os.path.join(HERE, "g6_input.png"))      # the input travels with the fixture
            for cw, chh in ((20, 20), (32, 24)):
                ref_chunk.chunk("g6.png", output_dir="rendered/chunks", chunk_width=cw, chunk_height=chh, verbose=False, scale=2.0, force=True)
                meta = json.load(open("rendered/chunks/metadata.json"))
                files = sorted(os.listdir("rendered/chunks"))
                sha = {f: hashlib.sha256(np.asarray(Image.open(os.path.join("rendered/chunks", f)).convert("RGB")).tobytes()).hexdigest()
                       for f in files if f.endswith(".png")}
                out["cases"].append({"chunk_width": cw, "chunk_height": chh, "metadata": meta, "files": files, "tile_sha256": sha})
            # refusal semantics (chunk.py:23-34): a foreign non-empty directory is never replaced
            os.makedirs("foreign"); open("foreign/keep.txt", "w").write("x")
            ref_chunk.chunk("g6.png", output_dir="foreign", verbose=False, force=True)
            out["foreign_after_force"] = sorted(os.listdir("foreign"))
            ref_chunk.chunk("g6.png", output_dir="rendered/chunks", verbose=False, force=False)
            out["existing_after_noforce"] = sorted(os.listdir("rendered/chunks")) == out["cases"][-1]["files"]
        finally:
            os.chdir(cwd)
    json.dump(out, open(os.path.join(HERE, "g6_chunk_cli.json"), "w"))
    print("g6:", [(c["chunk_width"], len(c["files"])) for c in out["cases"]], out["foreign_after_force"], out["existing_after_noforce"])


# ----------------------------------------------------------------------------- G7
def gen_g7():
    from ft_grandprix import bracket as ref_bracket
    from ft_grandprix.colors import colors as ref_colors
    strings = ["", "a", "drivers.template", "drivers.template.", "ft_grandprix.nidc", "x" * 40, "drivers.Zeta9", "drivers.my_driver"]
    hashes = {s: ref_bracket.Hasher(10).hash(s) for s in strings}
    hashes_seed3 = {s: ref_bracket.Hasher(3).hash(s) for s in strings}
    palette = [a[1] for a in sorted(list(ref_colors.items()), key=lambda t: t[0])]      # what compute_driver_files indexes into
    names = ["alpha.py", "beta_driver.py", "__init__.py", "notes.txt", "gamma.py", "zz_top.pyc"]
    with tempfile.TemporaryDirectory() as td:
        cwd = os.getcwd(); os.chdir(td)
        try:
            os.makedirs("drivers")
            for n in names:
                open(os.path.join("drivers", n), "w").write("# generated\n")
            ref_bracket.compute_driver_files("drivers", silent=True)
            written = sorted(f for f in os.listdir("drivers") if f.endswith(".json"))
            items = {f: json.load(open(os.path.join("drivers", f))) for f in written}
        finally:
            os.chdir(cwd)
    json.dump({"hash_seed10": hashes, "hash_seed3": hashes_seed3, "palette": palette, "files": names, "written": written, "items": items},
              open(os.path.join(HERE, "g7_bracket.json"), "w"))
    print("g7:", hashes, written)



# ----------------------------------------------------------------------------- G8
def gen_g8():
    """Inputs are prepared with numpy's element-wise binary64 arithmetic (IEEE: one rounding per operation, what FtgpConfig.lidar_mode =
    FTGP_LIDAR_FAKELIDAR specifies); everything from `fakelidar(...)` on is the reference's own code."""
    from scipy.ndimage import distance_transform_edt
    from ft_grandprix_amd.track import load_track_from_template
    rng = np.random.default_rng(8)
    data = {}
    s = 20 * 2.0                                                    # s = 20 * map_metadata["scale"], custom.py:1155,1382
    for name in ("track", "circle", "small-circle", "inkscape"):
        t = load_track_from_template(os.path.join(REF, "template"), name)
        W, H = t.width, t.height
        balls = ~t.wall_mask()                                      # custom.py:1149-1151: 0 on pure-white pixels, non-zero elsewhere
        dt = distance_transform_edt(balls)                          # custom.py:1152-1153, raycast.py:27
        data[f"{name}_edt_sha256"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(dt, dtype=np.float64).tobytes()).digest(), dtype=np.uint8)
        n = 16
        idx = rng.choice(100, n, replace=False)
        xy = t.path[idx] + rng.uniform(-0.1, 0.1, (n, 2))
        yaw = rng.uniform(-np.pi, np.pi, n)
        qw0, qz0 = np.cos(yaw / 2), np.sin(yaw / 2)
        nrm = np.sqrt(qw0 * qw0 + qz0 * qz0)                        # ftgp_set_pose normalises the quaternion
        qw, qz = qw0 / nrm, qz0 / nrm
        data[f"{name}_xy"], data[f"{name}_quat"] = xy, np.stack([qw0, qz0], axis=1)
        ch, sh = 1.0 - 2.0 * (qz * qz), 2.0 * (qw * qz)
        i_x = (xy[:, 0] / s) * W                                    # custom.py:1383
        i_y = -(xy[:, 1] / s) * H                                   # custom.py:1384
        for R in (36, 1080):
            phi = ((360.0 / R) * np.arange(R) - 90.0) * (np.pi / 180.0)     # mushr.em.xml:112-117
            fan = np.stack([np.sin(phi), -np.cos(phi)], axis=1)
            data[f"fan_{R}"] = fan
            scan = np.zeros((n, R)); rngs = np.zeros((n, R), dtype=np.float32)
            for k in range(n):
                dxw = ch[k] * fan[:, 0] - sh[k] * fan[:, 1]
                dyw = sh[k] * fan[:, 0] + ch[k] * fan[:, 1]
                ranges, _ = raycast.fakelidar(i_x[k], i_y[k], dt, R, dxw, -dyw)
                scan[k] = ranges
                ranges = ranges.copy()
                ranges /= W                                         # custom.py:1392 (original_width)
                ranges *= s                                         # custom.py:1393
                rngs[k] = ranges.astype(np.float32)
            data[f"{name}_{R}_scan"], data[f"{name}_{R}_ranges"] = scan, rngs
    np.savez_compressed(os.path.join(HERE, "g8_fakelidar_step.npz"), **data)
    print("g8: ok")


if __name__ == "__main__":
    if "--only-g8" in sys.argv:
        gen_g8()
        raise SystemExit(0)
    gen_g6()
    gen_g7()
    gen_g8()
    if "--only-new" in sys.argv:
        raise SystemExit(0)
    gen_g1()
    gen_g2()
    gen_g3()
    custom = import_custom()
    gen_g4(custom)
    gen_g5()
