"""Track asset front-end: PNG -> wall bitmap + chunk grid, SVG -> 100-point centre-line.

Host-side (numpy + pillow only) restatement of the reference's offline asset
pipeline, producing the packed "track blob" consumed by ``ftgp_create``
(include/ftgp.h).  Nothing here runs per step.

Reference behaviour followed (files under the reference repo):
  * wall threshold: a pixel is wall iff R+G+B == 765 after ``convert("RGB")``
    -- ft_grandprix/chunk.py:39-43
  * chunk grid / ``metadata.json`` fields and the "kept iff any white" rule,
    i-major order -- ft_grandprix/chunk.py:45-79
  * centre-line: first ``<g>/<path>`` ``d`` attribute, group transform ignored,
    100 samples at t = i/100 by arc-length-proportional segment lookup
    -- ft_grandprix/curve.py:6-18 (svg.path 6.3 semantics, restated; that
    package is not installed here, so the samples are "parity unpinned")
  * path world mapping x = px/W*20*scale, y = -py/H*20*scale
    -- ft_grandprix/custom.py:1184-1186
  * wall world mapping: chunk (cx, cy) is a tile centred at
    (size_x*cx, -size_y*cy) with half extents size/2, size = 20*scale/n_chunks
    -- template/mushr.em.xml:17-20,55,92
"""
from __future__ import annotations

import dataclasses
import json
import math
import os
import re
import xml.etree.ElementTree as ET
from bisect import bisect
from typing import List, Optional, Sequence, Tuple

import numpy as np

CHUNK_PX = 20            # chunk.py:13-14 defaults, used by custom.py:1155
MAP_SCALE = 2.0          # custom.py:1155 (scale=2.0)
MAP_EXTENT = CHUNK_PX * MAP_SCALE   # 40 world units, mushr.em.xml:17
PATH_POINTS = 100        # curve.py:8

# ----------------------------------------------------------------------------
# PNG -> wall mask, chunk metadata
# ----------------------------------------------------------------------------

def threshold_image(image) -> np.ndarray:
    """bool[H, W] wall mask; wall iff pure white (chunk.py:39-43)."""
    rgb = np.asarray(image.convert("RGB")).astype(np.int32)
    return rgb.sum(axis=2) == 255 * 3


def chunk_metadata(wall: np.ndarray, name: str, scale: float = MAP_SCALE,
                   chunk_width: int = CHUNK_PX, chunk_height: int = CHUNK_PX) -> dict:
    """The dictionary chunk.py writes to ``rendered/chunks/metadata.json``.

    Same keys, same ``chunks`` order (outer loop over columns i, inner over
    rows j) and the same keep rule (tile has any white pixel) as
    chunk.py:45-79.
    """
    height, width = wall.shape
    hc = math.ceil(width / chunk_width)
    vc = math.ceil(height / chunk_height)
    pad = np.zeros((vc * chunk_height, hc * chunk_width), dtype=bool)
    pad[:height, :width] = wall
    occ = pad.reshape(vc, chunk_height, hc, chunk_width).any(axis=(1, 3))  # [vc, hc]
    chunks = [[i, j] for i in range(hc) for j in range(vc) if occ[j, i]]
    return {
        "original_width": width,
        "original_height": height,
        "chunk_width": chunk_width,
        "chunk_height": chunk_height,
        "horizontal_chunks": hc,
        "vertical_chunks": vc,
        "chunks": chunks,
        "width": width,
        "height": height,
        "name": name,
        "scale": scale,
    }


# ----------------------------------------------------------------------------
# SVG path sampling (svg.path 6.3 semantics restated; see module docstring)
# ----------------------------------------------------------------------------

_LEN_ERROR = 1e-12   # svg.path ERROR
_LEN_MIN_DEPTH = 5   # svg.path MIN_DEPTH


class _Move:
    def __init__(self, to):
        self.start = self.end = to

    def point(self, pos):
        return self.start

    def length(self):
        return 0.0


class _Line:
    def __init__(self, start, end):
        self.start, self.end = start, end

    def point(self, pos):
        return self.start + (self.end - self.start) * pos

    def length(self):
        return abs(self.end - self.start)


class _Cubic:
    def __init__(self, start, c1, c2, end):
        self.start, self.c1, self.c2, self.end = start, c1, c2, end

    def point(self, pos):
        return ((1 - pos) ** 3 * self.start
                + 3 * (1 - pos) ** 2 * pos * self.c1
                + 3 * (1 - pos) * pos ** 2 * self.c2
                + pos ** 3 * self.end)

    def length(self):
        return _segment_length(self, 0.0, 1.0, self.point(0), self.point(1), 0)


class _Quad:
    def __init__(self, start, c, end):
        self.start, self.c, self.end = start, c, end

    def point(self, pos):
        return ((1 - pos) ** 2 * self.start + 2 * (1 - pos) * pos * self.c
                + pos ** 2 * self.end)

    def length(self):
        return _segment_length(self, 0.0, 1.0, self.point(0), self.point(1), 0)


class _Arc:
    """Elliptical arc, SVG endpoint parameterisation (svg.path 6.3 ``Arc``): centre / start angle / sweep from the two
    endpoints, radii, x-axis rotation and the large-arc / sweep flags; radii too small for the chord are scaled up."""

    def __init__(self, start, radius, rotation, arc, sweep, end):
        self.start, self.radius, self.rotation, self.arc, self.sweep, self.end = start, radius, rotation, bool(arc), bool(sweep), end
        self._parameterize()

    def _parameterize(self):
        if self.start == self.end:
            return                                  # degenerate: a point
        if self.radius.real == 0 or self.radius.imag == 0:
            return                                  # degenerate: a straight line
        cosr, sinr = math.cos(math.radians(self.rotation)), math.sin(math.radians(self.rotation))
        dx, dy = (self.start.real - self.end.real) / 2, (self.start.imag - self.end.imag) / 2
        x1prim, y1prim = cosr * dx + sinr * dy, -sinr * dx + cosr * dy
        x1prim_sq, y1prim_sq = x1prim * x1prim, y1prim * y1prim
        rx, ry = self.radius.real, self.radius.imag
        rx_sq, ry_sq = rx * rx, ry * ry
        radius_scale = (x1prim_sq / rx_sq) + (y1prim_sq / ry_sq)
        if radius_scale > 1:
            radius_scale = math.sqrt(radius_scale)
            rx *= radius_scale; ry *= radius_scale
            rx_sq, ry_sq = rx * rx, ry * ry
            self.radius_scale = radius_scale
        else:
            self.radius_scale = 1
        t1, t2 = rx_sq * y1prim_sq, ry_sq * x1prim_sq
        c = math.sqrt(abs((rx_sq * ry_sq - t1 - t2) / (t1 + t2)))
        if self.arc == self.sweep:
            c = -c
        cxprim, cyprim = c * rx * y1prim / ry, -c * ry * x1prim / rx
        self.center = complex((cosr * cxprim - sinr * cyprim) + ((self.start.real + self.end.real) / 2),
                              (sinr * cxprim + cosr * cyprim) + ((self.start.imag + self.end.imag) / 2))
        ux, uy = (x1prim - cxprim) / rx, (y1prim - cyprim) / ry
        vx, vy = (-x1prim - cxprim) / rx, (-y1prim - cyprim) / ry
        n = math.sqrt(ux * ux + uy * uy)
        theta = math.degrees(math.acos(max(-1.0, min(1.0, ux / n))))
        if uy < 0:
            theta = -theta
        self.theta = theta % 360
        n = math.sqrt((ux * ux + uy * uy) * (vx * vx + vy * vy))
        d = max(-1.0, min(1.0, (ux * vx + uy * vy) / n))
        delta = math.degrees(math.acos(d))
        if (ux * vy - uy * vx) < 0:
            delta = -delta
        self.delta = delta % 360
        if not self.sweep:
            self.delta -= 360

    def point(self, pos):
        if self.start == self.end:
            return self.start
        if self.radius.real == 0 or self.radius.imag == 0:
            return self.start + (self.end - self.start) * pos
        angle = math.radians(self.theta + (self.delta * pos))
        cosr, sinr = math.cos(math.radians(self.rotation)), math.sin(math.radians(self.rotation))
        radius = self.radius * self.radius_scale
        x = cosr * math.cos(angle) * radius.real - sinr * math.sin(angle) * radius.imag + self.center.real
        y = sinr * math.cos(angle) * radius.real + cosr * math.sin(angle) * radius.imag + self.center.imag
        return complex(x, y)

    def length(self):
        if self.start == self.end:
            return 0.0
        if self.radius.real == 0 or self.radius.imag == 0:
            return abs(self.end - self.start)
        if self.radius.real == self.radius.imag:   # circular: radius x swept angle
            return abs((self.radius.real * self.radius_scale) * self.delta * math.pi / 180)
        return _segment_length(self, 0.0, 1.0, self.point(0), self.point(1), 0)


def _segment_length(curve, start, end, start_point, end_point, depth):
    """Recursive chord subdivision until two half-chords add < 1e-12 to one chord."""
    mid = (start + end) / 2
    mid_point = curve.point(mid)
    length = abs(end_point - start_point)
    length2 = abs(mid_point - start_point) + abs(end_point - mid_point)
    if (length2 - length > _LEN_ERROR) or (depth < _LEN_MIN_DEPTH):
        depth += 1
        return (_segment_length(curve, start, mid, start_point, mid_point, depth)
                + _segment_length(curve, mid, end, mid_point, end_point, depth))
    return length2


_CMD_RE = re.compile(r"([MmZzLlHhVvCcSsQqTtAa])")
_NUM_RE = re.compile(r"[-+]?(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?")


def parse_svg_path(d: str) -> list:
    """``d`` attribute -> list of segments (Move / Line / Cubic / Quad / Arc / closing Line)."""
    tokens = _CMD_RE.split(d)
    segments: list = []
    cur = 0j
    start = 0j
    last_cmd = None
    last_ctrl: Optional[complex] = None
    i = 1
    while i < len(tokens):
        cmd = tokens[i]
        nums = [float(x) for x in _NUM_RE.findall(tokens[i + 1])] if i + 1 < len(tokens) else []
        i += 2
        rel = cmd.islower()
        c = cmd.upper()
        if c == "Z":
            segments.append(_Line(cur, start))
            cur = start
            last_cmd, last_ctrl = c, None
            continue
        k = 0
        first = True
        while k < len(nums):
            if c == "M":
                p = complex(nums[k], nums[k + 1]); k += 2
                if first:
                    cur = cur + p if (rel and segments) else p
                    start = cur
                    segments.append(_Move(cur))
                else:  # implicit lineto
                    p = cur + p if rel else p
                    segments.append(_Line(cur, p)); cur = p
                last_ctrl = None
            elif c == "L":
                p = complex(nums[k], nums[k + 1]); k += 2
                p = cur + p if rel else p
                segments.append(_Line(cur, p)); cur = p; last_ctrl = None
            elif c == "H":
                x = nums[k]; k += 1
                p = complex(cur.real + x if rel else x, cur.imag)
                segments.append(_Line(cur, p)); cur = p; last_ctrl = None
            elif c == "V":
                y = nums[k]; k += 1
                p = complex(cur.real, cur.imag + y if rel else y)
                segments.append(_Line(cur, p)); cur = p; last_ctrl = None
            elif c == "C":
                c1 = complex(nums[k], nums[k + 1]); c2 = complex(nums[k + 2], nums[k + 3])
                e = complex(nums[k + 4], nums[k + 5]); k += 6
                if rel:
                    c1, c2, e = cur + c1, cur + c2, cur + e
                segments.append(_Cubic(cur, c1, c2, e)); cur = e; last_ctrl = c2
            elif c == "S":
                c2 = complex(nums[k], nums[k + 1]); e = complex(nums[k + 2], nums[k + 3]); k += 4
                if rel:
                    c2, e = cur + c2, cur + e
                c1 = cur + cur - last_ctrl if (last_cmd in "CS" and last_ctrl is not None) else cur
                segments.append(_Cubic(cur, c1, c2, e)); cur = e; last_ctrl = c2
            elif c == "Q":
                c1 = complex(nums[k], nums[k + 1]); e = complex(nums[k + 2], nums[k + 3]); k += 4
                if rel:
                    c1, e = cur + c1, cur + e
                segments.append(_Quad(cur, c1, e)); cur = e; last_ctrl = c1
            elif c == "A":
                radius = complex(nums[k], nums[k + 1]); rotation = nums[k + 2]
                large, sweep = nums[k + 3], nums[k + 4]
                e = complex(nums[k + 5], nums[k + 6]); k += 7
                if rel:
                    e = cur + e
                segments.append(_Arc(cur, radius, rotation, large, sweep, e)); cur = e; last_ctrl = None
            elif c == "T":
                e = complex(nums[k], nums[k + 1]); k += 2
                if rel:
                    e = cur + e
                c1 = cur + cur - last_ctrl if (last_cmd in "QT" and last_ctrl is not None) else cur
                segments.append(_Quad(cur, c1, e)); cur = e; last_ctrl = c1
            first = False
            last_cmd = c
        last_cmd = c
    return segments


def sample_path_points(segments: Sequence, points: int = PATH_POINTS) -> np.ndarray:
    """``[Path.point(i/points) for i in range(points)]`` -> float64[points, 2]."""
    lengths = [s.length() for s in segments]
    total = sum(lengths)
    fractions: List[float] = []
    acc = 0.0
    for each in lengths:
        acc += (each / total) if total != 0 else each
        fractions.append(acc)
    out = np.empty((points, 2), dtype=np.float64)
    for n in range(points):
        pos = n / points
        if pos == 0.0:
            z = segments[0].point(pos)
        elif total == 0:
            z = segments[0].point(0.0)
        else:
            i = min(bisect(fractions, pos), len(segments) - 1)
            if i == 0:
                seg_pos = pos / fractions[0]
            else:
                seg_pos = (pos - fractions[i - 1]) / (fractions[i] - fractions[i - 1])
            z = segments[i].point(seg_pos)
        out[n, 0] = z.real
        out[n, 1] = z.imag
    return out


def extract_path_from_svg(path: str, points: int = PATH_POINTS) -> np.ndarray:
    """Same contract as the reference's ``curve.extract_path_from_svg`` (curve.py:6-18)."""
    root = ET.parse(path).getroot()
    m = re.match(r"\{(.+)\}", root.tag)
    ns = "{" + m.group(1) + "}" if m else ""
    d = root.find(f"{ns}g").find(f"{ns}path").attrib["d"]
    return sample_path_points(parse_svg_path(d), points)


# ----------------------------------------------------------------------------
# Track object and blob
# ----------------------------------------------------------------------------

@dataclasses.dataclass
class Track:
    """Everything ``ftgp_create`` needs to know about one track."""
    name: str
    width: int                    # pixels
    height: int
    bits: np.ndarray              # uint32[height, words_per_row]; bit (x & 31) of word x >> 5
    path: np.ndarray              # float64[100, 2], world frame (custom.py:1185-1186)
    hc: int                       # horizontal_chunks
    vc: int                       # vertical_chunks
    px_size_x: float              # world units per pixel, wall frame
    px_size_y: float
    origin_x: float               # world x of the left edge of pixel column 0
    origin_y: float               # world y of the top edge of pixel row 0
    chunks: Optional[list] = None  # metadata["chunks"]

    @property
    def words_per_row(self) -> int:
        return int(self.bits.shape[1])

    def wall_mask(self) -> np.ndarray:
        """bool[H, W] (unpacked)."""
        b = np.unpackbits(self.bits.view(np.uint8), axis=1, bitorder="little")
        return b[:, : self.width].astype(bool)

    def chunk_mask(self) -> np.ndarray:
        m = np.zeros((self.vc, self.hc), dtype=np.uint8)
        for i, j in (self.chunks or []):
            m[j, i] = 1
        return m

    # -- persistence (derived data only: bitmap + centre-line, no reference files) --
    def save_npz(self, path: str) -> None:
        np.savez_compressed(
            path, name=np.array(self.name), width=self.width, height=self.height,
            bits=self.bits, path=self.path, hc=self.hc, vc=self.vc,
            px_size=np.array([self.px_size_x, self.px_size_y]),
            origin=np.array([self.origin_x, self.origin_y]),
            chunks=np.array(self.chunks or [], dtype=np.int32).reshape(-1, 2))

    @staticmethod
    def load_npz(path: str) -> "Track":
        z = np.load(path, allow_pickle=False)
        return Track(
            name=str(z["name"]), width=int(z["width"]), height=int(z["height"]),
            bits=np.ascontiguousarray(z["bits"], dtype=np.uint32),
            path=np.ascontiguousarray(z["path"], dtype=np.float64),
            hc=int(z["hc"]), vc=int(z["vc"]),
            px_size_x=float(z["px_size"][0]), px_size_y=float(z["px_size"][1]),
            origin_x=float(z["origin"][0]), origin_y=float(z["origin"][1]),
            chunks=[list(map(int, c)) for c in z["chunks"]])


def pack_bits(wall: np.ndarray) -> np.ndarray:
    """bool[H, W] -> uint32[H, ceil(W/32)] little-endian bit order."""
    h, w = wall.shape
    wpr = (w + 31) // 32
    pad = np.zeros((h, wpr * 32), dtype=np.uint8)
    pad[:, :w] = wall
    return np.ascontiguousarray(np.packbits(pad, axis=1, bitorder="little")).view(np.uint32).reshape(h, wpr)


def wall_frame(width: int, height: int, frame: str = "mjcf") -> Tuple[float, float, float, float, int, int]:
    """(px_size_x, px_size_y, origin_x, origin_y, hc, vc) of the wall bitmap in the world.

    ``"mjcf"`` mirrors template/mushr.em.xml:17-20,55,92: the map is squashed to
    40 x 40 units over hc x vc tiles of 20 px, tile (cx, cy) *centred* at
    (size_x*cx, -size_y*cy) -- i.e. pixel column 0 starts at -size_x/2 and pixel
    row 0 at +size_y/2.  ``"pixel"`` is the frame of the centre-line and of
    ``fakelidar`` (custom.py:1185-1186,1382-1384): x = px/W*40, y = -py/H*40.
    """
    hc = math.ceil(width / CHUNK_PX)
    vc = math.ceil(height / CHUNK_PX)
    if frame == "mjcf":
        size_x = MAP_EXTENT / hc
        size_y = MAP_EXTENT / vc
        return size_x / CHUNK_PX, size_y / CHUNK_PX, -size_x / 2, size_y / 2, hc, vc
    if frame == "pixel":
        return MAP_EXTENT / width, MAP_EXTENT / height, 0.0, 0.0, hc, vc
    raise ValueError(f"unknown wall frame {frame!r}")


def build_track(wall: np.ndarray, path_px: np.ndarray, name: str, frame: str = "mjcf") -> Track:
    """Assemble a Track from a wall mask and centre-line samples in pixel coordinates."""
    height, width = wall.shape
    sx, sy, ox, oy, hc, vc = wall_frame(width, height, frame)
    meta = chunk_metadata(wall, name)
    path = np.empty_like(path_px, dtype=np.float64)
    # custom.py:1185-1186, same operation order (divide, multiply, multiply)
    path[:, 0] = path_px[:, 0] / width * CHUNK_PX * MAP_SCALE
    path[:, 1] = -path_px[:, 1] / height * CHUNK_PX * MAP_SCALE
    return Track(name=name, width=width, height=height, bits=pack_bits(wall), path=path,
                 hc=hc, vc=vc, px_size_x=sx, px_size_y=sy, origin_x=ox, origin_y=oy,
                 chunks=meta["chunks"])


def load_track_from_template(template_dir: str, name: str, frame: str = "mjcf") -> Track:
    """``<template_dir>/<name>.png`` + ``<name>-path.svg`` -> Track (the reference's input layout)."""
    from PIL import Image
    wall = threshold_image(Image.open(os.path.join(template_dir, f"{name}.png")))
    path_px = extract_path_from_svg(os.path.join(template_dir, f"{name}-path.svg"))
    return build_track(wall, path_px, name, frame)


def synthetic_oval(width: int = 1600, height: int = 1600, half_width_px: float = 24.0,
                   wall_px: float = 1.6, name: str = "synthetic-oval", frame: str = "mjcf") -> Track:
    """Procedural closed track (superellipse corridor) for users without PNG/SVG assets."""
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float64)
    cx, cy = width / 2.0, height / 2.0
    a, b = width * 0.36, height * 0.27
    t = np.linspace(0.0, 2 * math.pi, 4000, endpoint=False)
    # arc-length uniform resample of the ellipse centre-line, clockwise in pixel space
    ex, ey = cx + a * np.cos(t), cy + b * np.sin(t)
    seg = np.hypot(np.diff(ex, append=ex[0]), np.diff(ey, append=ey[0]))
    s = np.concatenate([[0.0], np.cumsum(seg)])[:-1]
    total = seg.sum()
    target = np.arange(PATH_POINTS) / PATH_POINTS * total
    px = np.interp(target, s, ex)
    py = np.interp(target, s, ey)
    # distance of every pixel centre to the dense centre-line (coarse grid search, exact enough for a wall band)
    from scipy.spatial import cKDTree
    tree = cKDTree(np.stack([ex, ey], axis=1))
    dist, _ = tree.query(np.stack([xx.ravel() + 0.5, yy.ravel() + 0.5], axis=1))
    dist = dist.reshape(height, width)
    wall = np.abs(dist - half_width_px) <= wall_px
    return build_track(wall, np.stack([px, py], axis=1), name, frame)


def bundled_track_path(name: str) -> str:
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", f"{name}.npz")


def load_track(name: str) -> Track:
    """Load a derived track blob shipped in ``ft_grandprix_amd/assets`` (see tools/make_track_blobs.py)."""
    p = bundled_track_path(name)
    if not os.path.exists(p):
        raise FileNotFoundError(f"no bundled track blob {p}; build one with tools/make_track_blobs.py "
                                f"or ft_grandprix_amd.track.load_track_from_template()")
    return Track.load_npz(p)


# ----------------------------------------------------------------------------
# Command line: the reference's ``python -m ft_grandprix.chunk`` (chunk.py:10-98), plus the track blob
# ----------------------------------------------------------------------------

def write_chunks(image_path: str, output_dir: str = "rendered/chunks", chunk_width: int = CHUNK_PX, chunk_height: int = CHUNK_PX,
                 verbose: bool = True, scale: float = 1, force: bool = False) -> Optional[dict]:
    """The files ``chunk()`` leaves behind (chunk.py:10-79): one ``IIIxJJJ.png`` per tile that holds a wall pixel -- walls white,
    everything else black -- and ``metadata.json``.  Same refusals: a non-empty output directory is only replaced with
    ``force`` and only if it carries a ``metadata.json`` (i.e. looks like ours)."""
    import shutil
    import sys
    from PIL import Image
    parent = os.path.dirname(output_dir)
    if parent and not os.path.isdir(parent):
        os.makedirs(parent)
    if os.path.exists(output_dir):
        existing = os.listdir(output_dir)
        if len(existing) != 0:
            if not force:
                print(f"Refusing to overwrite existing directory `{output_dir}`", file=sys.stderr)
                return None
            if verbose:
                print(f"`{output_dir}` exists but the force option was specified", file=sys.stderr)
            if "metadata.json" not in existing:
                print("Refusing to overwrite directory without a `metadata.json` (we may not have created it)")
                return None
        if verbose:
            print(f"Removing `{output_dir}`")
        shutil.rmtree(output_dir)
    os.mkdir(output_dir)
    wall = threshold_image(Image.open(image_path))
    name = ".".join(os.path.basename(image_path).split(".")[:-1])
    meta = chunk_metadata(wall, name, scale, chunk_width, chunk_height)
    kept = {(i, j) for i, j in meta["chunks"]}
    rgb = np.repeat((wall.astype(np.uint8) * 255)[:, :, None], 3, axis=2)
    for i in range(meta["horizontal_chunks"]):
        for j in range(meta["vertical_chunks"]):
            base = f"{i:03}x{j:03}.png"
            if (i, j) in kept:
                if verbose:
                    print(f"Going to produce non-empty chunk {base}")
                tile = rgb[j * chunk_height:(j + 1) * chunk_height, i * chunk_width:(i + 1) * chunk_width]
                Image.fromarray(np.ascontiguousarray(tile)).save(os.path.join(output_dir, base))
            elif verbose:
                print(f"Not going to produce empty chunk {base}")
    with open(os.path.join(output_dir, "metadata.json"), "w") as f:
        json.dump(meta, f)
    return meta


def main(argv=None) -> int:
    import argparse
    ap = argparse.ArgumentParser(prog="python -m ft_grandprix_amd.track",
                                 description="chunk a track image like ft_grandprix.chunk; with --svg/--blob also build the track blob for ftgp_create")
    ap.add_argument("-i", dest="input", required=True, help="the image file split into chunks")
    ap.add_argument("-o", dest="output", default="rendered/chunks", help="the output directory chunks")
    ap.add_argument("-W", dest="chunk_width", default=CHUNK_PX, type=int, help="the chunk width to use in pixels")
    ap.add_argument("-H", dest="chunk_height", default=CHUNK_PX, type=int, help="the chunk height to use in pixels")
    ap.add_argument("-v", dest="verbose", action="store_true", help="verbose output")
    ap.add_argument("-f", dest="force", action="store_true", help="overwrite any directory")
    ap.add_argument("--svg", help="centre-line SVG (<name>-path.svg): also sample the 100 path points")
    ap.add_argument("--blob", help="write the track blob (.npz) consumed by ft_grandprix_amd.track.Track.load_npz / capi.Env")
    a = ap.parse_args(argv)
    write_chunks(a.input, a.output, a.chunk_width, a.chunk_height, a.verbose, force=a.force)
    if a.blob:
        from PIL import Image
        if not a.svg:
            ap.error("--blob needs --svg")
        name = ".".join(os.path.basename(a.input).split(".")[:-1])
        t = build_track(threshold_image(Image.open(a.input)), extract_path_from_svg(a.svg), name)
        t.save_npz(a.blob)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
