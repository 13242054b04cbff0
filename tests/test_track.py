"""Asset front-end (ft_grandprix_amd/track.py): SVG path sampling semantics, bit packing, frames, synthetic track."""
import numpy as np
import pytest

from ft_grandprix_amd import track as tr


def test_pack_bits_roundtrip_and_word_layout():
    rng = np.random.default_rng(0)
    wall = rng.random((37, 70)) < 0.2
    bits = tr.pack_bits(wall)
    assert bits.shape == (37, 3) and bits.dtype == np.uint32
    t = tr.build_track(wall, np.zeros((100, 2)), "x")
    np.testing.assert_array_equal(t.wall_mask(), wall)
    y, x = 5, 41
    assert bool((bits[y, x >> 5] >> (x & 31)) & 1) == bool(wall[y, x])     # bit (x & 31) of word x >> 5, include/ftgp.h


def test_wall_frames():
    # mushr.em.xml:17-20,55,92: tiles centred on (size_x*cx, -size_y*cy) -> pixel column 0 starts at -size_x/2
    sx, sy, ox, oy, hc, vc = tr.wall_frame(1600, 1600, "mjcf")
    assert (hc, vc) == (80, 80) and sx == sy == 0.025 and ox == -0.25 and oy == 0.25
    sx, sy, ox, oy, hc, vc = tr.wall_frame(2133, 1600, "mjcf")
    assert (hc, vc) == (107, 80) and sx == pytest.approx(40 / 2140) and ox == pytest.approx(-20 / 107)
    sx, sy, ox, oy, _, _ = tr.wall_frame(2133, 1600, "pixel")         # custom.py:1185-1186, 1382-1384
    assert sx == 40 / 2133 and sy == 40 / 1600 and ox == oy == 0


def test_svg_path_semantics():
    # Path.point(t): segments weighted by arc length, linear local parameter; Move has length 0; Close is a line
    seg = tr.parse_svg_path("m 10,10 l 30,0 0,10 z")           # lengths 30, 10, then the closing line sqrt(30^2+10^2)
    assert [type(s).__name__ for s in seg] == ["_Move", "_Line", "_Line", "_Line"]
    pts = tr.sample_path_points(seg, 4)
    total = 30 + 10 + np.hypot(30, 10)
    np.testing.assert_allclose(pts[0], [10, 10])
    s1 = 0.25 * total                                          # still on the first line
    np.testing.assert_allclose(pts[1], [10 + s1, 10])
    # relative cubic: control points are relative to the segment start (SVG 'c')
    seg = tr.parse_svg_path("M 0,0 c 0,10 10,10 10,0 C 10,-10 20,-10 20,0")
    assert seg[1].c1 == 10j and seg[1].end == 10 and seg[2].start == 10 and seg[2].end == 20
    p = tr.sample_path_points(seg, 2)
    np.testing.assert_allclose(p[1], [10, 0], atol=1e-9)       # half the arc length = the joint of two mirror-image cubics
    # smooth cubic reflects the previous control point
    seg = tr.parse_svg_path("M 0,0 C 0,5 5,5 5,0 S 10,-5 10,0")
    assert seg[2].c1 == complex(5, -5)


def test_cubic_length_matches_fine_polyline():
    c = tr._Cubic(0j, 30 + 40j, 60 - 40j, 100 + 0j)
    t = np.linspace(0, 1, 200001)
    z = (1 - t) ** 3 * c.start + 3 * (1 - t) ** 2 * t * c.c1 + 3 * (1 - t) * t ** 2 * c.c2 + t ** 3 * c.end
    assert c.length() == pytest.approx(np.abs(np.diff(z)).sum(), rel=1e-9)


def test_bundled_tracks_are_closed_loops_on_the_road():
    from scipy.ndimage import distance_transform_edt
    for name in ("track", "circle", "small-circle", "inkscape"):
        t = tr.load_track(name)
        assert t.path.shape == (100, 2) and len(t.chunks) == tr.chunk_metadata(t.wall_mask(), name)["chunks"].__len__()
        edt = distance_transform_edt(~t.wall_mask())
        px, py = t.path[:, 0] / 40 * t.width, -t.path[:, 1] / 40 * t.height
        assert edt[py.astype(int), px.astype(int)].min() > 8          # centre-line stays clear of the walls
        seg = np.hypot(*(np.roll(t.path, -1, axis=0) - t.path).T)
        assert seg.max() < 1.3                                        # closed: the wrap-around segment is as short as the others


def test_synthetic_oval_runs_through_the_oracle(oracle):
    from ft_grandprix_amd import capi
    t = tr.synthetic_oval(640, 480, half_width_px=20)
    with capi.Env(oracle, t, n_envs=4, n_rays=90, spawn_mode=1) as e:
        e.rollout("nidc", 400)
        r = e.lidar()
        assert (r > 0).mean() > 0.95 and np.abs(e.pose()[:, 7:9]).max() > 0.3


# ---------------------------------------------------------------- elliptical arcs, CLI (fixture G6), bracket (fixture G7)
def test_svg_elliptical_arcs():
    # half circle of radius 5 from (0,0) to (10,0): length pi*5, midpoint at (5, -5) or (5, 5) by the sweep flag
    seg = tr.parse_svg_path("M 0,0 A 5,5 0 0 1 10,0")
    assert type(seg[1]).__name__ == "_Arc" and seg[1].length() == pytest.approx(np.pi * 5)
    m = seg[1].point(0.5)
    np.testing.assert_allclose([m.real, m.imag], [5, -5], atol=1e-12)
    m = tr.parse_svg_path("M 0,0 A 5,5 0 0 0 10,0")[1].point(0.5)
    np.testing.assert_allclose([m.real, m.imag], [5, 5], atol=1e-12)
    # radii too small for the chord are scaled up (SVG implementation notes F.6.6)
    a = tr.parse_svg_path("M 0,0 a 1,1 0 0 1 10,0")[1]
    assert a.radius_scale == pytest.approx(5.0) and a.length() == pytest.approx(np.pi * 5)
    # a rotated ellipse: the arc ends where it should and its length matches a fine polyline
    e = tr.parse_svg_path("M 10,20 A 30,12 25 1 0 40,45")[1]
    t = np.linspace(0, 1, 20001)
    z = np.array([e.point(x) for x in t])
    assert abs(z[0] - (10 + 20j)) < 1e-9 and abs(z[-1] - (40 + 45j)) < 1e-9
    assert e.length() == pytest.approx(np.abs(np.diff(z)).sum(), rel=1e-6)
    # degenerate forms
    assert tr.parse_svg_path("M 0,0 A 0,5 0 0 1 3,4")[1].length() == pytest.approx(5.0)
    assert tr.parse_svg_path("M 1,1 A 5,5 0 0 1 1,1")[1].length() == 0.0


def test_g6_chunk_cli_matches_the_reference(tmp_path):
    """python -m ft_grandprix_amd.track against ft_grandprix.chunk.chunk() (fixture G6: files, metadata.json, tile pixels)."""
    import hashlib, json, os, subprocess, sys
    from PIL import Image
    from tests.helpers import golden, ROOT
    g = json.load(open(golden("g6_chunk_cli.json")))
    out = str(tmp_path / "rendered" / "chunks")
    for case in g["cases"]:
        r = subprocess.run([sys.executable, "-m", "ft_grandprix_amd.track", "-i", golden("g6_input.png"), "-o", out, "-f",
                            "-W", str(case["chunk_width"]), "-H", str(case["chunk_height"])], cwd=ROOT, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert sorted(os.listdir(out)) == case["files"]
        meta = json.load(open(os.path.join(out, "metadata.json")))
        ref = dict(case["metadata"], name="g6_input", scale=meta["scale"])      # the fixture image was called g6.png; scale is a caller argument
        assert meta == ref
        for f, sha in case["tile_sha256"].items():
            px = np.asarray(Image.open(os.path.join(out, f)).convert("RGB"))
            assert hashlib.sha256(px.tobytes()).hexdigest() == sha, f
    # refusal semantics of chunk.py:23-34
    foreign = tmp_path / "foreign"; foreign.mkdir(); (foreign / "keep.txt").write_text("x")
    assert tr.write_chunks(golden("g6_input.png"), str(foreign), verbose=False, force=True) is None
    assert sorted(os.listdir(foreign)) == g["foreign_after_force"]
    assert tr.write_chunks(golden("g6_input.png"), out, verbose=False, force=False) is None
    assert sorted(os.listdir(out)) == g["cases"][-1]["files"]


def test_g7_bracket_matches_the_reference(tmp_path):
    """Hasher and compute_driver_files against the reference's outputs (fixture G7)."""
    import json, os
    from ft_grandprix_amd import bracket
    from tests.helpers import golden
    g = json.load(open(golden("g7_bracket.json")))
    for seed, table in ((10, g["hash_seed10"]), (3, g["hash_seed3"])):
        for s, h in table.items():
            assert bracket.Hasher(seed).hash(s) == h, s
    d = tmp_path / "drivers"; d.mkdir()
    for n in g["files"]:
        (d / n).write_text("# generated\n")
    items = bracket.compute_driver_files(str(d), silent=True, palette=g["palette"])
    written = sorted(f for f in os.listdir(d) if f.endswith(".json"))
    assert written == g["written"]
    for f in written:
        assert json.load(open(d / f)) == g["items"][f]
    assert [i["driver"] for i in items] == [g["items"][f]["driver"] for f in written]
    # the DEFAULT palette (assets/bracket_palette.json) is the table the reference indexes: same files without palette=
    assert bracket.default_palette() == g["palette"]
    out = tmp_path / "defaults"; out.mkdir()
    said = []
    bracket.compute_driver_files(str(d), output_dir=str(out), report=said.append)
    assert sorted(os.listdir(out)) == g["written"] and len(said) >= len(g["written"])
    for f in g["written"]:
        assert json.load(open(out / f)) == g["items"][f]


def test_user_track_from_png_and_svg_through_the_oracle(oracle, tmp_path):
    """f-2 end to end on the CPU: PNG + SVG (H V L A Q C S commands) -> load_track_from_template -> world -> cars drive."""
    from ft_grandprix_amd import capi
    from tests.helpers import write_template_track
    wall = write_template_track(str(tmp_path), "generated")
    t = tr.load_track_from_template(str(tmp_path), "generated")
    np.testing.assert_array_equal(t.wall_mask(), wall)                     # off-white pixels are not walls
    assert t.path.shape == (100, 2) and (t.hc, t.vc) == (32, 24)
    np.testing.assert_allclose(t.path[0], [120 / 640 * 40, -80 / 480 * 40])   # group transform ignored (curve.py:13-16), custom.py:1185-1186
    with capi.Env(oracle, t, n_envs=6, n_rays=90, spawn_mode=1, seed=3) as e:
        e.rollout("nidc", 600)
        assert (e.lidar() > 0).mean() > 0.95 and np.abs(e.pose()[:, 7:9]).max() > 0.3
