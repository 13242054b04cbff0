"""FtgpConfig.lidar_mode = FTGP_LIDAR_FAKELIDAR: the reference's own 2-D LiDAR (raycast.py:5-21, wired at custom.py:1381-1393) as
the K2 of the step loop, so that the closed loop runs on reference-pinned arithmetic except for K1.

Pins (fixtures produced by RUNNING the reference, tests/golden/make_golden.py):
  G8  raycast.fakelidar on scipy's distance transform from car poses mapped to pixels with custom.py:1382-1384, fan = the
      rangefinders', ranges scaled as custom.py:1392-1393 -- equal bit for bit to one ftgp_step from those poses
  G2  the standalone goldens (32 origins x {36, 1080} rays x 4 tracks) -- equal bit for bit THROUGH ftgp_step, with the car
      posed at yaw 0 on the origin and each origin's own fan handed in as FtgpConfig.fan_dirs
  the distance transform itself: built without scipy (exact integer squared distances, one sqrt), compared with
      scipy.ndimage.distance_transform_edt (the reference's recipe, custom.py:1152-1153) here and by SHA-256 in G8

CPU: the oracle; `-m gpu`: libftgp.so through the C-ABI, and closed loops GPU against oracle.
"""
import hashlib

import numpy as np
import pytest

from ft_grandprix_amd import capi
from ft_grandprix_amd.track import load_track
from tests.helpers import golden

TRACKS = ["track", "circle", "small-circle", "inkscape"]
MAP = 40.0            # 20 * scale, custom.py:1155,1382


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a, dtype=np.float64).tobytes()).digest(), dtype=np.uint8)


def step_from_poses(lib, track, xy, quat, R, fan):
    """One env per pose, one ftgp_step: the scan the drivers would be handed (float32 [n, R])."""
    n = len(xy)
    with capi.Env(lib, track, n_envs=n, n_rays=R, lidar_mode="fakelidar", fan_dirs=fan) as e:
        pose = e.pose()
        pose[:, 0:2] = xy
        pose[:, 3], pose[:, 6] = quat[:, 0], quat[:, 1]
        pose[:, 7:] = 0.0
        e.set_pose(pose)
        e.step(1)                       # sensors are evaluated at the pose the step starts from
        return e.lidar()


def check_g8(lib, name):
    g = np.load(golden("g8_fakelidar_step.npz"))
    t = load_track(name)
    for R in (36, 1080):
        got = step_from_poses(lib, t, g[f"{name}_xy"], g[f"{name}_quat"], R, g[f"fan_{R}"])
        np.testing.assert_array_equal(got, g[f"{name}_{R}_ranges"])
        assert (got > 0).all()
        # fan_dirs = NULL: the fan include/ftgp.h documents -- (sin phi_j, -cos phi_j) in binary64 for every site -- is the fixture's fan,
        # so the default-fan path (bench --lidar fakelidar, sim --lidar fakelidar) is pinned by the reference too
        np.testing.assert_array_equal(step_from_poses(lib, t, g[f"{name}_xy"], g[f"{name}_quat"], R, None), g[f"{name}_{R}_ranges"])


def preimage(target, scale, size):
    """A world coordinate x with (x / MAP) * size == target (custom.py:1383-1384), or the nearest miss: the map is a step
    function of x, and not every double has a preimage."""
    x = target / size * scale
    best = x
    for _ in range(4):
        got = (x / scale) * size
        if got == target:
            return x, True
        if abs(got - target) < abs((best / scale) * size - target):
            best = x
        x = np.nextafter(x, np.inf if got < target else -np.inf)
    return best, False


def check_g2_through_step(lib, name, n_origins=8):
    """G2 ray k has the image-frame direction (cos a_k, sin a_k).  With the car at yaw 0 the mode's direction is (bx, -by) of the
    fan entry, so fan = (cos a, -sin a) reproduces it exactly (negation is exact; 1 * b - 0 * b' == b)."""
    g = np.load(golden("g2_fakelidar.npz"))
    t = load_track(name)
    origins = g[f"{name}_origins"][:n_origins]
    exact = 0
    for R in (36, 1080):
        ang, scan = g[f"{name}_{R}_angles"], g[f"{name}_{R}_scan"]
        for k, (ox, oy) in enumerate(origins):
            x, okx = preimage(ox, MAP, t.width)
            y, oky = preimage(-oy, MAP, t.height)          # i_y = -(y / s) * H: (y / s) * H = -oy, and the negation is exact
            exact += okx and oky
            fan = np.stack([np.cos(ang[k]), -np.sin(ang[k])], axis=1)
            got = step_from_poses(lib, t, np.array([[x, y]]), np.array([[1.0, 0.0]]), R, fan)[0]
            want = ((scan[k] / t.width) * MAP).astype(np.float32)      # ranges /= original_width; ranges *= s (custom.py:1392-1393)
            np.testing.assert_array_equal(got, want, err_msg=f"{name} R={R} origin {k}")
    return exact


# ------------------------------------------------------------------------------------------------ CPU: the oracle
@pytest.mark.parametrize("name", TRACKS)
def test_oracle_distance_transform_is_scipys(oracle, name):
    """custom.py:1149-1153 / raycast.py:24-27 restated without scipy: identical doubles, and the hash the fixture holds."""
    from scipy.ndimage import distance_transform_edt
    t = load_track(name)
    with capi.Env(oracle, t, n_envs=1, n_rays=8, lidar_mode="fakelidar") as e:
        dt = np.empty((t.height, t.width))
        assert oracle.dll.oracle_get_distance_field(e.h, dt.ctypes.data) == 0
    ref = distance_transform_edt(~t.wall_mask())
    np.testing.assert_array_equal(dt, ref)
    np.testing.assert_array_equal(sha(dt), np.load(golden("g8_fakelidar_step.npz"))[f"{name}_edt_sha256"])


def test_oracle_distance_transform_on_odd_images(oracle):
    """Ragged shapes, a single wall pixel, walls on the border, a wall-free column set: against scipy."""
    import dataclasses
    from scipy.ndimage import distance_transform_edt
    from ft_grandprix_amd.track import Track
    rng = np.random.default_rng(3)
    base = load_track("small-circle")
    for (H, W, p) in ((1, 1, 1.0), (7, 45, 0.02), (33, 5, 0.1), (61, 127, 0.001), (40, 40, 0.5), (19, 70, 0.0)):
        wall = rng.uniform(size=(H, W)) < p
        if not wall.any():
            wall[rng.integers(H), rng.integers(W)] = True
        bits = np.packbits(np.pad(wall, ((0, 0), (0, (-W) % 32))), axis=1, bitorder="little").view(np.uint32)
        t = dataclasses.replace(base, width=W, height=H, bits=np.ascontiguousarray(bits), px_size_x=MAP / W, px_size_y=MAP / H)
        with capi.Env(oracle, t, n_envs=1, n_rays=8, lidar_mode="fakelidar") as e:
            dt = np.empty((H, W))
            assert oracle.dll.oracle_get_distance_field(e.h, dt.ctypes.data) == 0
        np.testing.assert_array_equal(dt, distance_transform_edt(~wall), err_msg=f"{H}x{W}")


@pytest.mark.parametrize("name", TRACKS)
def test_oracle_step_reproduces_the_reference_fakelidar_g8(oracle, name):
    check_g8(oracle, name)


@pytest.mark.parametrize("name", TRACKS)
def test_oracle_step_reproduces_g2_bit_for_bit(oracle, name):
    assert check_g2_through_step(oracle, name, n_origins=4) >= 1      # some origins have an exact preimage; all scans agree regardless


def test_oracle_rays_that_leave_the_image_read_minus_one_and_negative_indices_wrap(oracle):
    """raycast.py:13-17 tests the bounds only after the next lookup: past the right / bottom edge that is an IndexError (here: the
    ray reads -1), past the left / top edge numpy wraps.  A car outside every wall sees both."""
    t = load_track("small-circle")
    with capi.Env(oracle, t, n_envs=1, n_rays=72, lidar_mode="fakelidar") as e:
        pose = e.pose()
        pose[0, 0:2] = (39.5, -39.5)                       # bottom-right corner of the map, outside the track
        pose[0, 3], pose[0, 6] = 1.0, 0.0
        e.set_pose(pose)
        e.step(1)
        r = e.lidar()[0]
    assert (r == -1.0).any() and (r >= 0).any()


def test_fakelidar_mode_is_refused_where_it_does_not_apply(oracle):
    t = load_track("small-circle")
    with capi.Env(oracle, t, n_envs=1, n_rays=8) as e:
        assert oracle.dll.oracle_get_distance_field(e.h, np.empty(4).ctypes.data) != 0      # RANGEFINDER mode has no distance field
    with pytest.raises(KeyError):
        capi.Env(oracle, t, n_envs=1, n_rays=8, lidar_mode="sonar")


# ------------------------------------------------------------------------------------------------ GPU: libftgp.so
@pytest.mark.gpu
@pytest.mark.parametrize("name", TRACKS)
def test_gpu_distance_transform_and_g8(product, oracle, name):
    t = load_track(name)
    with capi.Env(product, t, n_envs=1, n_rays=8, lidar_mode="fakelidar") as e:
        dt = e.distance_field()
    np.testing.assert_array_equal(sha(dt), np.load(golden("g8_fakelidar_step.npz"))[f"{name}_edt_sha256"])      # scipy's, by hash
    check_g8(product, name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", TRACKS)
def test_gpu_step_reproduces_g2_bit_for_bit(product, name):
    check_g2_through_step(product, name, n_origins=8)


@pytest.mark.gpu
def test_gpu_distance_field_only_in_fakelidar_mode(product):
    with capi.Env(product, load_track("small-circle"), n_envs=1, n_rays=8) as e:
        with pytest.raises(capi.FtgpError) as ei:
            e.distance_field()
        assert ei.value.code == -4


@pytest.mark.gpu
@pytest.mark.parametrize("name,cars,policy,R", [("track", 1, "fast", 1080), ("circle", 1, "nidc", 90), ("small-circle", 3, "nidc", 36),
                                                ("inkscape", 1, "random", 1080)])
def test_gpu_closed_loop_in_fakelidar_mode_matches_the_oracle(product, oracle, name, cars, policy, R):
    """The whole loop -- device driver on the fakelidar scan, K1, K3 -- GPU against oracle: scans and counters bit for bit."""
    t = load_track(name)
    kw = dict(n_envs=24, cars_per_env=cars, n_rays=R, spawn_mode=1, seed=11, lidar_mode="fakelidar")
    with capi.Env(product, t, **kw) as g, capi.Env(oracle, t, **kw) as o:
        oracle.dll.oracle_set_threads(o.h, 8)
        for n in (1, 7, 292):
            g.rollout(policy, n); o.rollout(policy, n)
            np.testing.assert_array_equal(g.lidar(), o.lidar())
            np.testing.assert_array_equal(g.progress(), o.progress())
            np.testing.assert_allclose(g.pose(), o.pose(), rtol=0, atol=1e-12)
            np.testing.assert_array_equal(g.metrics_local(), o.metrics_local())
        assert g.kernel_name().endswith(", true>")


@pytest.mark.gpu
def test_gpu_host_drivers_see_the_fakelidar_scan(product, oracle):
    """ftgp_step + ftgp_get_lidar (the host-driver path) in FAKELIDAR mode, incl. a masked reset in between."""
    t = load_track("track")
    kw = dict(n_envs=5, n_rays=90, spawn_mode=1, seed=2, lidar_mode="fakelidar")
    with capi.Env(product, t, **kw) as g, capi.Env(oracle, t, **kw) as o:
        ctrl = np.tile([1.0, 0.1], (5, 1))
        for e in (g, o):
            e.set_ctrl(ctrl); e.step(40)
            e.reset(np.array([0, 1, 0, 0, 1], dtype=np.uint8)); e.step(3)
        np.testing.assert_array_equal(g.lidar(), o.lidar())
        np.testing.assert_array_equal(g.steps(), o.steps())
