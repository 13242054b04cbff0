"""Headless frame dump: geometry of the picture = geometry of the kernels."""
import numpy as np
from scipy.ndimage import distance_transform_edt

from ft_grandprix_amd import capi, render
from ft_grandprix_amd.track import load_track
from tests.helpers import load_oracle


def test_frame_dump_matches_the_simulated_geometry(tmp_path):
    t = load_track("small-circle")
    ora = load_oracle()
    with capi.Env(ora, t, n_envs=4, n_rays=360, spawn_mode=1, seed=3) as env:
        env.rollout("nidc", 60)
        pose, ranges = env.pose(), env.lidar()
    # every returned range ends on (or right next to) a wall pixel: ray order, ring offset and axes are the kernels'
    near_wall = distance_transform_edt(~t.wall_mask()) <= 3.0      # the range ends where the ray enters the wall pixel
    for ci in range(4):
        hx, hy = render.lidar_points(t, pose[ci], ranges[ci])
        hu, hv = render.world_to_pixel(t, hx, hy)
        ok = np.isfinite(hu)
        assert ok.sum() > 300
        assert near_wall[hv[ok].astype(int), hu[ok].astype(int)].all()
    frame = render.render_frame(t, pose, ranges, downscale=1)
    assert frame.shape == (t.height, t.width, 3) and frame.dtype == np.uint8
    cu, cv = render.world_to_pixel(t, pose[0, 0], pose[0, 1])
    assert (frame[int(cv), int(cu)] == render.PALETTE[0]).all() or (frame[int(cv), int(cu)] == render.PALETTE[0] // 2).all()
    assert (frame[t.wall_mask()] == 40).mean() > 0.95              # walls stay visible
    small = render.render_frame(t, pose, None, cars=[1], downscale=4)
    assert small.shape == (t.height // 4, t.width // 4, 3)
    out = tmp_path / "frame.png"
    render.save_png(small, str(out))
    assert out.stat().st_size > 1000
