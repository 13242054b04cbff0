"""Headless top-down frame dump (what stands in for the reference's GUI viewport, custom.py's renderer).

    frame = render_frame(track, env.pose(), env.lidar())      # H x W x 3 uint8: walls, centre-line, cars, LiDAR returns
    save_png(frame, "frame.png")
    python -m ft_grandprix_amd.render --track track --policy nidc --envs 16 --steps 300 -o frame.png

Geometry is the one the kernels use: pixel (u, v) = ((x - origin_x) / px_size_x, (origin_y - y) / px_size_y); ray j of a car
leaves the LiDAR ring at centre - ring_radius * dir_j with dir_j = R(yaw) (sin phi_j, -cos phi_j), phi_j = radians(360 / R * j - 90)
(template/mushr.em.xml:98-117), so a returned range r ends at start + r * dir_j -- on a wall pixel.  Host-side numpy only."""
import argparse
import sys

import numpy as np

PALETTE = np.array([[230, 57, 70], [42, 157, 143], [233, 196, 106], [69, 123, 157], [244, 162, 97], [131, 56, 236],
                    [58, 134, 255], [255, 0, 110]], dtype=np.uint8)


def world_to_pixel(track, x, y):
    return (np.asarray(x) - track.origin_x) / track.px_size_x, (track.origin_y - np.asarray(y)) / track.px_size_y


def lidar_points(track, pose_row, ranges, lidar_x=-0.0525, lidar_y=0.0, ring_radius=0.03):
    """World coordinates of the points where the rays of one car end (NaN where a ray returned no hit)."""
    x, y, qw, qz = pose_row[0], pose_row[1], pose_row[3], pose_row[6]
    ch, sh = 1.0 - 2.0 * qz * qz, 2.0 * qw * qz
    lcx, lcy = x + (ch * lidar_x - sh * lidar_y), y + (sh * lidar_x + ch * lidar_y)
    n = len(ranges)
    phi = np.radians(360.0 / n * np.arange(n) - 90.0)
    bx, by = np.sin(phi), -np.cos(phi)
    dx, dy = ch * bx - sh * by, sh * bx + ch * by
    r = np.where(np.asarray(ranges) > 0.0, ranges, np.nan).astype(np.float64)
    return lcx + (r - ring_radius) * dx, lcy + (r - ring_radius) * dy


def render_frame(track, poses, ranges=None, cars=None, box=(-0.22, 0.22, -0.14, 0.14), downscale=1, ray_stride=4):
    """poses: rows of ftgp_get_pose (x, y, ., qw, ., ., qz, ...); ranges: matching rows of ftgp_get_lidar (optional);
    cars: indices to draw (default: all, at most 64).  Returns an (H / downscale) x (W / downscale) x 3 uint8 image."""
    wall = track.wall_mask()
    img = np.full(wall.shape + (3,), 245, dtype=np.uint8)
    img[wall] = (40, 40, 40)
    pu, pv = world_to_pixel(track, track.path[:, 0], track.path[:, 1])
    for k in range(len(pu)):                                      # centre-line: 100 points joined by straight runs
        a, b = (pu[k], pv[k]), (pu[(k + 1) % len(pu)], pv[(k + 1) % len(pu)])
        m = int(max(abs(b[0] - a[0]), abs(b[1] - a[1]))) + 1
        uu = np.clip(np.linspace(a[0], b[0], m).astype(int), 0, track.width - 1)
        vv = np.clip(np.linspace(a[1], b[1], m).astype(int), 0, track.height - 1)
        img[vv, uu] = (190, 200, 230)
    poses = np.asarray(poses, dtype=np.float64)
    cars = list(range(min(len(poses), 64))) if cars is None else list(cars)
    for ci in cars:
        colour = PALETTE[ci % len(PALETTE)]
        row = poses[ci]
        if not np.isfinite(row[[0, 1, 3, 6]]).all():
            continue
        if ranges is not None:
            hx, hy = lidar_points(track, row, np.asarray(ranges[ci])[::ray_stride])
            hu, hv = world_to_pixel(track, hx, hy)
            ok = np.isfinite(hu) & (hu >= 0) & (hu < track.width) & (hv >= 0) & (hv < track.height)
            img[hv[ok].astype(int), hu[ok].astype(int)] = (colour // 2 + 100).astype(np.uint8)
        # chassis: the pixels whose centre lies in the body-frame box
        ch, sh = 1.0 - 2.0 * row[6] * row[6], 2.0 * row[3] * row[6]
        cu, cv = world_to_pixel(track, row[0], row[1])
        rad = int(np.ceil(max(abs(box[0]), abs(box[1]), abs(box[2]), abs(box[3])) * 1.5 / min(track.px_size_x, track.px_size_y))) + 1
        u0, u1 = max(int(cu) - rad, 0), min(int(cu) + rad + 1, track.width)
        v0, v1 = max(int(cv) - rad, 0), min(int(cv) + rad + 1, track.height)
        if u0 >= u1 or v0 >= v1:
            continue
        gu, gv = np.meshgrid(np.arange(u0, u1) + 0.5, np.arange(v0, v1) + 0.5)
        wx, wy = track.origin_x + gu * track.px_size_x - row[0], track.origin_y - gv * track.px_size_y - row[1]
        bxx, byy = ch * wx + sh * wy, -sh * wx + ch * wy
        inside = (bxx >= box[0]) & (bxx <= box[1]) & (byy >= box[2]) & (byy <= box[3])
        sub = img[v0:v1, u0:u1]
        sub[inside] = colour
        sub[inside & (bxx > box[1] - 0.08)] = (colour // 2).astype(np.uint8)        # darker nose
    if downscale > 1:
        h, w = (img.shape[0] // downscale) * downscale, (img.shape[1] // downscale) * downscale
        img = img[:h, :w].reshape(h // downscale, downscale, w // downscale, downscale, 3).mean(axis=(1, 3)).astype(np.uint8)
    return img


def save_png(frame: np.ndarray, path: str) -> None:
    from PIL import Image
    Image.fromarray(frame).save(path)


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description="roll a batch out on the GPU and dump a top-down frame of one env")
    ap.add_argument("--track", default="track")
    ap.add_argument("--policy", default="nidc")
    ap.add_argument("--envs", type=int, default=16)
    ap.add_argument("--cars", type=int, default=1)
    ap.add_argument("--rays", type=int, default=1080)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--env", type=int, default=0, help="which env of the batch to draw")
    ap.add_argument("--downscale", type=int, default=2)
    ap.add_argument("-o", "--output", default="frame.png")
    args = ap.parse_args(argv)
    from . import capi
    from .track import load_track
    lib = capi.load()                                              # the HIP library: fails loudly without a GPU
    t = load_track(args.track)
    with capi.Env(lib, t, n_envs=args.envs, cars_per_env=args.cars, n_rays=args.rays, spawn_mode=1 if args.cars == 1 else 0) as env:
        env.rollout(args.policy, args.steps)
        cars = range(args.env * args.cars, (args.env + 1) * args.cars)
        frame = render_frame(t, env.pose(), env.lidar(), cars=cars, downscale=args.downscale)
    save_png(frame, args.output)
    print(f"{args.output}: {frame.shape[1]} x {frame.shape[0]}, env {args.env} of {args.envs} after {args.steps} steps of {args.policy}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
