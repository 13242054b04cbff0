#!/usr/bin/env python3
"""Inputs of tools/sweep_model.cpp from the CPU oracle (tools only): model_inputs.py [track] [policy] [envs] [steps] [cars]
 -> /tmp/track.raw (int32 W, H, wpr; bitmap) and /tmp/poses.bin (n_cars, px sizes, origin; x y qw qz per car)."""
import os, struct, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from ft_grandprix_amd import capi
from ft_grandprix_amd.track import load_track
from tests.helpers import load_oracle
name = sys.argv[1] if len(sys.argv) > 1 else "track"
policy = sys.argv[2] if len(sys.argv) > 2 else "fast"
envs = int(sys.argv[3]) if len(sys.argv) > 3 else 512
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 300
cars = int(sys.argv[5]) if len(sys.argv) > 5 else 1
t = load_track(name)
ora = load_oracle()
bits = np.ascontiguousarray(t.bits, dtype=np.uint32)
with open("/tmp/track.raw", "wb") as f:
    f.write(struct.pack("<3i", t.width, t.height, bits.shape[1])); f.write(bits.tobytes())
with capi.Env(ora, t, n_envs=envs, cars_per_env=cars, n_rays=1080, spawn_mode=1 if cars == 1 else 0, seed=1234) as o:
    ora.dll.oracle_set_threads(o.h, 8)
    o.rollout(policy, steps)
    p = o.pose()
    o.rollout(policy, 1)
    p_next = o.pose()            # the same cars one step later (what-ifs that carry something from step to step: PREV=/tmp/poses_next.bin)
for path, q in (("/tmp/poses.bin", p), ("/tmp/poses_next.bin", p_next)):
    with open(path, "wb") as f:
        f.write(struct.pack("<6d", len(q), t.px_size_x, t.px_size_y, t.origin_x, t.origin_y, 0.0))
        f.write(np.ascontiguousarray(q[:, [0, 1, 3, 6]]).tobytes())
print("wrote /tmp/track.raw /tmp/poses.bin /tmp/poses_next.bin:", len(p), "cars after", steps, "steps of", policy)
