#!/usr/bin/env python3
"""What would another env -> workgroup assignment buy?  Emulated at the workload level (no kernel change): after 100 steps the
poses of the headline batch are permuted across envs -- the work of every car is unchanged, only which cars share a workgroup.
   consecutive   as it is: workgroup b = envs 8b .. 8b+7 (neighbours on the track under the benchmark's spawn rule)
   interleaved   workgroup b = the poses of envs b, b + 512, b + 1024, ...
   cost-dealt    envs sorted by their sweep cost (march iterations of their 1080 rays, tools/car_cost.cpp) and dealt to the
                 workgroups in snake order, so that every workgroup carries the same cost
   interleave_probe.py [lib.so] [steps]"""
import os, struct, subprocess, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ft_grandprix_amd import capi
from ft_grandprix_amd.track import load_track
lib = capi.CLib(sys.argv[1], "ftgp_") if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else capi.load()
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 500
N, CPB = 4096, 8
nb = N // CPB
t = load_track("track")
subprocess.check_call(["g++", "-O2", "-std=c++17", "-I.", "tools/car_cost.cpp", "-o", "/tmp/car_cost"])
with open("/tmp/track.raw", "wb") as f:
    f.write(struct.pack("<3i", t.width, t.height, t.bits.shape[1])); f.write(np.ascontiguousarray(t.bits, dtype=np.uint32).tobytes())


def costs(pose):
    with open("/tmp/poses.bin", "wb") as f:
        f.write(struct.pack("<6d", len(pose), t.px_size_x, t.px_size_y, t.origin_x, t.origin_y, 0.0))
        f.write(np.ascontiguousarray(pose[:, [0, 1, 3, 6]]).tobytes())
    out = subprocess.check_output(["/tmp/car_cost", "/tmp/track.raw", "/tmp/poses.bin", "1080"]).decode().split("\n")
    return np.array([float(l.split()[0]) for l in out if l.strip()])


for mode in ("consecutive", "interleaved", "cost-dealt", "consecutive"):
    with capi.Env(lib, t, n_envs=N, n_rays=1080, spawn_mode=1, seed=1234) as e:
        e.rollout("fast", 100)
        pose = e.pose()
        if mode == "interleaved":
            src = np.array([(i % CPB) * nb + i // CPB for i in range(N)])
        elif mode == "cost-dealt":
            order = np.argsort(-costs(pose), kind="stable")
            src = np.empty(N, dtype=np.int64)
            for r, env in enumerate(order):
                row = r // nb
                col = r % nb if row % 2 == 0 else nb - 1 - r % nb
                src[col * CPB + row] = env
        else:
            src = np.arange(N)
        e.set_pose(pose[src]); e.eval_progress()
        e.rollout("fast", 20); e.last_kernel_ms()
        out = []
        for n in (20, 20, 20, steps, steps):
            e.rollout("fast", n); out.append(e.last_kernel_ms() * 1e3 / n)
        print(f"{mode:12s} us/step: 20-step launches {out[0]:.2f} {out[1]:.2f} {out[2]:.2f} | {steps}-step launches {out[3]:.2f} {out[4]:.2f}", flush=True)
