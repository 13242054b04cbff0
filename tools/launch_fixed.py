#!/usr/bin/env python3
"""What a launch costs beyond its steps: kernel time (HIP events) of 1-, 2-, 3- and 21-step launches on the headline workload, interleaved
so that the scene is the same for all; fixed = t(1) - (t(21) - t(1)) / 20.   launch_fixed.py [policy] [envs]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ft_grandprix_amd import capi
from ft_grandprix_amd.track import load_track
policy = sys.argv[1] if len(sys.argv) > 1 else "fast"
envs = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
lib = capi.load()
with capi.Env(lib, load_track("track"), n_envs=envs, n_rays=1080, spawn_mode=1, seed=1234) as e:
    e.rollout(policy, 200); e.last_kernel_ms()
    t = {1: [], 2: [], 3: [], 21: []}
    for rep in range(12):
        for n in (1, 21, 2, 3):
            e.rollout(policy, n); t[n].append(e.last_kernel_ms() * 1e3)
    m = {n: float(np.median(v)) for n, v in t.items()}
    step = (m[21] - m[1]) / 20
    print(f"{policy} {envs} envs: kernel us: 1 step {m[1]:.1f}, 2 steps {m[2]:.1f}, 3 steps {m[3]:.1f}, 21 steps {m[21]:.1f}; steady {step:.2f} us/step; fixed per launch {m[1] - step:.1f} us "
          f"(second step {m[2] - m[1]:.1f}, third {m[3] - m[2]:.1f})")
