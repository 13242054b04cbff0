#!/usr/bin/env python3
"""What a launch costs beyond its steps: kernel time (HIP events) AND wall time (rollout + wait, as bench.py times a launch) of 1-, 2-, 3-
and 21-step launches on the headline workload, interleaved so that the scene is the same for all; fixed = t(1) - (t(21) - t(1)) / 20.
   launch_fixed.py [policy] [envs]
The library's environment switches select the host side: FTGP_LAUNCH_PLAIN=1 (hipEventRecord around the launch instead of events on the
dispatch packet), FTGP_NO_FUSED_METRICS=1."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ft_grandprix_amd import capi
from ft_grandprix_amd.track import load_track
policy = sys.argv[1] if len(sys.argv) > 1 else "fast"
envs = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
lib = capi.load()
tag = " ".join(k for k in ("FTGP_LAUNCH_PLAIN", "FTGP_NO_FUSED_METRICS", "FTGP_NO_HOST_SUM", "FTGP_WAIT_SPIN") if os.environ.get(k)) or "default"
with capi.Env(lib, load_track("track"), n_envs=envs, n_rays=1080, spawn_mode=1, seed=1234) as e:
    e.rollout(policy, 200); e.last_kernel_ms()
    t = {1: [], 2: [], 3: [], 21: []}
    w = {1: [], 2: [], 3: [], 21: []}
    for rep in range(24):
        for n in (1, 21, 2, 3):
            t0 = time.perf_counter()
            e.rollout(policy, n); k = e.last_kernel_ms()
            w[n].append((time.perf_counter() - t0) * 1e6); t[n].append(k * 1e3)
    m = {n: float(np.median(v)) for n, v in t.items()}
    mw = {n: float(np.median(v)) for n, v in w.items()}
    step = (m[21] - m[1]) / 20
    wstep = (mw[21] - mw[1]) / 20
    print(f"{policy} {envs} envs [{tag}]: kernel us: 1 step {m[1]:.1f}, 2 steps {m[2]:.1f}, 3 steps {m[3]:.1f}, 21 steps {m[21]:.1f}; steady {step:.2f} us/step; "
          f"fixed per launch {m[1] - step:.1f} us (second step {m[2] - m[1]:.1f}, third {m[3] - m[2]:.1f})")
    print(f"{policy} {envs} envs [{tag}]: wall   us: 1 step {mw[1]:.1f}, 21 steps {mw[21]:.1f}; steady {wstep:.2f} us/step; fixed per launch {mw[1] - wstep:.1f} us; "
          f"a 20-step launch runs at {(mw[1] + 19 * wstep) / 20 / wstep:.3f} x the steady step time, a 100-step launch at {(mw[1] + 99 * wstep) / 100 / wstep:.3f} x")
