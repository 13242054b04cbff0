#!/usr/bin/env python3
"""Fixed cost of a launch: kernel time (HIP events) and host wall time of ftgp_rollout for n_steps = 1 .. 500 on the headline
workload, and the least-squares fit  time = a + b * n.   launch_sweep.py [lib.so] [policy] [envs] [cars]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ft_grandprix_amd import capi
from ft_grandprix_amd.track import load_track
lib = capi.CLib(sys.argv[1], "ftgp_") if len(sys.argv) > 1 else capi.load()
policy = sys.argv[2] if len(sys.argv) > 2 else "fast"
envs = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
cars = int(sys.argv[4]) if len(sys.argv) > 4 else 1
sizes = [1, 2, 5, 10, 20, 40, 100, 200, 500]
with capi.Env(lib, load_track("track"), n_envs=envs, cars_per_env=cars, n_rays=1080, spawn_mode=1 if cars == 1 else 0, seed=1234) as e:
    e.rollout(policy, 200); e.metrics_allgather(); e.last_kernel_ms()
    rows = []
    for n in sizes:
        k, w = [], []
        for rep in range(5):
            t0 = time.perf_counter()
            e.rollout(policy, n); e.metrics_allgather(); ms = e.last_kernel_ms()
            w.append((time.perf_counter() - t0) * 1e6); k.append(ms * 1e3)
        rows.append((n, min(k), min(w)))
        print(f"{n:4d} steps: kernel {min(k):9.1f} us ({min(k) / n:7.2f} us/step)  wall {min(w):9.1f} us ({min(w) / n:7.2f} us/step)", flush=True)
    a = np.array(rows)
    for col, name in ((1, "kernel"), (2, "wall")):
        b, c = np.polyfit(a[2:, 0], a[2:, col], 1)
        print(f"{name}: {c:6.1f} us fixed + {b:6.3f} us/step")
