#!/usr/bin/env python3
"""End-of-launch metrics folded into the step kernel vs the separate metrics kernel: kernel time (HIP events) and host wall time of
`rollout(20 steps) + metrics + event sync`, the driver's bench shape.   metrics_ab.py [reps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ft_grandprix_amd import capi
from ft_grandprix_amd.track import load_track
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
lib = capi.load(); t = load_track("track")
envs = {}
for name in ("fused", "kernel"):
    if name == "kernel":
        os.environ["FTGP_NO_FUSED_METRICS"] = "1"
    envs[name] = capi.Env(lib, t, n_envs=4096, n_rays=1080, spawn_mode=1, seed=1234)
    os.environ.pop("FTGP_NO_FUSED_METRICS", None)
for e in envs.values():
    e.rollout("fast", 300); e.metrics_allgather()
res = {k: ([], []) for k in envs}
for r in range(reps):
    for name, e in envs.items():
        t0 = time.perf_counter()
        e.rollout("fast", 20); m = e.metrics_allgather(); ms = e.last_kernel_ms()
        res[name][0].append(ms * 1e3); res[name][1].append((time.perf_counter() - t0) * 1e6)
for name, (k, w) in res.items():
    k, w = np.array(k), np.array(w)
    print(f"{name:7s} 20-step launch: kernel median {np.median(k):7.1f} us (min {k.min():7.1f}) | wall median {np.median(w):7.1f} us (min {w.min():7.1f}) | host share {np.median(w - k):5.1f} us")
for e in envs.values():
    e.close()
