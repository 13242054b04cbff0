#!/usr/bin/env python3
"""Workgroup-shape sweep of the headline configuration: shape_sweep.py lib.so "cpb:wpb[:ldscap]" ..."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ft_grandprix_amd import capi
from ft_grandprix_amd.track import load_track
lib = capi.CLib(sys.argv[1], "ftgp_")
t = load_track("track")
for spec in sys.argv[2:]:
    p = spec.split(":")
    os.environ["FTGP_CARS_PER_BLOCK"], os.environ["FTGP_WAVES_PER_BLOCK"] = p[0], p[1]
    os.environ["FTGP_LDS_CAP_KB"] = p[2] if len(p) > 2 else "80"
    res = []
    for policy in ("fast", "random"):
        with capi.Env(lib, t, n_envs=4096, n_rays=1080, spawn_mode=1, seed=1234) as e:
            e.rollout(policy, 100); e.last_kernel_ms(); best = 1e9
            for _ in range(3):
                e.rollout(policy, 300); best = min(best, e.last_kernel_ms())
        res.append(f"{policy} {best * 1e3 / 300:7.2f} us/step")
    print(f"cars/block {p[0]:>2s} waves/block {p[1]:>2s} lds cap {os.environ['FTGP_LDS_CAP_KB']:>3s} KB: " + "  ".join(res), flush=True)
