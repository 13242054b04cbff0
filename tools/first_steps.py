#!/usr/bin/env python3
"""Kernel time of consecutive short launches right after a reset (headline workload): first_steps.py [lib.so] [steps_per_launch] [launches]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ft_grandprix_amd import capi
from ft_grandprix_amd.track import load_track
lib = capi.CLib(sys.argv[1], "ftgp_") if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else capi.load()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
k = int(sys.argv[3]) if len(sys.argv) > 3 else 24
for policy in ("fast", "lobotomy"):
    with capi.Env(lib, load_track("track"), n_envs=4096, n_rays=1080, spawn_mode=1, seed=1234) as e:
        out = []
        for i in range(k):
            e.rollout(policy, n); out.append(e.last_kernel_ms() * 1e3 / n)
        print(policy, f"{n}-step launches after reset, us/step:", " ".join(f"{x:.1f}" for x in out), flush=True)
