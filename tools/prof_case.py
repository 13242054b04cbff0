#!/usr/bin/env python3
"""One configurable rollout for rocprofv3 runs: prof_case.py <n_envs> <n_rays> <policy> <steps> [cars]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ft_grandprix_amd import capi
from ft_grandprix_amd.track import load_track
n_envs, n_rays, policy, steps = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
cars = int(sys.argv[5]) if len(sys.argv) > 5 else 1
lib = capi.CLib(os.environ["FTGP_LIB"], "ftgp_") if os.environ.get("FTGP_LIB") else capi.load()
with capi.Env(lib, load_track("track"), n_envs=n_envs, cars_per_env=cars, n_rays=n_rays, spawn_mode=1, seed=1234) as e:
    e.rollout(policy, 100)
    e.rollout(policy, steps)
    print("kernel ms", e.last_kernel_ms(), "us/step", e.last_kernel_ms() * 1e3 / steps)
