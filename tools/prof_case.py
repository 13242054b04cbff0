#!/usr/bin/env python3
"""One configurable rollout for the counter passes: prof_case.py envs rays policy steps [cars] [track]  (the last launch is the measured one;
FTGP_PROF_LIDAR=fakelidar: the FAKELIDAR mode of the step kernel)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ft_grandprix_amd import capi
from ft_grandprix_amd.track import load_track
envs, rays, policy, steps = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
cars = int(sys.argv[5]) if len(sys.argv) > 5 else 1
track = sys.argv[6] if len(sys.argv) > 6 else "track"
lib = capi.CLib(os.environ["FTGP_LIB"], "ftgp_") if os.environ.get("FTGP_LIB") else capi.load()
with capi.Env(lib, load_track(track), n_envs=envs, cars_per_env=cars, n_rays=rays, spawn_mode=int(os.environ.get('PROF_SPAWN_MODE', '1')), seed=1234, lidar_mode=os.environ.get("FTGP_PROF_LIDAR", "rangefinder")) as e:      # the spawn rule bench.py uses
    e.rollout(policy, 50); e.last_kernel_ms()
    e.rollout(policy, steps)
    print("kernel ms", e.last_kernel_ms(), flush=True)
