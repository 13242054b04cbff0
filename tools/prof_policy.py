#!/usr/bin/env python3
"""K5 alone under rocprofv3: prof_policy.py <n_cars> <n_rays> <policy> -- ftgp_policy_eval on scans taken from a short rollout"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ft_grandprix_amd import capi
from ft_grandprix_amd.track import load_track
n, R, policy = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
lib = capi.load()
with capi.Env(lib, load_track("track"), n_envs=n, n_rays=R, spawn_mode=1, seed=1234) as e:
    e.rollout(policy, 200)
    scans = e.lidar()
    for _ in range(3):
        e.policy_eval(policy, scans)
print("done")
