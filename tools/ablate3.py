#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ft_grandprix_amd import capi
from ft_grandprix_amd.track import load_track
t = load_track("track")
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for name, path in (("rpl1", capi.product_library_path()), ("rpl2", os.path.join(root, "gpurun_out", "libftgp_rpl2.so"))):
    lib = capi.CLib(path, "ftgp_")
    for n_envs, n_rays, policy, steps in ((4096, 1080, "fast", 200), (1024, 1080, "nidc", 200), (256, 1080, "nidc", 200), (64, 90, "nidc", 500), (65536, 1080, "fast", 30)):
        with capi.Env(lib, t, n_envs=n_envs, n_rays=n_rays, spawn_mode=1, seed=1234) as e:
            e.rollout(policy, 30); e.last_kernel_ms(); best = 1e9
            for _ in range(3):
                e.rollout(policy, steps); best = min(best, e.last_kernel_ms())
            print(f"{name} envs {n_envs:6d} rays {n_rays:5d} {policy:6s} {best*1e3/steps:9.2f} us/step {n_envs*steps/best*1e3:14.0f} env-steps/s", flush=True)
