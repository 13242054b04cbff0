#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ft_grandprix_amd import capi
from ft_grandprix_amd.track import load_track
lib = capi.load(); t = load_track("track")
for mode in ("global", "lds"):
    os.environ["FTGP_FIELD"] = mode
    for n_envs, cars, n_rays, policy, steps in ((4096, 1, 1080, "fast", 200), (4096, 1, 1080, "nidc", 200), (4096, 1, 8, "lobotomy", 200), (16384, 1, 1080, "fast", 50), (4096, 4, 1080, "fast", 100), (1024, 1, 1080, "nidc", 200)):
        with capi.Env(lib, t, n_envs=n_envs, cars_per_env=cars, n_rays=n_rays, spawn_mode=1, seed=1234) as e:
            e.rollout(policy, 50); e.last_kernel_ms(); best = 1e9
            for _ in range(3):
                e.rollout(policy, steps); best = min(best, e.last_kernel_ms())
            print(f"{mode:6s} envs {n_envs:6d} cars {cars} rays {n_rays:5d} {policy:9s} {best*1e3/steps:9.2f} us/step {n_envs*steps/best*1e3:14.0f} env-steps/s  [{e.kernel_name()}]", flush=True)
