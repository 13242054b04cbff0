#!/usr/bin/env python3
"""Kernel-time ablations on the GPU box (HIP-event ms per launch): policy, ray count, cars per block."""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ft_grandprix_amd import capi
from ft_grandprix_amd.track import load_track

def run(lib, track, n_envs, n_rays, policy, steps, cars=1, warm=50):
    with capi.Env(lib, track, n_envs=n_envs, cars_per_env=cars, n_rays=n_rays, spawn_mode=1, seed=1234) as e:
        e.rollout(policy, warm); e.last_kernel_ms()
        best = 1e9
        for _ in range(3):
            e.rollout(policy, steps); best = min(best, e.last_kernel_ms())
    return best

def main():
    lib = capi.load(); t = load_track(sys.argv[1] if len(sys.argv) > 1 else "track")
    steps = 200
    rows = []
    for n_envs in (4096,):
        for n_rays, policy in ((1080, "fast"), (1080, "nidc"), (1080, "random"), (1080, "lobotomy"), (8, "lobotomy"), (8, "random"), (64, "random"), (256, "random")):
            ms = run(lib, t, n_envs, n_rays, policy, steps)
            rows.append(dict(n_envs=n_envs, n_rays=n_rays, policy=policy, us_per_step=ms * 1e3 / steps, env_steps_per_s=n_envs * steps / ms * 1e3))
            print(rows[-1], flush=True)
    for n_envs in (16384, 65536):
        ms = run(lib, t, n_envs, 1080, "fast", 50, warm=10)
        print(dict(n_envs=n_envs, n_rays=1080, policy="fast", us_per_step=ms * 1e3 / 50, env_steps_per_s=n_envs * 50 / ms * 1e3), flush=True)
    ms = run(lib, t, 4096, 1080, "fast", 100, cars=4, warm=20)
    print(dict(n_envs=4096, cars=4, n_rays=1080, policy="fast", us_per_step=ms * 1e3 / 100, env_steps_per_s=4096 * 100 / ms * 1e3), flush=True)

if __name__ == "__main__":
    main()
