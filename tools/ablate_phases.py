#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ft_grandprix_amd import capi
from ft_grandprix_amd.track import load_track
t = load_track("track")
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
libs = {"full": capi.product_library_path()}
for tag in ("K1", "K2"):
    libs["no" + tag] = os.path.join(root, "gpurun_out", f"libftgp_no{tag}.so")
for n_rays, policy in ((1080, "fast"), (1080, "lobotomy"), (8, "lobotomy")):
    for name, path in libs.items():
        lib = capi.CLib(path, "ftgp_")
        with capi.Env(lib, t, n_envs=4096, n_rays=n_rays, spawn_mode=1, seed=1234) as e:
            e.rollout(policy, 50); e.last_kernel_ms()
            best = 1e9
            for _ in range(3):
                e.rollout(policy, 200); best = min(best, e.last_kernel_ms())
        print(f"rays {n_rays:5d} {policy:9s} {name:6s} {best*1e3/200:8.2f} us/step", flush=True)
