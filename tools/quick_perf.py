#!/usr/bin/env python3
"""Kernel time of the three throughput configurations (BASELINE.json configs[2], [1], [4]): quick_perf.py [lib.so ...]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ft_grandprix_amd import capi
from ft_grandprix_amd.track import load_track
libs = sys.argv[1:] or [capi.product_library_path()]
cases = (("track", "fast", 4096, 1, 300), ("circle", "nidc", 1024, 1, 300), ("track", "fast", 4096, 4, 100), ("track", "random", 4096, 1, 300))
if os.environ.get("QUICK_ROSTER"):         # + template/cars/cars.json (nidc, fast, nidc) with every car's own driver on the device
    cases = cases + (("track", "roster", 4096, 3, 100),)
if os.environ.get("QUICK_CASES"):          # e.g. QUICK_CASES=0,2: only those rows
    cases = tuple(cases[int(i)] for i in os.environ["QUICK_CASES"].split(","))
if os.environ.get("QUICK_ENVS"):           # another batch size for the selected rows
    cases = tuple((n, p, int(os.environ["QUICK_ENVS"]), c, s) for n, p, _, c, s in cases)
for path in libs:
    lib = capi.CLib(path, "ftgp_")
    out = []
    for name, policy, envs, cars, steps in cases:
        with capi.Env(lib, load_track(name), n_envs=envs, cars_per_env=cars, n_rays=1080, spawn_mode=int(os.environ.get('QUICK_SPAWN_MODE', 1 if (cars == 1 or os.environ.get('QUICK_SPAWN')) else 0)), seed=1234) as e:      # QUICK_SPAWN=1: bench.py's spawn rule for multi-car envs too
            if policy == "roster":
                e.set_car_policies(["nidc", "fast", "nidc"]); policy = "per_car"
            e.rollout(policy, 100); e.last_kernel_ms(); best = 1e9
            for _ in range(3):
                e.rollout(policy, steps); best = min(best, e.last_kernel_ms())
            short = 1e9
            if os.environ.get("QUICK_SHORT"):          # the driver's bench shape: 20 steps per launch
                for _ in range(6):
                    e.rollout(policy, 20); short = min(short, e.last_kernel_ms())
        out.append(f"{name}/{'roster' if policy == 'per_car' else policy}/{envs}x{cars}: {best * 1e3 / steps:7.2f} us/step = {envs * steps / best / 1e3:6.2f} M env-steps/s"
                   + (f" [20-step launch {short * 1e3 / 20:6.2f} us/step]" if short < 1e9 else ""))
    print(os.path.basename(path), " | ".join(out), flush=True)
