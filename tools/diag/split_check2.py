#!/usr/bin/env python3
"""diagnostic: two handles alive, kernels of both in flight at once (the shape of test_config2_and_config5_full_size_properties)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from ft_grandprix_amd import capi
from ft_grandprix_amd.track import load_track
lib = capi.load()
t = load_track("track")
kw = dict(n_envs=4096, cars_per_env=4, n_rays=1080, spawn_mode=0, seed=1234, lap_target=3)
ref = capi.Env(lib, t, **kw); ref.rollout("fast", 60); rl = ref.lidar(); ref.close()
for trial in range(3):
    g, g2 = capi.Env(lib, t, **kw), capi.Env(lib, t, **kw)
    g.rollout("fast", 60); g2.rollout("fast", 20); g2.rollout("fast", 40)
    a, b = g.lidar(), g2.lidar()
    for name, x in (("g", a), ("g2", b)):
        dl = x != rl
        cars = np.nonzero(dl.any(1))[0]
        print(trial, name, "diffs", int(dl.sum()), "cars", cars[:16], "n", len(cars), flush=True)
        for c in cars[:3]:
            j = np.nonzero(dl[c])[0]
            print("   car", c, "rays", j[:8], "...", len(j), "got", x[c, j[:4]], "ref", rl[c, j[:4]], flush=True)
    g.close(); g2.close()
