#!/usr/bin/env python3
"""diagnostic: determinism / launch-split invariance of the multi-car config at full size"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from ft_grandprix_amd import capi
from ft_grandprix_amd.track import load_track
from tests.helpers import load_oracle
lib = capi.load(); ora = load_oracle()
t = load_track("track")
kw = dict(n_envs=4096, cars_per_env=4, n_rays=1080, spawn_mode=0, seed=1234, lap_target=3)
def run(chunks):
    e = capi.Env(lib, t, **kw)
    for c in chunks: e.rollout("fast", c)
    out = (e.lidar(), e.pose(), e.progress()); e.close(); return out
a = run([60]); b = run([60]); c = run([20, 40]); d = run([1] * 60)
o = capi.Env(ora, t, **dict(kw, n_envs=8)); ora.dll.oracle_set_threads(o.h, 8); o.rollout("fast", 60)
ol, op = o.lidar(), o.pose()
for name, x in (("same-again", b), ("20+40", c), ("60x1", d)):
    dl = (a[0] != x[0]); dp = (a[1] != x[1]).any(1)
    print(name, "lidar diffs", int(dl.sum()), "cars with lidar diffs", int(dl.any(1).sum()), "pose diffs", int(dp.sum()),
          "first cars", np.nonzero(dl.any(1))[0][:12], flush=True)
for name, x in (("60", a), ("20+40", c), ("60x1", d)):
    print(name, "vs oracle prefix: lidar", int((x[0][:32] != ol).sum()), "pose", int((np.abs(x[1][:32] - op) > 1e-12).sum()), flush=True)
