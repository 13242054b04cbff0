#!/usr/bin/env python3
"""Step the GPU and the oracle side by side and report the first quantity that differs."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ft_grandprix_amd import capi
from ft_grandprix_amd.track import load_track
from tests.helpers import load_oracle

policy = sys.argv[1] if len(sys.argv) > 1 else "fast"
track = sys.argv[2] if len(sys.argv) > 2 else "track"
n_envs = int(sys.argv[3]) if len(sys.argv) > 3 else 48
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
chunk = int(sys.argv[5]) if len(sys.argv) > 5 else 1
lib, ora = capi.load(), load_oracle()
t = load_track(track)
kw = dict(n_envs=n_envs, n_rays=1080, spawn_mode=1, seed=1234, lap_target=2)
g, o = capi.Env(lib, t, **kw), capi.Env(ora, t, **kw)
ora.dll.oracle_set_threads(o.h, 8)
for s in range(0, steps, chunk):
    g.rollout(policy, chunk); o.rollout(policy, chunk)
    cg, co = g.ctrl(), o.ctrl()
    pg, po = g.pose(), o.pose()
    rg, ro = g.lidar(), o.lidar()
    bad = []
    if not np.array_equal(cg, co): bad.append(("ctrl", np.argwhere(cg != co)[:4].tolist(), cg[cg != co][:4], co[cg != co][:4]))
    if not np.array_equal(pg, po): bad.append(("pose", np.argwhere(pg != po)[:4].tolist(), pg[pg != po][:4], po[pg != po][:4]))
    if not np.array_equal(rg, ro): bad.append(("lidar", np.argwhere(rg != ro)[:6].tolist(), rg[rg != ro][:6], ro[rg != ro][:6]))
    if not np.array_equal(g.progress(), o.progress()): bad.append(("progress",))
    if bad:
        print("first divergence after step", s + chunk)
        for b in bad: print(b)
        e = bad[0][1][0][0] if len(bad[0]) > 1 else 0
        print("pose gpu", pg[e]); print("pose ora", po[e]); print("ctrl gpu", cg[e], "ora", co[e])
        break
else:
    print("no divergence in", steps, "steps")
