#!/usr/bin/env python3
"""A variant library (tools/build_variant.sh) against the oracle before its timings are believed: tools/diag/variant_check.py lib.so [...]
64 envs x 1080 rays on `track` (16-sector field forced) and 24 envs on `circle`, 300 closed-loop steps of `fast` / `nidc`: scans and counters bit for bit."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from ft_grandprix_amd import capi
from ft_grandprix_amd.track import load_track
from tests.helpers import load_oracle
ora = load_oracle()
for path in sys.argv[1:]:
    lib = capi.CLib(path, "ftgp_")
    ok = True
    for name, policy, envs, sectors in (("track", "fast", 64, "16"), ("track", "fast", 40, "8"), ("circle", "nidc", 24, "64"), ("inkscape", "random", 33, "16")):
        os.environ["FTGP_SECTORS_RT"] = sectors
        t = load_track(name)
        kw = dict(n_envs=envs, n_rays=1080, spawn_mode=1, seed=99)
        with capi.Env(lib, t, **kw) as g, capi.Env(ora, t, **kw) as o:
            for k in range(3):
                g.rollout(policy, 100); o.rollout(policy, 100)
                same = np.array_equal(g.lidar(), o.lidar()) and np.array_equal(g.progress(), o.progress()) and np.allclose(g.pose(), o.pose(), rtol=0, atol=1e-9)
                ok = ok and same
                if not same:
                    d = np.argwhere(g.lidar() != o.lidar())
                    print(f"  MISMATCH {name}/{policy} sectors {sectors} after {100 * (k + 1)} steps: {len(d)} ranges differ, first {d[:3].tolist()}")
                    break
    del os.environ["FTGP_SECTORS_RT"]
    print(os.path.basename(path), "OK" if ok else "FAILED", flush=True)
