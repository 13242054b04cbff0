#!/usr/bin/env python3
"""Does the metrics exchange of launch k really run BESIDE launch k + 1 when a communicator exists?  One rank, a one-rank RCCL communicator (the path N > 1
takes, on the box's one GPU), the driver's bench shape (20-step launches of the headline batch):  tools/diag/exchange_overlap.py [lib.so ...]
Prints, per library, the median wall time of a launch and how long collecting an exchange held the host up, with and without the communicator."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from ft_grandprix_amd import capi, dist as ftdist
from ft_grandprix_amd.track import load_track
libs = sys.argv[1:] or [capi.product_library_path()]
for path in libs:
    lib = capi.CLib(path, "ftgp_")
    for comm in (False, True):
        with capi.Env(lib, load_track("track"), n_envs=4096, n_rays=1080, spawn_mode=1, seed=1234) as e:
            if comm:
                e.comm_init(capi.comm_unique_id(lib), 0, 1)
            ex = ftdist.DeviceExchange(e)
            e.rollout("fast", 100); e.last_kernel_ms()
            ftdist.run_timed(e, "fast", 20, 5, ex)
            r = ftdist.run_timed(e, "fast", 20, 31, ex)
            wall, ends, km = np.array(r["wall_s"]) * 1e6, np.array(r["exchange_end_s"]) * 1e6, np.array(r["kernel_ms"]) * 1e3
            print(f"{os.path.basename(path):24s} {'one-rank RCCL communicator' if comm else 'no communicator':28s}: launch wall median {np.median(wall):7.1f} us (kernel {np.median(km):7.1f}), "
                  f"collecting an exchange held the host up: median {np.median(ends):6.1f} us, max {ends.max():6.1f} us", flush=True)
