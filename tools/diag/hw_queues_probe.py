import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
from ft_grandprix_amd import capi, dist as ftdist
from ft_grandprix_amd.track import load_track
lib = capi.load()
class NoExchange:
    after_sync = False; open = False
    def begin(self): pass
    def end(self): return None
with capi.Env(lib, load_track("track"), n_envs=4096, n_rays=1080, spawn_mode=1, seed=1234) as e:
    if os.environ.get("XO_COMM"): e.comm_init(capi.comm_unique_id(lib), 0, 1)
    e.rollout("fast", 100); e.last_kernel_ms()
    ftdist.run_timed(e, "fast", 20, 5, NoExchange())
    r = ftdist.run_timed(e, "fast", 20, 31, NoExchange())
    wall, km = np.array(r["wall_s"]) * 1e6, np.array(r["kernel_ms"]) * 1e3
    print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', 'default')} comm={bool(os.environ.get('XO_COMM'))}: launch wall median {np.median(wall):7.1f} us, kernel {np.median(km):7.1f}", flush=True)
