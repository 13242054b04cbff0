import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
from ft_grandprix_amd import capi, dist as ftdist
from ft_grandprix_amd.track import load_track
lib = capi.CLib(sys.argv[1], "ftgp_")
class NoExchange:
    after_sync = False; open = False
    def begin(self): pass
    def end(self): return None
for comm, exch in ((False, True), (True, False), (True, True), (False, False)):
    with capi.Env(lib, load_track("track"), n_envs=4096, n_rays=1080, spawn_mode=1, seed=1234) as e:
        if comm: e.comm_init(capi.comm_unique_id(lib), 0, 1)
        ex = ftdist.DeviceExchange(e) if exch else NoExchange()
        e.rollout("fast", 100); e.last_kernel_ms()
        STEPS = int(os.environ.get("XO_STEPS", "20")); ftdist.run_timed(e, "fast", STEPS, 5, ex)
        r = ftdist.run_timed(e, "fast", STEPS, 31, ex)
        wall, km = np.array(r["wall_s"]) * 1e6, np.array(r["kernel_ms"]) * 1e3
        print(f"communicator {comm!s:5s} exchange calls {exch!s:5s}: launch wall median {np.median(wall):7.1f} us, kernel {np.median(km):7.1f}", flush=True)
