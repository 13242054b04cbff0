#!/usr/bin/env python3
"""What of a communicator's presence costs short launches their time?  Candidates tried one at a time on the headline batch, 20-step launches:
extra active HIP streams (more hardware queues for the command processor to arbitrate), librccl loaded without a communicator (ncclGetUniqueId only), a one-rank communicator.
tools/diag/queue_count_probe.py [lib.so]"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from ft_grandprix_amd import capi, dist as ftdist
from ft_grandprix_amd.track import load_track
lib = capi.CLib(sys.argv[1] if len(sys.argv) > 1 else capi.product_library_path(), "ftgp_")
hip = ctypes.CDLL("libamdhip64.so")
class NoExchange:
    after_sync = False; open = False
    def begin(self): pass
    def end(self): return None
def timed(tag, prepare):
    with capi.Env(lib, load_track("track"), n_envs=4096, n_rays=1080, spawn_mode=1, seed=1234) as e:
        keep = prepare(e)
        e.rollout("fast", 100); e.last_kernel_ms()
        ftdist.run_timed(e, "fast", 20, 5, NoExchange())
        r = ftdist.run_timed(e, "fast", 20, 31, NoExchange())
        wall, km = np.array(r["wall_s"]) * 1e6, np.array(r["kernel_ms"]) * 1e3
        print(f"{tag:44s}: launch wall median {np.median(wall):7.1f} us, kernel {np.median(km):7.1f}", flush=True)
        del keep
def streams(n):
    def prep(e):
        out = []
        for _ in range(n):
            s = ctypes.c_void_p(); assert hip.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0
            p = ctypes.c_void_p(); assert hip.hipMalloc(ctypes.byref(p), 4096) == 0
            assert hip.hipMemsetAsync(p, 0, 4096, s) == 0 and hip.hipStreamSynchronize(s) == 0      # the stream has had work: it owns a hardware queue
            out.append((s, p))
        return out
    return prep
for rnd in range(2):
    timed("plain", lambda e: None)
    timed("4 extra streams that have had work", streams(4))
    timed("12 extra streams that have had work", streams(12))
    timed("librccl loaded, ncclGetUniqueId only", lambda e: capi.comm_unique_id(lib))
    timed("one-rank communicator", lambda e: e.comm_init(capi.comm_unique_id(lib), 0, 1))
