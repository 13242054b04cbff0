#!/usr/bin/env python3
"""Phase timing of the step kernel from a -DFTGP_STAMPS diagnostic build: stamps.py lib.so [policy] [envs] [cars] [track]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ft_grandprix_amd import capi
from ft_grandprix_amd.track import load_track
lib = capi.CLib(sys.argv[1], "ftgp_")
policy = sys.argv[2] if len(sys.argv) > 2 else "fast"
envs = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
cars = int(sys.argv[4]) if len(sys.argv) > 4 else 1
track = sys.argv[5] if len(sys.argv) > 5 else "track"
buf = (C.c_ulonglong * 16)()
with capi.Env(lib, load_track(track), n_envs=envs, cars_per_env=cars, n_rays=1080, spawn_mode=1 if (cars == 1 or os.environ.get("QUICK_SPAWN")) else 0, seed=1234) as e:
    e.rollout(policy, 100); e.last_kernel_ms(); lib.dll.ftgp_debug_stamps(buf)
    e.rollout(policy, 300); ms = e.last_kernel_ms(); lib.dll.ftgp_debug_stamps(buf)
s = list(buf)
nw, n0 = max(s[5], 1), max(s[7], 1)
print(f"{policy} {envs}x{cars}: {ms * 1e3 / 300:.2f} us/step; shader clocks per wave-step: driver (summed over driver waves / all waves) {s[0] / nw:.0f}, "
      f"sweep {s[3] / nw:.0f}, barrier wait {s[4] / nw:.0f}, whole step {s[6] / nw:.0f}; dynamics (per call) {s[2] / n0:.0f}")
print(f"  inside the sweep, per wave-step: {s[9] / nw:.1f} rounds, store + hand-out + init {s[8] / nw:.0f} clocks ({s[8] / max(s[9], 1):.0f} per round), "
      f"march {s[10] / nw:.0f} clocks over {s[11] / nw:.1f} iterations ({s[10] / max(s[11], 1):.0f} per iteration)")
