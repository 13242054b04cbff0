#!/usr/bin/env python3
"""Sub-phases of the driver (K5) and of the dynamics wave (K1 + K3) from a -DFTGP_DIAG -DFTGP_STAMPS library (tools/stamps.sh builds it):
shader clocks per visit.   substamps.py lib.so [policy] [envs] [cars] [track]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ft_grandprix_amd import capi
from ft_grandprix_amd.track import load_track
lib = capi.CLib(sys.argv[1], "ftgp_")
policy = sys.argv[2] if len(sys.argv) > 2 else "fast"
envs = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
cars = int(sys.argv[4]) if len(sys.argv) > 4 else 1
track = sys.argv[5] if len(sys.argv) > 5 else "track"
buf = (C.c_ulonglong * 32)()
names = ["K1 wheel terms", "K1 wall circles + softeners", "K1 car-car contacts", "K1 sync", "K1 sum + integrate", "K1 commit", "K3 100 path points + argmin",
         "K3 race state", "frames", "K5 flag pass", "K5 ordered extension", "K5 argmax pass", "K5 wave reductions", "K5 controls"]
slots = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11]
label = {0: "K1 wheel terms", 1: "K1 wall circles (+ softeners)", 2: "K1 car-car contacts + sync", 3: "K1 sum + integrate + sync", 4: "K1 commit + sync",
         5: "K3 path points + argmin", 6: "K3 race state + sync", 7: "frames", 8: "K5 flag pass", 9: "K5 ordered extension", 10: "K5 argmax + reductions", 11: "K5 controls (binary64)"}
with capi.Env(lib, load_track(track), n_envs=envs, cars_per_env=cars, n_rays=1080, spawn_mode=1 if (cars == 1 or os.environ.get("QUICK_SPAWN")) else 0, seed=1234) as e:
    e.rollout(policy, 100); e.last_kernel_ms(); lib.dll.ftgp_debug_substamps(buf)
    e.rollout(policy, 300); ms = e.last_kernel_ms(); lib.dll.ftgp_debug_substamps(buf)
s = list(buf)
print(f"{policy} {envs}x{cars} {track}: {ms * 1e3 / 300:.2f} us/step (instrumented); shader clocks per visit:")
for k in slots:
    if s[16 + k]:
        print(f"  {label[k]:34s} {s[k] / s[16 + k]:8.0f}   ({s[16 + k]} visits)")
print(f"  K1 + K3 + frames together {sum(s[k] / max(s[16 + k], 1) for k in range(0, 8)):8.0f}; K5 together {sum(s[k] / max(s[16 + k], 1) for k in range(8, 12)):8.0f}")
