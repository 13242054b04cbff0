#!/usr/bin/env python3
"""How evenly the persistent workgroups finish (diagnostic -DFTGP_STAMPS build): wg_spread.py lib.so [policy] [envs] [cars] [steps ...]
Per launch size: kernel time, and the spread of the workgroups' entry / exit times on the 100-MHz wall clock."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ft_grandprix_amd import capi
from ft_grandprix_amd.track import load_track
lib = capi.CLib(sys.argv[1], "ftgp_")
policy = sys.argv[2] if len(sys.argv) > 2 else "fast"
envs = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
cars = int(sys.argv[4]) if len(sys.argv) > 4 else 1
sizes = [int(a) for a in sys.argv[5:]] or [20, 100, 500]
with capi.Env(lib, load_track("track"), n_envs=envs, cars_per_env=cars, n_rays=1080, spawn_mode=1 if cars == 1 else 0, seed=1234) as e:
    cpb = 8 if cars == 1 else 8
    nb = (envs * cars + cpb - 1) // cpb
    buf = (C.c_ulonglong * (4 * nb))()
    e.rollout(policy, 100); e.last_kernel_ms()
    for n in sizes:
        for rep in range(2):
            e.rollout(policy, n); ms = e.last_kernel_ms()
        lib.dll.ftgp_debug_wg_times(buf, nb)
        raw = np.array(list(buf), dtype=np.uint64).reshape(nb, 4)
        t = raw[:, :2].astype(np.float64) * 0.01      # us
        t0 = t[:, 0].min()
        start, end = t[:, 0] - t0, t[:, 1] - t0
        dur = end - start
        print(f"{policy} {envs}x{cars} {n:4d} steps: kernel {ms * 1e3:8.1f} us = {ms * 1e3 / n:6.2f} us/step | entry spread {start.max():5.1f} us | "
              f"exit min/mean/max {end.min():8.1f} {end.mean():8.1f} {end.max():8.1f} us | per-WG duration mean {dur.mean():8.1f} p99 {np.percentile(dur, 99):8.1f} max {dur.max():8.1f}",
              flush=True)
        # which CU ran which workgroup: HW_ID bits [11:8] CU, [12] SH, [15:13] SE (gfx9), XCC_ID bits [3:0]
        hw, xcc = raw[:, 2].astype(np.int64), raw[:, 3].astype(np.int64) & 15
        cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)
        ends = {}
        for b in range(nb):
            ends.setdefault(int(cu[b]), []).append(end[b])
        cu_end = np.array([max(v) for v in ends.values()]); per = np.array([len(v) for v in ends.values()])
        print(f"     {len(ends)} CUs, workgroups per CU {per.min()}..{per.max()}; CU busy-until min/mean/max {cu_end.min():8.1f} {cu_end.mean():8.1f} {cu_end.max():8.1f} us "
              f"-> idle tail {1 - cu_end.mean() / cu_end.max():.3f} of the kernel", flush=True)
        np.save(f"gpurun_out/wg_times_{policy}_{envs}x{cars}_{n}.npy", np.concatenate([t - t0, raw[:, 2:].astype(np.float64)], axis=1))
