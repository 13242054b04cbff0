#!/usr/bin/env python3
"""Would pairing cheap and dear workgroups on a CU shorten a launch?  (diagnostic -DFTGP_STAMPS build: wg_pairing.py lib.so [steps] [warm])
Workgroups b and b + 256 share a CU for the whole launch; the kernel ends when the slowest CU does.  The experiment replays the
bench's shape -- reset, `warm` steps, then the timed launch of `steps` steps -- three ways from the same state:
   identity      workgroup slot b takes cars 8b .. 8b+7 (the product)
   paired        the slots take the groups in an order that puts the cheapest group next to the dearest, the second cheapest
                 next to the second dearest, ...  (cost = the group's own duration in the WARM-UP launch: what a feedback rule
                 could know before the timed launch starts)
   paired-oracle the same with the durations of the timed launch itself under identity (an upper bound for any predictor)
Results are identical in all three (cars are independent; checked)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ft_grandprix_amd import capi
from ft_grandprix_amd.track import load_track
lib = capi.CLib(sys.argv[1], "ftgp_")
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 5
N, CPB = 4096, 8
nb = N // CPB
buf = (C.c_ulonglong * (4 * nb))()
lib.dll.ftgp_debug_set_wg_groups.argtypes = [C.POINTER(C.c_int), C.c_int]


def set_groups(order):
    if order is None:
        assert lib.dll.ftgp_debug_set_wg_groups(None, 0) == 0
    else:
        arr = (C.c_int * nb)(*[int(x) for x in order])
        assert lib.dll.ftgp_debug_set_wg_groups(arr, nb) == 0


def durations(order):
    lib.dll.ftgp_debug_wg_times(buf, nb)
    raw = np.array(list(buf), dtype=np.uint64).reshape(nb, 4)
    d_slot = (raw[:, 1].astype(np.float64) - raw[:, 0].astype(np.float64)) * 0.01          # us, per slot
    d = np.empty(nb)
    d[np.arange(nb) if order is None else np.asarray(order)] = d_slot                      # per group of cars
    return d


def pairing(cost):
    """slot c (c < 256) and slot c + 256 share a CU: rank c next to rank 511 - c"""
    rank = np.argsort(cost, kind="stable")
    order = np.empty(nb, dtype=np.int64)
    order[:nb // 2] = rank[:nb // 2]
    order[nb // 2:] = rank[::-1][:nb // 2]
    return order


def run(order_warm, order_timed):
    with capi.Env(lib, load_track("track"), n_envs=N, n_rays=1080, spawn_mode=1, seed=1234) as e:
        set_groups(order_warm)
        e.rollout("fast", warm); e.last_kernel_ms()
        dw = durations(order_warm)
        set_groups(order_timed(dw) if callable(order_timed) else order_timed)
        o = order_timed(dw) if callable(order_timed) else order_timed
        e.rollout("fast", steps); ms = e.last_kernel_ms()
        dt = durations(o)
        state = e.pose().copy(), e.lidar().copy()
    set_groups(None)
    return ms * 1e3, dw, dt, state


for rep in range(2):
    t_id, dw, dt, ref = run(None, None)
    print(f"identity      {steps:3d}-step launch {t_id:8.1f} us = {t_id / steps:6.2f} us/step | per-group duration mean {dt.mean():7.1f} max {dt.max():7.1f} "
          f"| corr(warm-up, timed) {np.corrcoef(dw, dt)[0, 1]:+.2f}", flush=True)
    t_p, _, _, st = run(None, lambda d: pairing(d))
    same = all(np.array_equal(a, b) for a, b in zip(ref, st))
    print(f"paired        {steps:3d}-step launch {t_p:8.1f} us = {t_p / steps:6.2f} us/step  ({(t_p / t_id - 1) * 100:+.1f} %)  results identical: {same}", flush=True)
    t_o, _, _, st = run(None, pairing(dt))
    same = all(np.array_equal(a, b) for a, b in zip(ref, st))
    print(f"paired-oracle {steps:3d}-step launch {t_o:8.1f} us = {t_o / steps:6.2f} us/step  ({(t_o / t_id - 1) * 100:+.1f} %)  results identical: {same}", flush=True)
