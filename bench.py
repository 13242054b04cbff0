#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec at 4096 envs x 1080-ray LiDAR per GPU (BASELINE.json configs[2]).

A "step" is one pass of the hot path over the whole batch: driver (fast, on device) -> LiDAR sweep written
to HBM -> integrate -> lap progress, for every env.  All inputs are resident in HBM before the timed region;
the K timed steps run as ONE persistent launch (the state stays in registers between steps).

    python bench.py --gpus N --steps K --warmup W        # N > 1: launched by torch.distributed.run, one rank per GPU

Envs shard across ranks with no data-path collective (weak scaling: 4096 envs per GPU); the only exchange is the
end-of-launch metrics all-gather over RCCL, issued on a side stream.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from ft_grandprix_amd import capi  # noqa: E402
from ft_grandprix_amd.track import load_track  # noqa: E402

ALGO_BYTES_PER_ENV_STEP = lambda n_rays, cars: cars * (4 * n_rays + 832)   # 5152 B at R = 1080
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_MEASURED_COPY_GBS = 6290.0


def measured_traffic(n_envs, n_rays, cars, policy, steps):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (tools/traffic.sh: FETCH_SIZE and WRITE_SIZE in
    separate runs, KB -> bytes; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  The counters cannot
    be read from inside this process, so the per-env-step figure of the same configuration is scaled to this launch."""
    p = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if not os.path.exists(p):
        return None
    t = json.load(open(p))
    if (t.get("n_envs"), t.get("n_rays", 1080), t.get("cars", 1), t.get("policy", "fast")) != (n_envs, n_rays, cars, policy):
        return None
    return t["traffic_bytes_per_env_step"] * n_envs * steps


def _try(fn):
    try:
        fn()
        return None
    except Exception as exc:      # noqa: BLE001 - reported by the caller
        return exc


def cpu_baseline(track, n_rays, policy, cars, seed):
    """The CPU oracle ("port") on a bounded sample of the same workload, on this host's cores (rank 0, N = 1 only)."""
    from tests.helpers import load_oracle
    ora = load_oracle()
    threads = max(1, min(os.cpu_count() or 1, int(os.environ.get("FTGP_CPU_THREADS", "16"))))
    threads = min(threads, ora.dll.oracle_max_threads()) if threads > 1 else 1
    n_envs = 64 * threads
    with capi.Env(ora, track, n_envs=n_envs, cars_per_env=cars, n_rays=n_rays, spawn_mode=1, seed=seed) as o:
        ora.dll.oracle_set_threads(o.h, threads)
        t0 = time.perf_counter()
        o.rollout(policy, 20)                       # calibration: size the timed sample to ~15 s of CPU work
        rate = n_envs * 20 / (time.perf_counter() - t0)
        steps = int(max(50, min(5000, 15.0 * rate / n_envs)))
        t0 = time.perf_counter()
        o.rollout(policy, steps)
        dt = time.perf_counter() - t0
    return {"value": n_envs * steps / dt, "unit": "env-steps/s", "cores": threads, "kind": "port",
            "sample": f"{n_envs} envs x {steps} steps of the same workload (first {n_envs} envs of the batch), "
                      f"oracle/ftgp_oracle.c with OpenMP over envs, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--envs-per-gpu", type=int, default=4096)
    ap.add_argument("--rays", type=int, default=1080)
    ap.add_argument("--cars", type=int, default=1)
    ap.add_argument("--track", default="track")
    ap.add_argument("--policy", default="fast")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--repeats", type=int, default=1, help="timed launches of K steps (the best is reported in ms_per_step_best)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    dist = None
    if world > 1:
        import torch.distributed as dist   # rendezvous / barrier plumbing only; the data path is HIP + RCCL
        dist.init_process_group(backend="gloo")

    lib = capi.load()
    if lib.fn("device_count")() < 1:
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    track = load_track(args.track)
    seed = 1234
    # rank r owns envs [r * envs_per_gpu, (r + 1) * envs_per_gpu) of one world-sized batch (BASELINE.json configs[3] at N = 8)
    n_dev = lib.fn("device_count")()
    env = capi.Env(lib, track, n_envs=args.envs_per_gpu, cars_per_env=args.cars, n_rays=args.rays, spawn_mode=1,
                   seed=seed, device_id=local_rank % n_dev, env_base=rank * args.envs_per_gpu)
    collective = "none (1 rank)"
    gloo_gather = None
    if world > 1:
        import torch
        from ft_grandprix_amd import dist as ftdist
        collective = "rccl ncclAllGather (xGMI), side stream"
        ok = 1
        try:
            if world > n_dev:
                raise RuntimeError("more ranks than GPUs: RCCL refuses two ranks on one device")
            if os.environ.get("FTGP_BENCH_COLLECTIVE", "rccl") != "rccl":
                raise RuntimeError("FTGP_BENCH_COLLECTIVE requests the gloo gather")
            uid = [capi.comm_unique_id(lib) if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            # watchdog: a stalled RCCL bootstrap must not hang the scaling run
            import threading
            box = {}
            th = threading.Thread(target=lambda: box.setdefault("err", _try(lambda: env.comm_init(uid[0], rank, world))), daemon=True)
            th.start(); th.join(timeout=float(os.environ.get("FTGP_RCCL_INIT_TIMEOUT", "180")))
            if th.is_alive():
                raise RuntimeError("ncclCommInitRank did not return in time")
            if box.get("err") is not None:
                raise box["err"]
        except Exception as exc:   # rehearsal on a box with fewer GPUs than ranks, or an RCCL problem: keep going over gloo, and say so
            ok = 0
            print(f"[rank {rank}] RCCL communicator not used ({exc}); metrics go over the gloo gather", file=sys.stderr)
        flag = torch.tensor([ok]); dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag[0]) == 0:
            gloo_gather = ftdist.GlooGather()
            collective = "gloo all_gather (RCCL communicator not used, see stderr)"

    def gather():
        if gloo_gather is not None:
            return gloo_gather.all_gather(env.metrics_local())
        return env.metrics_allgather()

    def barrier():
        if dist is not None:
            dist.barrier()

    # warmup (untimed)
    env.rollout(args.policy, args.warmup)
    gather()
    env.last_kernel_ms()

    best_ms = None
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.repeats):
        env.rollout(args.policy, args.steps)     # EXACTLY K steps per launch
        metrics = gather()                       # the only collective; side stream
        kms = env.last_kernel_ms()               # HIP events on the launch stream; also synchronises
        best_ms = kms if best_ms is None else min(best_ms, kms)
    barrier()
    wall = (time.perf_counter() - t0) / args.repeats
    kernel_ms = kms if args.repeats == 1 else best_ms

    if dist is not None:
        import torch
        tmax = torch.tensor([wall, kernel_ms / 1e3], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        wall, kernel_s = float(tmax[0]), float(tmax[1])
    else:
        kernel_s = kernel_ms / 1e3

    if rank == 0:
        total_envs = args.envs_per_gpu * world
        value = total_envs * args.steps / wall
        bytes_per_launch = ALGO_BYTES_PER_ENV_STEP(args.rays, args.cars) * args.envs_per_gpu * args.steps
        achieved = bytes_per_launch / kernel_s / 1e9
        out = {
            "metric": "env-steps/sec (4096 envs, 1080-ray LiDAR) at 1/2/4/8 MI355X; HBM roofline %",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 rays + f64 state", "data": "synthetic",
            "config": {"workload": f"{args.envs_per_gpu} envs/GPU x {args.cars} car(s), {args.track} track blob, "
                                   f"{args.rays}-ray LiDAR, {args.policy} driver on device (BASELINE.json configs[2])",
                       "envs_per_gpu": args.envs_per_gpu, "n_rays": args.rays, "cars_per_env": args.cars,
                       "policy": args.policy, "steps_per_launch": args.steps, "parallelism": f"env-shard x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": measured_traffic(args.envs_per_gpu, args.rays, args.cars, args.policy, args.steps),
                         "kernel": env.kernel_name(), "kernel_ms_per_launch": kernel_s * 1e3,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "frac_of_measured_copy_6290": achieved / HBM_MEASURED_COPY_GBS,
                         "rays_per_s": args.envs_per_gpu * args.cars * args.rays * args.steps / kernel_s},
            "metrics_allgather": {"collective": collective, "ranks": int(metrics.shape[0]), "sum_laps": float(metrics[:, 2].sum()),
                                  "sum_steps": float(metrics[:, 0].sum())},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(track, args.rays, args.policy, args.cars, seed)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if gloo_gather is not None and world <= n_dev:
        os._exit(0)            # a stalled RCCL thread may still hold the handle: do not wait for it
    env.close()


if __name__ == "__main__":
    main()
