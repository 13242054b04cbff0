#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec at 4096 envs x 1080-ray LiDAR per GPU (BASELINE.json configs[2]).

A "step" is one pass of the hot path over the whole batch: driver (fast, on device) -> LiDAR sweep written
to HBM -> integrate -> lap progress, for every env.  All inputs are resident in HBM before the timed region;
the K timed steps run as ONE persistent launch (the state stays in LDS between steps).  A launch of few steps is far shorter than
the clock differences between boxes, so `--repeats` such launches are timed one by one (default: enough of them for about 20 ms of
kernel time) and the MEDIAN launch is what `value` and `ms_per_step` report (`repeats`, best and worst are in the line).

    python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU.  Launched by `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py
--gpus N` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env), or, when no launcher set WORLD_SIZE, bench.py starts
its own N ranks.  The host path is torch-free: rendezvous, barrier and the max-over-ranks timing go over
ft_grandprix_amd.dist.Rendezvous (TCP on MASTER_ADDR); envs shard across ranks with no data-path collective (weak scaling:
4096 envs per GPU); the only exchange is the end-of-launch metrics all-gather over RCCL, issued on a side stream behind launch k
and collected while launch k + 1 runs (ftgp_metrics_allgather_begin / _end, ft_grandprix_amd.dist.run_timed).
Rank 0 prints ONE JSON line.
"""
import argparse
import glob
import hashlib
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from ft_grandprix_amd import capi  # noqa: E402
from ft_grandprix_amd import dist as ftdist  # noqa: E402
from ft_grandprix_amd.track import load_track  # noqa: E402

ALGO_BYTES_PER_ENV_STEP = lambda n_rays, cars: cars * (4 * n_rays + 832)   # 5152 B at R = 1080
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_MEASURED_COPY_GBS = 6290.0
N_SIMD = 256 * 4               # MI355X: 256 CUs x 4 SIMDs
# What a wave64 vector instruction costs a SIMD at 8 waves per SIMD, measured with tools/issue_calib.sh (profiles/round5/issue_calib.log):
# SIMD-cycles per instruction of the kind.  SQ_ACTIVE_INST_VALU is NOT a cycle count: it ticks once per vector instruction (twice for a
# transcendental), so round 4's "busy" figure (that counter x 4) was an instruction count in disguise and could exceed 1.
VALU_CYCLES_CHEAPEST = 2.28    # v_add_u32 (and logic / shift / move): no vector instruction issues faster
VALU_CYCLES_DEAREST = 4.32     # v_cmp / v_cndmask / conversions / binary64 (v_fma_f32: 3.56); only v_rcp_f32 (8.1, two per ray set-up) costs more
PROFILE_DIRS = [os.path.join(ROOT, "profiles", d) for d in ("round5", "round4", "round3", "round2")]     # newest first


def kernel_source_sha():
    """Identity of the kernel sources a committed counter file was measured on."""
    h = hashlib.sha256()
    for rel in ("ft_grandprix_amd/csrc/ftgp_kernels.hip", "ft_grandprix_amd/csrc/ftgp_march.h", "ft_grandprix_amd/csrc/ftgp_device.h",
                "ft_grandprix_amd/csrc/ftgp_api.hip", "ft_grandprix_amd/csrc/diag/ftgp_diag.inc", "include/ftgp.h"):
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def committed_counters(prefix, n_envs, n_rays, cars, policy, track="track"):
    """The newest counter summary (`<prefix>_*.json`) committed under profiles/roundN by tools/collect_profile.py (rocprofv3 --pmc,
    separate passes) that was measured on this configuration.  The counters cannot be read from inside this process; `stale`
    says whether the kernel sources have changed since."""
    for d in PROFILE_DIRS:
        for p in sorted(glob.glob(os.path.join(d, prefix + "_*.json"))):
            t = json.load(open(p))
            have = (t.get("n_envs"), t.get("n_rays", 1080), t.get("cars", 1), t.get("policy", "fast"), t.get("track", "track"))
            if have != (n_envs, n_rays, cars, policy, track):
                continue
            t["stale"] = t.get("kernel_source_sha") != kernel_source_sha()
            t["file"] = os.path.relpath(p, ROOT)
            return t
    return None


def baseline_config(args, world):
    """Which BASELINE.json config the flags amount to."""
    key = (args.envs_per_gpu, args.cars, args.track, args.rays, args.policy)
    if key == (4096, 1, "track", 1080, "fast"):
        return "BASELINE.json configs[2], the headline" if world == 1 else f"BASELINE.json configs[2] per GPU, x{world} GPUs"
    if key == (4096, 1, "track", 1080, "random"):
        return "BASELINE.json configs[3]: 32768 envs over 8 GPUs" if world == 8 else f"BASELINE.json configs[3] per-GPU shard, x{world} GPUs"
    if key == (1024, 1, "circle", 1080, "nidc"):
        return "BASELINE.json configs[1]"
    if key[:4] == (4096, 4, "track", 1080):
        return "BASELINE.json configs[4]: 4-car worlds with inter-vehicle rays"
    return "not a BASELINE.json config"


def cpu_share():
    """CPUs this process may actually use: the affinity mask, cut by a cgroup CPU quota if there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(track, n_rays, policy, cars, seed, budget_s=14.0, lidar_mode="rangefinder"):
    """The CPU oracle ("port") on bounded samples of the same workload on this host (rank 0, N = 1 only), as SURVEY.md 8d
    asks: ONE thread, and all the host cores this process may use (OpenMP over envs) -- plus every logical CPU of the host
    when the process's share is smaller than that.  About `budget_s` seconds of CPU work in all."""
    from tests.helpers import load_oracle
    ora = load_oracle()
    host_cpus = os.cpu_count() or 1
    share = cpu_share()
    omp_max = max(1, ora.dll.oracle_max_threads())
    legs = [1]
    for t in (min(share, omp_max), min(host_cpus, omp_max)):
        if t > 1 and t not in legs:
            legs.append(t)
    per_leg = budget_s / len(legs)
    runs = []
    for threads in legs:
        n_envs = max(8, min(4096, 8 * threads))                    # the first n_envs envs of the batch; >= 8 per thread
        with capi.Env(ora, track, n_envs=n_envs, cars_per_env=cars, n_rays=n_rays, spawn_mode=1, seed=seed, lidar_mode=lidar_mode) as o:
            ora.dll.oracle_set_threads(o.h, threads)
            t0 = time.perf_counter()
            o.rollout(policy, 10)                                    # calibration
            rate = n_envs * 10 / (time.perf_counter() - t0)
            steps = int(max(10, min(5000, per_leg * 0.85 * rate / n_envs)))
            t0 = time.perf_counter()
            o.rollout(policy, steps)
            dt = time.perf_counter() - t0
        runs.append({"threads": threads, "value": n_envs * steps / dt, "n_envs": n_envs, "steps": steps, "seconds": round(dt, 2)})
    best = max(runs, key=lambda r: r["value"])
    return {"value": best["value"], "unit": "env-steps/s", "cores": best["threads"], "kind": "port",
            "single_thread": runs[0]["value"], "host_cpus": host_cpus, "cpu_share_of_this_process": share, "runs": runs,
            "sample": "; ".join(f"{r['threads']} thread(s): first {r['n_envs']} envs of the batch x {r['steps']} steps in {r['seconds']} s" for r in runs)
                      + f" -- oracle/ftgp_oracle.c (the CPU restatement, OpenMP over envs); host has {host_cpus} logical CPUs, this process may use {share}"}


def launch_ranks(n, argv, script=None, poll_s=0.2):
    """No launcher set WORLD_SIZE: start the N ranks ourselves (one child per GPU).  Returns 0 only if every rank exits 0; as soon
    as one rank fails the others are stopped (they would wait for it in the rendezvous until their timeout) and its code is
    returned."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   FTGP_JOB_TOKEN=f"bench-{os.getpid()}")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(script or __file__)] + argv, env=env))
    worst = 0
    try:
        while any(p.poll() is None for p in procs):
            bad = [p.returncode for p in procs if p.poll() not in (None, 0)]
            if bad:
                worst = max(abs(rc) for rc in bad)
                break
            time.sleep(poll_s)
        bad = [p.returncode for p in procs if p.poll() not in (None, 0)]
        if bad:
            worst = max(worst, max(abs(rc) for rc in bad))
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    return worst


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
    ap.add_argument("--repeats", type=int, default=0, help="timed launches of K steps each, timed one by one; the median is reported "
                                                           "(0 = as many as make about 20 ms of kernel time, at most 31)")
    ap.add_argument("--lidar", default="rangefinder", choices=("rangefinder", "fakelidar"))
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))        # never fall through to a 1-rank run
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with `python -m torch.distributed.run --nnodes=1 "
                         f"--nproc-per-node {args.gpus} --master-addr 127.0.0.1 --master-port P bench.py --gpus {args.gpus} ...`")

    rdzv = ftdist.Rendezvous.from_env() if world > 1 else None
    lib = capi.load()
    n_dev = lib.fn("device_count")()
    if n_dev < 1:
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    track = load_track(args.track)
    seed = 1234
    # rank r owns envs [r * envs_per_gpu, (r + 1) * envs_per_gpu) of one world-sized batch (BASELINE.json configs[3] at N = 8)
    env = capi.Env(lib, track, n_envs=args.envs_per_gpu, cars_per_env=args.cars, n_rays=args.rays, spawn_mode=1,
                   seed=seed, device_id=local_rank % n_dev, env_base=rank * args.envs_per_gpu, lidar_mode=args.lidar)
    collective = "none (1 rank)"
    host_gather = False
    if world > 1:
        want = os.environ.get("FTGP_BENCH_COLLECTIVE", "rccl")
        if want == "host":            # explicit rehearsal mode (e.g. more ranks than GPUs on one box): labelled as such
            host_gather = True
            collective = "host TCP gather (FTGP_BENCH_COLLECTIVE=host; RCCL not requested)"
        else:
            collective = "rccl ncclAllGather (xGMI), side stream"
            err = None
            try:
                uid = ftdist.exchange_unique_id(rdzv, lambda: capi.comm_unique_id(lib))   # symmetric: a rank-0 failure reaches every rank
                box = {}
                def init():
                    try:
                        env.comm_init(uid, rank, world)
                    except Exception as exc:      # noqa: BLE001 - reported below
                        box["err"] = exc
                th = threading.Thread(target=init, daemon=True)   # watchdog: a stalled RCCL bootstrap must not hang the scaling run
                th.start(); th.join(timeout=float(os.environ.get("FTGP_RCCL_INIT_TIMEOUT", "180")))
                if th.is_alive():
                    err = "ncclCommInitRank did not return in time"
                elif "err" in box:
                    err = str(box["err"])
            except Exception as exc:              # noqa: BLE001
                err = str(exc)
            oks = rdzv.allgather_bytes(b"\x01" if err is None else b"\x00" + err.encode()[:300])
            bad = [(r, o[1:].decode(errors="replace")) for r, o in enumerate(oks) if o[:1] != b"\x01"]
            if bad:
                # RCCL was requested and is not usable.  Every rank knows (the statuses were exchanged), so all take the same
                # branch.  A bootstrap that STALLED may still hold the handle: stop, loudly.  A bootstrap that RETURNED an
                # error leaves the handle usable: the 64-byte metrics record then travels over the host rendezvous instead
                # -- said on stderr and in the JSON line, never silently; the hot path (no collective) is unaffected.
                if rank == 0:
                    print(f"bench.py: the RCCL communicator could not be set up: {bad}", file=sys.stderr)
                if any("did not return in time" in why for _, why in bad) or os.environ.get("FTGP_BENCH_RCCL_REQUIRED"):
                    rdzv.close()
                    os._exit(3)       # do not wait for a stalled RCCL thread
                host_gather = True
                collective = ("host TCP gather -- FALLBACK: ncclCommInitRank failed on rank(s) "
                              + ", ".join(f"{r} ({why[:120]})" for r, why in bad))
                if rank == 0:
                    print("bench.py: falling back to the host TCP gather for the metrics record (set FTGP_BENCH_RCCL_REQUIRED=1 to "
                          "make this fatal)", file=sys.stderr)

    # The exchange of launch k's record overlaps launch k + 1 (ftdist.run_timed).  Over RCCL it rides the side stream; over the
    # host rendezvous a worker thread gathers on a connection set of its own (the main one carries the barriers meanwhile).
    rdzv_x = None
    if host_gather:
        rdzv_x = ftdist.Rendezvous.from_env(channel="x")
        exchange = ftdist.HostExchange(env, rdzv_x)
    else:
        exchange = ftdist.DeviceExchange(env)

    def barrier():
        if rdzv is not None:
            rdzv.barrier()

    repeats = args.repeats
    if repeats <= 0:           # a launch of few steps is shorter than anything a wall clock compares across boxes: time several
        repeats = max(1, min(31, int(20.0 / max(args.steps * 0.03 * args.cars, 1e-3) + 0.999)))
        repeats += 1 - repeats % 2                                   # odd: the median is a launch that happened

    # warmup (untimed); its exchange is begun here and collected beside the first timed launch
    env.rollout(args.policy, args.warmup)
    env.last_kernel_ms()
    exchange.begin()

    timed = ftdist.run_timed(env, args.policy, args.steps, repeats, exchange, barrier)      # EXACTLY K steps per launch
    metrics = timed["records"]
    walls, kmss = np.asarray(timed["wall_s"]), np.asarray(timed["kernel_ms"])
    barrier()
    # what every rank saw, before the max over ranks is taken: a scaling curve that bends can then be read -- one slow rank, a slow
    # exchange, or all ranks alike (DESIGN.md section 7 says what the design predicts: flat)
    ends = np.asarray(timed["exchange_end_s"] or [0.0])
    mine = np.array([float(np.median(kmss)), float(kmss.min()), float(kmss.max()), float(np.median(walls)) * 1e3, float(np.median(ends)) * 1e3, float(ends.max()) * 1e3])
    per_rank = rdzv.all_gather(mine) if rdzv is not None else mine[None, :]
    if rdzv is not None:
        both = rdzv.max(np.concatenate([walls, kmss]))               # per launch: the max over ranks
        walls, kmss = both[:repeats], both[repeats:]
    wall = float(np.median(walls))                                   # the median launch (max over ranks taken first)
    kernel_s = float(np.mean(kmss)) / 1e3                            # average launch duration of the kernel over the timed region (HIP events)

    if rank == 0:
        total_envs = args.envs_per_gpu * world
        value = total_envs * args.steps / wall
        bytes_per_launch = ALGO_BYTES_PER_ENV_STEP(args.rays, args.cars) * args.envs_per_gpu * args.steps
        achieved = bytes_per_launch / kernel_s / 1e9
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": None, "kernel": env.kernel_name(), "kernel_ms_per_launch": kernel_s * 1e3,
                "algorithmic_bytes_per_launch": bytes_per_launch, "frac_of_measured_copy_6290": achieved / HBM_MEASURED_COPY_GBS,
                "rays_per_s": args.envs_per_gpu * args.cars * args.rays * args.steps / kernel_s,
                # `frac` (= frac_hbm) is the figure SURVEY.md 8d prescribes for this path: its algorithmic bytes against the HBM peak.
                # `bound` says what the committed counters show to be the limit (set below); without counters it stays "hbm".
                "frac_hbm": achieved / HBM_PEAK_GBS}
        rangefinder = args.lidar == "rangefinder"         # the committed counters were taken in that mode
        tr = committed_counters("traffic", args.envs_per_gpu, args.rays, args.cars, args.policy, args.track) if rangefinder else None
        if tr is not None:
            # HBM bytes per launch = the committed PMC figure per env-step (FETCH_SIZE x2 + WRITE_SIZE, separate passes) scaled to this launch
            roof["traffic"] = tr["traffic_bytes_per_env_step"] * args.envs_per_gpu * args.steps
            roof["measured_traffic_gbs"] = roof["traffic"] / kernel_s / 1e9          # the same launch priced at its PMC bytes instead of the algorithmic ones
            roof["traffic_over_algorithmic"] = roof["traffic"] / bytes_per_launch
            roof["traffic_source"] = {"file": tr["file"], "measured_in_this_run": False,
                                      "kernel_source_sha": tr.get("kernel_source_sha"), "stale": tr["stale"],
                                      "bytes_per_env_step": tr["traffic_bytes_per_env_step"]}
        sq = committed_counters("sq", args.envs_per_gpu, args.rays, args.cars, args.policy, args.track) if rangefinder else None
        if sq is not None:
            c, steps_c = sq["counters"], sq["steps"]
            valu_per_car_step = c["SQ_INSTS_VALU"] / (sq["n_envs"] * sq.get("cars", 1) * steps_c)
            cycles = c["GRBM_GUI_ACTIVE"] / 8.0                         # the counter sums the 8 XCDs
            per_simd = c["SQ_INSTS_VALU"] / (N_SIMD * cycles)            # vector instructions per SIMD-cycle
            roof["valu"] = {"insts_per_car_step": valu_per_car_step,
                            # the vector pipe's occupancy lies between these two (every instruction priced at the cheapest / the dearest kind of
                            # the calibration); both are fractions of the SIMD-cycles and the upper one is capped at 1
                            "pipe_occupancy_lower": per_simd * VALU_CYCLES_CHEAPEST,
                            "pipe_occupancy_upper": min(1.0, per_simd * VALU_CYCLES_DEAREST),
                            "waves_waiting_frac": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"] if "SQ_WAIT_ANY" in c and c.get("SQ_WAVE_CYCLES") else None,
                            "waves_waiting_for_issue_frac": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"] if "SQ_WAIT_INST_ANY" in c and c.get("SQ_WAVE_CYCLES") else None,
                            "calibration": "profiles/round5/issue_calib.log",
                            "shader_clock_ghz": cycles / (sq["kernel_ms"] * 1e6) if sq.get("kernel_ms") else None,
                            "source": {"file": sq["file"], "measured_in_this_run": False,
                                       "kernel_source_sha": sq.get("kernel_source_sha"), "stale": sq["stale"]}}
        if "valu" in roof:
            # the vector pipes against their own peak, beside the HBM figure.  `bound` is "valu-issue" when even the LOWER bound of the pipe's
            # occupancy is above half of the SIMD-cycles while the fabric carries a small part of the HBM peak, and the waves do queue for issue
            v = roof["valu"]
            roof["frac_valu_pipe_lower"], roof["frac_valu_pipe_upper"] = v["pipe_occupancy_lower"], v["pipe_occupancy_upper"]
            fabric = roof.get("measured_traffic_gbs", achieved) / HBM_PEAK_GBS
            if v["pipe_occupancy_lower"] >= 0.5 and v["pipe_occupancy_lower"] > 2.0 * fabric:
                roof["bound"] = "valu-issue"
            wait, wait_any = v.get("waves_waiting_for_issue_frac"), v.get("waves_waiting_frac")
            # (round 5 measured both sides: 7 % fewer instructions bought 2.4 % of the cycles, the march's next look-up issued ahead of its near-boundary
            # test 5.4 % -- DESIGN.md section 6: the pipe's occupancy is near the LOWER price, the rest of a wave's life is the dependent chain of an iteration)
            roof["limiter"] = (f"vector-instruction issue and the dependent chain of a march iteration: {v['insts_per_car_step']:.0f} wave64 vector instructions per car-step occupy the SIMDs' vector pipes "
                               f"{100 * v['pipe_occupancy_lower']:.0f} - {100 * v['pipe_occupancy_upper']:.0f} % of the time (each priced at {VALU_CYCLES_CHEAPEST} - "
                               f"{VALU_CYCLES_DEAREST} cycles, tools/issue_calib.sh)" + (f", waves wait {100 * wait_any:.0f} % of their life" if wait_any is not None else "")
                               + (f", {100 * wait:.0f} % of it for an issue slot" if wait is not None else "")
                               + f"; the fabric traffic is {100 * fabric:.0f} % of the HBM peak"
                               if roof["bound"] == "valu-issue" else
                               f"HBM / fabric traffic at {100 * fabric:.0f} % of the peak; vector pipes occupied {100 * v['pipe_occupancy_lower']:.0f} - {100 * v['pipe_occupancy_upper']:.0f} %")
        out = {
            "metric": "env-steps/sec (4096 envs, 1080-ray LiDAR) at 1/2/4/8 MI355X; HBM roofline %",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall * 1e3 / args.steps, "repeats": repeats,
            "ms_per_step_best": float(walls.min()) * 1e3 / args.steps, "ms_per_step_worst": float(walls.max()) * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 rays + f64 state", "data": "synthetic",
            "config": {"workload": f"{args.envs_per_gpu} envs/GPU x {args.cars} car(s), {args.track} track blob, "
                                   f"{args.rays}-ray LiDAR, {args.policy} driver on device ({baseline_config(args, world)})"
                                   + ("" if args.lidar == "rangefinder" else "; FAKELIDAR mode (raycast.py:5-21 as the K2 of the loop), not a BASELINE config"),
                       "envs_per_gpu": args.envs_per_gpu, "n_rays": args.rays, "cars_per_env": args.cars,
                       "policy": args.policy, "steps_per_launch": args.steps, "parallelism": f"env-shard x{world}"},
            "roofline": roof,
            "metrics_allgather": {"collective": collective, "ranks": int(metrics.shape[0]), "sum_laps": float(metrics[:, 2].sum()),
                                  "sum_steps": float(metrics[:, 0].sum()),
                                  # how long collecting an exchange (end()) held up a launch's host side: median and worst, worst rank
                                  "end_wait_ms_median": float(per_rank[:, 4].max()), "end_wait_ms_max": float(per_rank[:, 5].max())},
            # per rank, before the max over ranks: kernel time of a launch (HIP events: median / min / max over the timed launches) and the
            # median wall time -- what explains a scaling curve that is not flat
            "per_rank": {"kernel_ms_median": [float(x) for x in per_rank[:, 0]], "kernel_ms_min": [float(x) for x in per_rank[:, 1]],
                         "kernel_ms_max": [float(x) for x in per_rank[:, 2]], "wall_ms_median": [float(x) for x in per_rank[:, 3]],
                         "kernel_ms_spread_over_ranks": float(per_rank[:, 0].max() / max(per_rank[:, 0].min(), 1e-12))},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(track, args.rays, args.policy, args.cars, seed, lidar_mode=args.lidar)
        print(json.dumps(out), flush=True)
    exchange.close()
    if rdzv_x is not None:
        rdzv_x.close()
    if rdzv is not None:
        rdzv.barrier()
        rdzv.close()
    env.close()


if __name__ == "__main__":
    main()
