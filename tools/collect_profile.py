#!/usr/bin/env python3
"""One GPU-box visit that produces the judged evidence for a build (run it through gpurun, then copy gpurun_out/profile/* to
profiles/roundN/):   python3 tools/collect_profile.py [steps] [envs] [policy] [cars] [track] [tag]   (default: the headline; a tag
other than "latest" names the files sq_<tag>.json, ... and leaves the files bench.py reads alone)
  bench.json                       python bench.py (the driver's line, with cpu_baseline)
  kernel_stats.csv                 rocprofv3 --kernel-trace --stats of the same bench command (average duration per kernel)
  sq_latest.json                   SQ counters of the step kernel, three --pmc passes (own runs, kernel-trace only)
  traffic_latest.json              FETCH_SIZE / WRITE_SIZE of the step kernel, separate --pmc passes (guide: KB; FETCH x2 on gfx950)
Every rocprofv3 command starts the python program directly (no shell / env hop) as a child of this orchestrator, which never
touches the GPU itself."""
import csv, glob, hashlib, json, os, re, shutil, subprocess, sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
os.chdir(ROOT)
OUT = "gpurun_out/profile"
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 500
ENVS = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
POLICY = sys.argv[3] if len(sys.argv) > 3 else "fast"
CARS = int(sys.argv[4]) if len(sys.argv) > 4 else 1
TRACK = sys.argv[5] if len(sys.argv) > 5 else "track"
TAG = sys.argv[6] if len(sys.argv) > 6 else "latest"
RAYS = 1080
os.environ.setdefault("TMPDIR", "/tmp")


sys.path.insert(0, os.path.join(ROOT, "tools"))
from evidence import sha, code_object_meta        # noqa: E402  (one definition of "the sources' hash", shared with bench.py's check)


def run(cmd, log, timeout=400):
    print("==", " ".join(cmd), flush=True)
    with open(log, "w") as f:
        rc = subprocess.run(cmd, stdout=f, stderr=subprocess.STDOUT, timeout=timeout).returncode
    if rc != 0:
        print(open(log).read()[-2000:]); raise SystemExit(f"{cmd[0]} failed ({rc})")


def pmc(tag, counters):
    d = f"{OUT}/raw_{tag}"
    shutil.rmtree(d, ignore_errors=True)
    run(["rocprofv3", "--kernel-trace", "--pmc", *counters, "--output-format", "csv", "-d", d, "--",
         "python3", "tools/prof_case.py", str(ENVS), str(RAYS), POLICY, str(STEPS), str(CARS), TRACK], f"{OUT}/raw_{tag}.log")
    rows = [r for r in csv.DictReader(open(glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0])) if "ftgp_step_kernel" in r["Kernel_Name"]]
    last = max(int(r["Dispatch_Id"]) for r in rows)
    c, meta = {}, {}
    for r in rows:
        if int(r["Dispatch_Id"]) == last:
            c[r["Counter_Name"]] = c.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            meta = {"kernel": r["Kernel_Name"], "grid": r.get("Grid_Size"), "workgroup": r.get("Workgroup_Size")}
    text = open(f"{OUT}/raw_{tag}.log").read()
    ms = [l for l in text.splitlines() if l.startswith("kernel ms")]
    meta["kernel_ms"] = float(ms[-1].split()[2]) if ms else None
    # registers / scratch from the code object's own metadata, the dynamic LDS from ftgp_create (FTGP_VERBOSE): rocprofv3's VGPR_Count and
    # LDS_Block_Size columns read 32 and 0 for this kernel (63 registers, ~73 KB of dynamic LDS)
    for k, v in KERNEL_META.items():
        if k in meta.get("kernel", ""):
            meta.update(v)
    lds = re.findall(r"(\d+) cars x (\d+) waves per workgroup, (\d+) B of LDS", text)
    if lds:
        meta["cars_per_workgroup"], meta["waves_per_workgroup"], meta["dynamic_lds_bytes"] = (int(x) for x in lds[-1])
    return c, meta


os.makedirs(OUT, exist_ok=True)
os.environ["FTGP_VERBOSE"] = "1"
KERNEL_META = code_object_meta()
base = {"config": f"{ENVS} envs x {CARS} car(s) x {RAYS} rays, {TRACK}, {POLICY}, {STEPS} steps per launch", "track": TRACK, "steps": STEPS, "n_envs": ENVS, "n_rays": RAYS, "cars": CARS,
        "policy": POLICY, "kernel_source_sha": sha()}

# 1. SQ counters (8 slots per pass)
sq = dict(base, counters={})
for tag, ctrs in (("a", ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_SALU", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY"]),
                  ("b", ["SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_THREAD_CYCLES_VALU", "SQ_WAVES"]),
                  ("c", ["GRBM_GUI_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_VMEM", "SQ_INST_CYCLES_VMEM"])):
    c, meta = pmc("sq_" + tag, ctrs)
    sq["counters"].update(c); sq.update(meta)
c = sq["counters"]; n = ENVS * CARS * STEPS
sq["derived"] = {"valu_insts_per_car_step": c["SQ_INSTS_VALU"] / n, "salu_insts_per_car_step": c["SQ_INSTS_SALU"] / n,
                 "lds_insts_per_car_step": c["SQ_INSTS_LDS"] / n, "vmem_rd_insts_per_car_step": c["SQ_INSTS_VMEM_RD"] / n,
                 # the vector pipe's occupancy lies between these two: every instruction priced at the cheapest (v_add_u32, 2.28 SIMD-cycles) / the
                 # dearest common kind (v_cmp, 4.32) of tools/issue_calib.sh; SQ_ACTIVE_INST_VALU ticks once per instruction and is no cycle count
                 "valu_pipe_occupancy_lower": c["SQ_INSTS_VALU"] * 2.28 / (1024 * c["GRBM_GUI_ACTIVE"] / 8.0),
                 "valu_pipe_occupancy_upper": min(1.0, c["SQ_INSTS_VALU"] * 4.32 / (1024 * c["GRBM_GUI_ACTIVE"] / 8.0)),
                 "wait_any_frac_of_wave_cycles": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], "wait_inst_frac_of_wave_cycles": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
                 "shader_clock_ghz": c["GRBM_GUI_ACTIVE"] / 8.0 / (sq["kernel_ms"] * 1e6)}
json.dump(sq, open(f"{OUT}/sq_{TAG}.json", "w"), indent=1)

# 2. HBM traffic (separate passes; FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2)
tr = dict(base)
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    c, meta = pmc(name, [name])
    tr[name + "_raw_KB"] = c[name]; tr.update(meta)
tr["write_bytes_per_launch"] = tr["WRITE_SIZE_raw_KB"] * 1024
tr["fetch_bytes_per_launch_x2"] = tr["FETCH_SIZE_raw_KB"] * 2048        # MI355X_MICROARCH.md: FETCH_SIZE reports half of the bytes on gfx950
tr["traffic_bytes_per_launch"] = tr["write_bytes_per_launch"] + tr["fetch_bytes_per_launch_x2"]
tr["traffic_bytes_per_env_step"] = tr["traffic_bytes_per_launch"] / (ENVS * STEPS)
tr["write_bytes_per_env_step"] = tr["write_bytes_per_launch"] / (ENVS * STEPS)
tr["algorithmic_bytes_per_env_step"] = CARS * (4 * RAYS + 832)
json.dump(tr, open(f"{OUT}/traffic_{TAG}.json", "w"), indent=1)

# 3. the bench line reads these two files: give it the ones of this very visit (same sources, so nothing is "stale")
PROFILES = "profiles/round5"
os.makedirs(PROFILES, exist_ok=True)
for name in (f"sq_{TAG}.json", f"traffic_{TAG}.json"):
    shutil.copy(f"{OUT}/{name}", f"{PROFILES}/{name}")

# 4. the bench line + kernel stats of the same command
cfg = ["--envs-per-gpu", str(ENVS), "--policy", POLICY, "--cars", str(CARS), "--track", TRACK]
SUFFIX = "" if TAG == "latest" else "_" + TAG
run(["python3", "bench.py", "--steps", str(STEPS), "--warmup", "50", *cfg] + ([] if TAG == "latest" else ["--no-cpu-baseline"]), f"{OUT}/bench{SUFFIX}.log", 600)
line = [l for l in open(f"{OUT}/bench{SUFFIX}.log").read().splitlines() if l.startswith("{")][-1]
open(f"{OUT}/bench{SUFFIX}.json", "w").write(line + "\n")
open(f"{OUT}/bench{SUFFIX}.json.sha", "w").write(sha() + "\n")
shutil.rmtree(f"{OUT}/raw_stats", ignore_errors=True)
run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", f"{OUT}/raw_stats", "--",
     "python3", "bench.py", "--steps", str(STEPS), "--warmup", "50", "--no-cpu-baseline", *cfg], f"{OUT}/raw_stats.log", 600)
shutil.copy(glob.glob(f"{OUT}/raw_stats/**/*kernel_stats.csv", recursive=True)[0], f"{OUT}/kernel_stats{SUFFIX}.csv")
open(f"{OUT}/kernel_stats{SUFFIX}.csv.sha", "w").write(sha() + "\n")

for d in glob.glob(f"{OUT}/raw_*"):
    if os.path.isdir(d):
        shutil.rmtree(d)
print(json.dumps(sq["derived"], indent=1)); print("traffic per env-step", tr["traffic_bytes_per_env_step"], "writes", tr["write_bytes_per_env_step"]); print(line[:300])
