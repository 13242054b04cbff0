#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the step kernel for one configuration (separate --pmc passes, kernel-trace only):
#   tools/traffic_case.sh <envs> <rays> <policy> <steps> [cars] [track]      (FTGP_LIB=... for a variant library)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
E=$1; R=$2; P=$3; S=$4; C=${5:-1}; T=${6:-track}
echo "# kernel_source_sha=$(python3 tools/evidence.py sha) $(date '+%Y-%m-%d %H:%M:%S') traffic_case $* lib=${FTGP_LIB:-product}"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/tc_$c
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/tc_$c -- python3 tools/prof_case.py $E $R $P $S $C $T > gpurun_out/tc_$c.log 2>&1 || { tail -3 gpurun_out/tc_$c.log; exit 1; }
done
python3 - $E $R $P $S $C <<'PY'
import csv, glob, sys
E, R, P, S, C = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
v = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/tc_{c}/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "ftgp_step_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c]
    last = max(int(r["Dispatch_Id"]) for r in rows)
    v[c] = sum(float(r["Counter_Value"]) for r in rows if int(r["Dispatch_Id"]) == last) * 1024
ms = [l for l in open("gpurun_out/tc_WRITE_SIZE.log").read().splitlines() if l.startswith("kernel ms")][-1].split()[2]
n = E * S
fetch2, wr, algo = 2 * v["FETCH_SIZE"] / n, v["WRITE_SIZE"] / n, C * (4 * R + 832)
print(f"{E} envs x {C} cars x {R} rays {P}, {S} steps: per env-step: fetch x2 {fetch2:.0f} B, write {wr:.0f} B, total {fetch2 + wr:.0f} B = {(fetch2 + wr) / algo:.2f} x algorithmic ({algo} B); "
      f"{(fetch2 + wr) * n / (float(ms) * 1e-3) / 1e9:.0f} GB/s at the fabric; {float(ms) * 1e3 / S:.2f} us/step profiled")
PY
