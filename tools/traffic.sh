#!/bin/bash
# HBM traffic of the step kernel from PMC counters (separate passes, kernel-trace only) -> gpurun_out/traffic.json
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
STEPS=${1:-200}
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_$c -- python3 tools/prof_case.py 4096 1080 fast $STEPS > gpurun_out/pmc_$c.log 2>&1 || exit 1
done
python3 - <<PY
import csv,glob,json
out={"config":"4096 envs x 1080 rays, fast, $STEPS steps per launch","steps":$STEPS,"n_envs":4096,"n_rays":1080,"cars":1,"policy":"fast"}
for c in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob(f"gpurun_out/pmc_{c}/**/*counter_collection.csv",recursive=True)[0]
    rows=[r for r in csv.DictReader(open(f)) if "ftgp_step_kernel" in r["Kernel_Name"] and r["Counter_Name"]==c]
    last=max(int(r["Dispatch_Id"]) for r in rows)
    out[c+"_raw_KB"]=sum(float(r["Counter_Value"]) for r in rows if int(r["Dispatch_Id"])==last)
# MI355X_MICROARCH.md HBM section: counters are in KB; FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 (uncalibrated for other widths)
out["write_bytes_per_launch"]=out["WRITE_SIZE_raw_KB"]*1024
out["fetch_bytes_per_launch_raw"]=out["FETCH_SIZE_raw_KB"]*1024
out["fetch_bytes_per_launch_x2"]=out["FETCH_SIZE_raw_KB"]*2048
out["traffic_bytes_per_launch"]=out["write_bytes_per_launch"]+out["fetch_bytes_per_launch_x2"]
out["traffic_bytes_per_env_step"]=out["traffic_bytes_per_launch"]/(4096*$STEPS)
json.dump(out,open("gpurun_out/traffic.json","w"),indent=1); print(json.dumps(out))
PY
