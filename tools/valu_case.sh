#!/bin/bash
# SQ_INSTS_VALU / SALU / wave cycles per car-step of the product library for one configuration: tools/valu_case.sh <envs> <rays> <policy> [steps]
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
E=$1; R=$2; P=$3; S=${4:-300}
rm -rf gpurun_out/vc
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d gpurun_out/vc -- python3 tools/prof_case.py $E $R $P $S > gpurun_out/vc.log 2>&1 || { tail -3 gpurun_out/vc.log; exit 1; }
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/vc/**/*counter_collection.csv",recursive=True)[0]
rows=[x for x in csv.DictReader(open(f)) if "ftgp_step_kernel" in x["Kernel_Name"]]
last=max(int(x["Dispatch_Id"]) for x in rows); c={}
for x in rows:
    if int(x["Dispatch_Id"])==last: c[x["Counter_Name"]]=c.get(x["Counter_Name"],0)+float(x["Counter_Value"])
n=$E*$S
print("$E envs $R rays $P:", " ".join(f"{k} {v/n:.1f}" for k,v in sorted(c.items())), [l for l in open("gpurun_out/vc.log").read().splitlines() if l.startswith("kernel ms")][-1])
PY
