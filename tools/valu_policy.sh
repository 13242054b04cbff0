#!/bin/bash
# SQ_INSTS_VALU of one K5 evaluation per car (ftgp_policy_kernel): tools/valu_policy.sh <n_cars> <n_rays> <policy>
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
rm -rf gpurun_out/vp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d gpurun_out/vp -- python3 tools/prof_policy.py $1 $2 $3 > gpurun_out/vp.log 2>&1 || { tail -3 gpurun_out/vp.log; exit 1; }
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/vp/**/*counter_collection.csv",recursive=True)[0]
rows=[x for x in csv.DictReader(open(f)) if "ftgp_policy_kernel" in x["Kernel_Name"]]
last=max(int(x["Dispatch_Id"]) for x in rows); c={}
for x in rows:
    if int(x["Dispatch_Id"])==last: c[x["Counter_Name"]]=c.get(x["Counter_Name"],0)+float(x["Counter_Value"])
print("K5 $3 R=$2 per car:", " ".join(f"{k} {v/$1:.1f}" for k,v in sorted(c.items())))
PY
