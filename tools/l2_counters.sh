#!/bin/bash
# L2 hit rate of the step kernel (own pass): tools/l2_counters.sh <tag>
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
tag=$1
rm -rf gpurun_out/l2_$tag
timeout -k 10 ${PMC_TIMEOUT:-150} rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum --output-format csv -d gpurun_out/l2_$tag -- python3 tools/prof_case.py 4096 1080 fast 500 > gpurun_out/l2_$tag.log 2>&1 || { tail -5 gpurun_out/l2_$tag.log; exit 1; }
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/l2_$tag/**/*counter_collection.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "ftgp_step_kernel" in r["Kernel_Name"]]
last=max(int(r["Dispatch_Id"]) for r in rows)
c={}
for r in rows:
    if int(r["Dispatch_Id"])==last: c[r["Counter_Name"]]=c.get(r["Counter_Name"],0)+float(r["Counter_Value"])
print(c, "L2 hit rate %.3f" % (c["TCC_HIT_sum"]/(c["TCC_HIT_sum"]+c["TCC_MISS_sum"])), "L2 requests per env-step %.0f" % (c["TCC_REQ_sum"]/(4096*500)))
PY
