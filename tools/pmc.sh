#!/bin/bash
# PMC pass (own run, no tracing besides kernel-trace): tools/pmc.sh <tag> <counters...> -- <prof_case args>
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
tag=$1; shift
ctrs=(); while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
rm -rf gpurun_out/pmc_$tag
timeout -k 10 ${PMC_TIMEOUT:-150} rocprofv3 --kernel-trace --pmc "${ctrs[@]}" --output-format csv -d gpurun_out/pmc_$tag -- python3 tools/prof_case.py "$@" > gpurun_out/pmc_$tag.log 2>&1
echo "pmc $tag rc=$?"
python3 - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/pmc_$tag/**/*counter_collection.csv",recursive=True)
if not f: print("no counter csv"); raise SystemExit
rows=list(csv.DictReader(open(f[0])))
agg=collections.OrderedDict()
for r in rows:
    if "ftgp_step_kernel" not in r["Kernel_Name"]: continue
    k=(r["Dispatch_Id"],r["Counter_Name"]); agg[k]=agg.get(k,0)+float(r["Counter_Value"])
last=max(int(k[0]) for k in agg) if agg else None
for (d,c),v in agg.items():
    if int(d)==last: print(c, v)
PY
