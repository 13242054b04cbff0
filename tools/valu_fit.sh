#!/bin/bash
# SQ_INSTS_VALU / SQ_INSTS_SALU per car-step for builds with different refill thresholds (cost model fit): tools/valu_fit.sh 16 32 48 64
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
for r in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -w -mllvm -amdgpu-atomic-optimizer-strategy=None -fno-slp-vectorize -DFTGP_REFILL=$r -o gpurun_out/libftgp_fit.so ft_grandprix_amd/csrc/ftgp_api.hip -ldl
  rm -rf gpurun_out/fit_$r
  FTGP_LIB=gpurun_out/libftgp_fit.so timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES --output-format csv -d gpurun_out/fit_$r -- python3 tools/prof_case.py 4096 1080 fast 300 > gpurun_out/fit_$r.log 2>&1 || { tail -3 gpurun_out/fit_$r.log; exit 1; }
  python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/fit_$r/**/*counter_collection.csv",recursive=True)[0]
rows=[x for x in csv.DictReader(open(f)) if "ftgp_step_kernel" in x["Kernel_Name"]]
last=max(int(x["Dispatch_Id"]) for x in rows); c={}
for x in rows:
    if int(x["Dispatch_Id"])==last: c[x["Counter_Name"]]=c.get(x["Counter_Name"],0)+float(x["Counter_Value"])
n=4096*300
print("refill $r:", " ".join(f"{k} {v/n:.1f}" for k,v in sorted(c.items())), open("gpurun_out/fit_$r.log").read().strip().splitlines()[-1])
PY
done
