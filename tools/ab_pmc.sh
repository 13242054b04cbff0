#!/bin/bash
# A/B with counters on one box: tools/ab_pmc.sh "<tag>:<-D flags>" ...   (CASE="envs rays policy steps [cars] [track]" to change the workload)
# Builds gpurun_out/libftgp_<tag>.so for each spec, then one --pmc pass per library (kernel-trace only) on the same rollout and prints,
# per car-step: vector / scalar / LDS instructions, busy quad-cycles, and the launch's shader cycles (GRBM_GUI_ACTIVE / 8) -- cycles
# compare across boxes, times do not.
set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
CASE=${CASE:-"4096 1080 fast 300"}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -w -mllvm -amdgpu-atomic-optimizer-strategy=None -fno-slp-vectorize"
echo "# kernel_source_sha=$(python3 tools/evidence.py sha) $(date '+%Y-%m-%d %H:%M:%S') ab_pmc CASE=$CASE"
for spec in "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc $FLAGS $flags -o gpurun_out/libftgp_$tag.so ft_grandprix_amd/csrc/ftgp_api.hip -ldl || exit 1
  rm -rf gpurun_out/abp_$tag
  FTGP_LIB=gpurun_out/libftgp_$tag.so timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE \
      --output-format csv -d gpurun_out/abp_$tag -- python3 tools/prof_case.py $CASE > gpurun_out/abp_$tag.log 2>&1 || { echo "$tag: pass failed"; tail -3 gpurun_out/abp_$tag.log; exit 1; }
  python3 - "$tag" $CASE <<'PY'
import csv, glob, sys
tag, envs, rays, policy, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
cars = int(sys.argv[6]) if len(sys.argv) > 6 else 1
f = glob.glob(f"gpurun_out/abp_{tag}/**/*counter_collection.csv", recursive=True)[0]
rows = [x for x in csv.DictReader(open(f)) if "ftgp_step_kernel" in x["Kernel_Name"]]
last = max(int(x["Dispatch_Id"]) for x in rows); c = {}
for x in rows:
    if int(x["Dispatch_Id"]) == last: c[x["Counter_Name"]] = c.get(x["Counter_Name"], 0) + float(x["Counter_Value"])
n = envs * cars * steps
cyc = c["GRBM_GUI_ACTIVE"] / 8
ms = [l for l in open(f"gpurun_out/abp_{tag}.log").read().splitlines() if l.startswith("kernel ms")][-1].split()[2]
print(f"{tag:>12s}: VALU {c['SQ_INSTS_VALU']/n:7.1f}  SALU {c['SQ_INSTS_SALU']/n:7.1f}  LDS {c['SQ_INSTS_LDS']/n:6.1f} per car-step | busy {c['SQ_ACTIVE_INST_VALU']*4/(1024*cyc):.3f} "
      f"issue {c['SQ_INSTS_VALU']*2/(1024*cyc):.3f} | wait_any {c['SQ_WAIT_ANY']/c['SQ_WAVE_CYCLES']:.3f} wait_inst {c['SQ_WAIT_INST_ANY']/c['SQ_WAVE_CYCLES']:.3f} | "
      f"{cyc/steps/1e3:7.2f} k cycles/step  ({float(ms)*1e3/steps:.2f} us/step profiled, {cyc/float(ms)/1e6:.2f} GHz)", flush=True)
PY
done
