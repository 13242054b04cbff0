#!/bin/bash
# SQ counter passes for the headline step kernel (own runs, kernel-trace only): tools/sq_counters.sh <tag> [steps] [envs] [policy]
# -> gpurun_out/sq_<tag>.json  (last dispatch of ftgp_step_kernel, summed over XCDs/SEs as rocprofv3 reports them)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
tag=$1; STEPS=${2:-500}; ENVS=${3:-4096}; POLICY=${4:-fast}
pass() {  # name counters...
  local n=$1; shift
  rm -rf gpurun_out/sq_${tag}_$n
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/sq_${tag}_$n -- python3 tools/prof_case.py $ENVS 1080 $POLICY $STEPS > gpurun_out/sq_${tag}_$n.log 2>&1 || { echo "pass $n failed"; tail -5 gpurun_out/sq_${tag}_$n.log; exit 1; }
}
pass a SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
pass b SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_WAVES
pass c GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU
python3 - <<PY
import csv,glob,json,collections
out={"config":"$ENVS envs x 1080 rays, $POLICY, $STEPS steps per launch","steps":$STEPS,"n_envs":$ENVS,"counters":{}}
for n in "abc":
    f=glob.glob(f"gpurun_out/sq_${tag}_{n}/**/*counter_collection.csv",recursive=True)
    if not f: continue
    rows=[r for r in csv.DictReader(open(f[0])) if "ftgp_step_kernel" in r["Kernel_Name"]]
    last=max(int(r["Dispatch_Id"]) for r in rows)
    for r in rows:
        if int(r["Dispatch_Id"])==last:
            out["counters"][r["Counter_Name"]]=out["counters"].get(r["Counter_Name"],0)+float(r["Counter_Value"]) if n!="c" or r["Counter_Name"]!="SQ_INSTS_VALU" else out["counters"].get(r["Counter_Name"],0)
            out["kernel"]=r["Kernel_Name"]; out["vgpr"]=r.get("VGPR_Count") or r.get("Arch_VGPR_Count"); out["scratch"]=r.get("Scratch_Size") or r.get("Private_Segment_Size"); out["lds"]=r.get("LDS_Block_Size")
    out["log_"+n]=open(f"gpurun_out/sq_${tag}_{n}.log").read().strip().splitlines()[-1]
c=out["counters"]; cars=$ENVS*$STEPS
d={}
if "SQ_INSTS_VALU" in c: d["valu_insts_per_car_step"]=c["SQ_INSTS_VALU"]/cars
if "SQ_INSTS_SALU" in c: d["salu_insts_per_car_step"]=c["SQ_INSTS_SALU"]/cars
if "SQ_INSTS_LDS" in c: d["lds_insts_per_car_step"]=c["SQ_INSTS_LDS"]/cars
if "SQ_INSTS_VMEM_RD" in c: d["vmem_rd_insts_per_car_step"]=c["SQ_INSTS_VMEM_RD"]/cars
if "SQ_INSTS_VMEM_WR" in c: d["vmem_wr_insts_per_car_step"]=c["SQ_INSTS_VMEM_WR"]/cars
if "SQ_WAVE_CYCLES" in c: d["wave_quadcycles_per_car_step"]=c["SQ_WAVE_CYCLES"]/cars
if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c: d["valu_lane_utilisation"]=c["SQ_THREAD_CYCLES_VALU"]/(64*c["SQ_ACTIVE_INST_VALU"])
for k in ("SQ_ACTIVE_INST_VALU","SQ_WAIT_INST_ANY","SQ_WAIT_ANY","SQ_ACTIVE_INST_ANY","SQ_ACTIVE_INST_SCA","SQ_ACTIVE_INST_LDS","SQ_ACTIVE_INST_VMEM"):
    if k in c and "SQ_WAVE_CYCLES" in c: d[k+"_frac_of_wave_cycles"]=c[k]/c["SQ_WAVE_CYCLES"]
out["derived"]=d
json.dump(out,open("gpurun_out/sq_${tag}.json","w"),indent=1); print(json.dumps(out["derived"],indent=1)); print(out.get("log_a"))
PY
