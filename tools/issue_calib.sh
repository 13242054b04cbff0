#!/bin/bash
# What the SQ counters count per instruction and what an instruction of a kind costs a SIMD (tools/diag/issue_calib.hip): every kind on 1 and
# on 8 waves per SIMD, one --pmc pass each (kernel-trace only).  -> gpurun_out/issue_calib.log
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/issue_calib.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -w tools/diag/issue_calib.hip -o gpurun_out/issue_calib || exit 1
echo "# $(date '+%Y-%m-%d %H:%M:%S') issue_calib: per kind and waves per SIMD -- counter / instruction of the kind, SIMD-cycles / instruction (1024 SIMDs; GRBM_GUI_ACTIVE / 8 = shader cycles of the launch)" > $out
for kind in add cndmask cmp cvt fma rcp fma64 salu nop; do for wps in 1 8; do
  rm -rf gpurun_out/calib_raw
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/calib_raw -- gpurun_out/issue_calib $kind $wps 2000 > gpurun_out/calib_run.log 2>&1 || { echo "$kind $wps failed" >> $out; continue; }
  python3 - >> $out <<'PY'
import csv, glob, re
line = [l for l in open("gpurun_out/calib_run.log") if l.startswith("calib ")][-1].split()
kind, wps, insts, ms = line[1], int(line[3]), float(line[9]), float(line[11])
rows = [r for r in csv.DictReader(open(glob.glob("gpurun_out/calib_raw/**/*counter_collection.csv", recursive=True)[0])) if "calib" in r["Kernel_Name"]]
last = max(int(r["Dispatch_Id"]) for r in rows); c = {}
for r in rows:
    if int(r["Dispatch_Id"]) == last: c[r["Counter_Name"]] = c.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
cyc = c["GRBM_GUI_ACTIVE"] / 8
print(f"{kind:8s} {wps} waves/SIMD: SIMD-cycles per instruction {cyc * 1024 / insts:6.3f} | per instruction: SQ_INSTS_VALU {c['SQ_INSTS_VALU'] / insts:5.3f} SQ_ACTIVE_INST_VALU {c['SQ_ACTIVE_INST_VALU'] / insts:5.3f} "
      f"SQ_INSTS_SALU {c['SQ_INSTS_SALU'] / insts:5.3f} SQ_ACTIVE_INST_SCA {c['SQ_ACTIVE_INST_SCA'] / insts:5.3f} SQ_ACTIVE_INST_ANY {c['SQ_ACTIVE_INST_ANY'] / insts:5.3f} | "
      f"SQ_WAVE_CYCLES per wave-instruction {c['SQ_WAVE_CYCLES'] / insts:6.3f} SQ_BUSY_CYCLES/shader cycle {c['SQ_BUSY_CYCLES'] / cyc:6.2f} | {ms:.3f} ms, {cyc / (ms * 1e6):.2f} GHz")
PY
done; done
cat $out
