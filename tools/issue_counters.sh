#!/bin/bash
# Where the SIMDs' issue cycles of the headline launch go, by instruction type, and how many lanes its vector instructions keep busy:
# SQ_ACTIVE_INST_* (cycles a SIMD spends on an instruction of the type), instruction counts by type, SQ_THREAD_CYCLES_VALU (lane-cycles).
# Separate --pmc passes (kernel-trace only).  -> gpurun_out/issue_by_type.log
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/issue_by_type.log
CASE="${CASE:-4096 1080 fast 500}"
echo "# kernel_source_sha=$(python3 tools/evidence.py sha) $(date '+%Y-%m-%d %H:%M:%S') issue_by_type: CASE=$CASE (last launch; sums over all SIMDs / XCDs as rocprofv3 reports them)" > $out
bash tools/pmc.sh i1 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE -- $CASE >> $out 2>&1
bash tools/pmc.sh i2 SQ_INSTS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_VSKIPPED -- $CASE >> $out 2>&1
bash tools/pmc.sh i3 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LEVEL_WAVES SQ_INST_CYCLES_SALU SQ_CYCLES GRBM_GUI_ACTIVE -- $CASE >> $out 2>&1
python3 - >> $out <<'PY'
import re
c = {}
for line in open("gpurun_out/issue_by_type.log"):
    m = re.match(r"^([A-Z_0-9]+) ([0-9.e+]+)$", line.strip())
    if m: c[m.group(1)] = float(m.group(2))
cyc = c["GRBM_GUI_ACTIVE"] / 8                      # shader cycles of the launch (8 XCDs report)
simds = 1024
print(f"shader cycles {cyc:.4g}; SIMD-cycles {cyc * simds:.4g}")
# SQ_ACTIVE_INST_* tick once per instruction of the type (twice for a transcendental): instruction counts, NOT cycles (tools/issue_calib.sh,
# profiles/round5/issue_calib.log) -- so they are printed per SIMD-cycle; what an instruction costs is in the calibration
for k in ("ANY", "VALU", "SCA", "LDS", "VMEM", "FLAT", "MISC"):
    v = c.get("SQ_ACTIVE_INST_" + k)
    if v is not None: print(f"SQ_ACTIVE_INST_{k:5s} per SIMD-cycle = {v / (cyc * simds):.3f}")
if "SQ_INSTS_VALU" in c:
    print(f"vector pipe occupancy between {c['SQ_INSTS_VALU'] * 2.28 / (cyc * simds):.2f} (every vector instruction at v_add_u32's 2.28 SIMD-cycles) and "
          f"{min(1.0, c['SQ_INSTS_VALU'] * 4.32 / (cyc * simds)):.2f} (at v_cmp's 4.32, capped at 1)")
if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c:
    print(f"lanes enabled per vector instruction: SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU = {c['SQ_THREAD_CYCLES_VALU'] / c['SQ_ACTIVE_INST_VALU']:.1f} of 64")
n = 4096 * 500
print("per car-step: " + ", ".join(f"{k[9:].lower() or 'all'} {c[k] / n:.0f}" for k in ("SQ_INSTS", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_SMEM") if k in c))
PY
cat $out
