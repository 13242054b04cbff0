#!/bin/bash
# The FAKELIDAR mode of the step kernel (raycast.py:5-21 as the K2 of the loop) at the headline's batch: bench lines (driver shape and
# 100-step launches), rocprofv3 kernel trace of the same command, SQ / traffic / L2 counters in separate --pmc passes.
#   tools/fakelidar_evidence.sh [tag]      -> gpurun_out/fakelidar_<tag>.log (+ bench / kernel-stats files beside it)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
tag=${1:-now}; out=gpurun_out/fakelidar_$tag.log
export FTGP_PROF_LIDAR=fakelidar
STEPS=${FAKE_STEPS:-100}
echo "# kernel_source_sha=$(python3 tools/evidence.py sha) $(date '+%Y-%m-%d %H:%M:%S') fakelidar_evidence $tag: 4096 envs x 1080 rays, track, fast, FAKELIDAR mode" > $out
timeout -k 10 300 python3 bench.py --lidar fakelidar --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/fakelidar_${tag}_bench20.json 2>> $out || { echo "bench20 failed" >> $out; tail -5 $out; exit 1; }
timeout -k 10 300 python3 bench.py --lidar fakelidar --steps $STEPS --warmup 20 > gpurun_out/fakelidar_${tag}_bench.json 2>> $out || { echo "bench failed" >> $out; tail -5 $out; exit 1; }
python3 - $tag >> $out <<'PY'
import json, sys
for f in ("bench20", "bench"):
    j = json.loads(open(f"gpurun_out/fakelidar_{sys.argv[1]}_{f}.json").read().strip().splitlines()[-1])
    r = j["roofline"]
    print(f"{f}: {j['value']:.4g} env-steps/s, {j['ms_per_step'] * 1e3:.2f} us/step wall, kernel {r['kernel_ms_per_launch'] * 1e3 / j['steps']:.2f} us/step, frac {r['frac']:.4f}, {r['kernel'][:60]}")
PY
rm -rf gpurun_out/fk_stats
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fk_stats -- python3 bench.py --lidar fakelidar --steps $STEPS --warmup 20 --no-cpu-baseline > gpurun_out/fk_stats.log 2>&1 || { tail -5 gpurun_out/fk_stats.log; exit 1; }
cp $(find gpurun_out/fk_stats -name '*kernel_stats.csv' | head -1) gpurun_out/fakelidar_${tag}_kernel_stats.csv
echo "kernel stats (rocprofv3 --kernel-trace --stats of the $STEPS-step bench command):" >> $out; head -4 gpurun_out/fakelidar_${tag}_kernel_stats.csv >> $out
CASE="4096 1080 fast $STEPS"
bash tools/pmc.sh fk1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -- $CASE >> $out 2>&1
bash tools/pmc.sh fk2 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_BUSY_CYCLES -- $CASE >> $out 2>&1
bash tools/pmc.sh fk3 FETCH_SIZE -- $CASE >> $out 2>&1
bash tools/pmc.sh fk4 WRITE_SIZE -- $CASE >> $out 2>&1
bash tools/pmc.sh fk5 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -- $CASE >> $out 2>&1
grep -h "^kernel ms" gpurun_out/pmc_fk1.log | tail -1 >> $out
cp $out $out.tmp
python3 - $STEPS $out.tmp >> $out <<'PY'
import re, sys
S = int(sys.argv[1]); n = 4096 * S; c = {}
for line in open(sys.argv[2]):
    m = re.match(r"^([A-Za-z_0-9]+) ([0-9.e+]+)$", line.strip())
    if m: c[m.group(1)] = float(m.group(2))
if "SQ_INSTS_VALU" in c:
    print(f"per car-step: valu {c['SQ_INSTS_VALU'] / n:.0f}, salu {c['SQ_INSTS_SALU'] / n:.0f}, lds {c['SQ_INSTS_LDS'] / n:.0f}, vmem_rd {c['SQ_INSTS_VMEM_RD'] / n:.0f}, branch {c.get('SQ_INSTS_BRANCH', 0) / n:.0f}")
    print(f"waves waiting {c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']:.2f} of their life, for an issue slot {c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']:.2f}")
if "SQ_THREAD_CYCLES_VALU" in c: print(f"lanes enabled per vector instruction {c['SQ_THREAD_CYCLES_VALU'] / c['SQ_ACTIVE_INST_VALU']:.1f} of 64")
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    f2, w = 2 * 1024 * c["FETCH_SIZE"] / n, 1024 * c["WRITE_SIZE"] / n
    print(f"traffic per env-step: fetch x2 {f2:.0f} B + write {w:.0f} B = {f2 + w:.0f} B = {(f2 + w) / 5152:.2f} x algorithmic (5152 B)")
if "TCC_HIT_sum" in c: print(f"L2 hit rate {c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']):.3f}, L2 requests per env-step {c['TCC_REQ_sum'] / n:.0f}")
PY
rm -f $out.tmp
cat $out
