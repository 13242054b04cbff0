#!/bin/bash
# The judged evidence of round 5 on one box, on the tree's sources: tools/round5_evidence.sh [a|b]
#   a: parity tests, the headline profile (tools/collect_profile.py: counters, traffic, bench line with cpu_baseline, kernel trace of the same command),
#      the driver's exact bench command plain and under rocprofv3 --kernel-trace --stats, instruction counts by type, launch fixed cost
#   b: configs 5 and 2 (counters, traffic, bench, kernel trace), the FAKELIDAR mode, the soak
# Everything is stamped with the hash of the kernel sources it was measured on; copy with tools/publish_round.sh profiles/round5.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/profile; export TMPDIR=/tmp
SHA=$(python3 tools/evidence.py sha)
part=${1:-a}
if [ "$part" = a ]; then
  timeout -k 10 400 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest_gpu.log
  bash tools/profile_round.sh headline || exit 1
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/profile/bench_driver_shape.log 2>&1 || exit 1
  grep '^{' gpurun_out/profile/bench_driver_shape.log | tail -1 > gpurun_out/profile/bench_driver_shape.json; echo $SHA > gpurun_out/profile/bench_driver_shape.json.sha
  rm -rf gpurun_out/profile/raw_driver
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/profile/raw_driver -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/profile/raw_driver.log 2>&1 || exit 1
  cp $(find gpurun_out/profile/raw_driver -name '*kernel_stats.csv' | head -1) gpurun_out/profile/kernel_stats_driver_shape_20steps.csv; echo $SHA > gpurun_out/profile/kernel_stats_driver_shape_20steps.csv.sha
  rm -rf gpurun_out/profile/raw_driver
  bash tools/issue_counters.sh > /dev/null 2>&1
  bash tools/launch_host.sh > /dev/null 2>&1
  python3 -c "import json; d=json.load(open('gpurun_out/profile/bench_driver_shape.json')); print('driver shape:', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['bound'])"
else
  bash tools/profile_round.sh multi || exit 1
  bash tools/profile_round.sh circle || exit 1
  bash tools/fakelidar_evidence.sh final > /dev/null 2>&1
  head -3 gpurun_out/fakelidar_final.log
  timeout -k 10 600 python3 tools/soak.py 4000 128 > gpurun_out/soak_r5.log 2>&1; echo "soak rc=$?"; tail -1 gpurun_out/soak_r5.log
  python3 tools/evidence.py stamp gpurun_out/soak_r5.log
fi
ls gpurun_out/profile
