#!/bin/bash
# The judged evidence of round 4 on one box (after tools/profile_round.sh headline): the other two profiled configurations, the driver's
# exact bench command under rocprofv3 --kernel-trace --stats, the secondary counters, instruction counts by phase.  Everything is stamped
# with the hash of the kernel sources it was measured on; copy with `python3 tools/evidence.py publish profiles/round4 ...`.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
SHA=$(python3 tools/evidence.py sha)
bash tools/profile_round.sh multi || exit 1
bash tools/profile_round.sh circle || exit 1
# the driver's command: bench line, then the same command under the kernel trace
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/profile/bench_driver_shape.log 2>&1 || exit 1
grep '^{' gpurun_out/profile/bench_driver_shape.log | tail -1 > gpurun_out/profile/bench_driver_shape.json; echo $SHA > gpurun_out/profile/bench_driver_shape.json.sha
rm -rf gpurun_out/profile/raw_driver
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/profile/raw_driver -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/profile/raw_driver.log 2>&1 || exit 1
cp $(find gpurun_out/profile/raw_driver -name '*kernel_stats.csv' | head -1) gpurun_out/profile/kernel_stats_driver_shape_20steps.csv; echo $SHA > gpurun_out/profile/kernel_stats_driver_shape_20steps.csv.sha
rm -rf gpurun_out/profile/raw_driver
bash tools/extra_counters.sh > /dev/null 2>&1 || exit 1
bash tools/valu_by_phase.sh > /dev/null 2>&1; python3 tools/evidence.py stamp gpurun_out/valu_by_phase.log
ls gpurun_out/profile
