#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
QUICK_SHORT=1 timeout -k 10 600 python3 tools/quick_perf.py > gpurun_out/quick_perf.log 2>&1 || exit 1; cat gpurun_out/quick_perf.log
timeout -k 10 300 python3 bench.py --steps 200 --warmup 50 --cars 4 --no-cpu-baseline > gpurun_out/bench_multi.log 2>&1 || exit 1; python3 -c "
import json; d=json.loads(open('gpurun_out/bench_multi.log').read().strip().split('\n')[-1]); print('multi bench', d['value'], d['ms_per_step'])"
