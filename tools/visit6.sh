#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu.log
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench20.log 2>&1 || exit 1; python3 -c "
import json; d=json.loads(open('gpurun_out/bench20.log').read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_launch'])"
timeout -k 10 300 python3 bench.py --steps 500 --warmup 50 --no-cpu-baseline > gpurun_out/bench500.log 2>&1 || exit 1; python3 -c "
import json; d=json.loads(open('gpurun_out/bench500.log').read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_launch'])"
