#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu.log
timeout -k 10 300 python3 tools/quick_perf.py > gpurun_out/quick_perf.log 2>&1 || exit 1; cat gpurun_out/quick_perf.log
timeout -k 10 300 bash tools/stamps.sh fast > gpurun_out/stamps.log 2>&1 || exit 1; cat gpurun_out/stamps.log
timeout -k 10 300 python3 tools/wg_spread.py gpurun_out/libftgp_stamps.so fast 4096 1 100 500 > gpurun_out/wg_spread.log 2>&1 || exit 1; cat gpurun_out/wg_spread.log
