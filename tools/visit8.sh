#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "metrics" > gpurun_out/pytest_m.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_m.log
timeout -k 10 300 python3 tools/metrics_ab.py > gpurun_out/metrics_ab.log 2>&1 || { tail -5 gpurun_out/metrics_ab.log; exit 1; }; cat gpurun_out/metrics_ab.log
