#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_k1_invariants.py -m gpu -x -q > gpurun_out/pytest_k1.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_k1.log
timeout -k 10 900 python3 tools/characterise_fast.py > gpurun_out/characterise_fast.log 2>&1 || { tail -5 gpurun_out/characterise_fast.log; exit 1; }; cat gpurun_out/characterise_fast.log
