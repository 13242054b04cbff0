#!/bin/bash
# One GPU-box visit for the judged evidence of a build: parity tests, then tools/collect_profile.py for the headline and, with
# arguments, for another config (see collect_profile.py).  profile_round.sh [headline|multi|circle|tests]
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
case "${1:-headline}" in
  tests)    timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log ;;
  headline) timeout -k 10 1100 python3 tools/collect_profile.py 500 > gpurun_out/collect_headline.log 2>&1 || { tail -20 gpurun_out/collect_headline.log; exit 1; }; tail -14 gpurun_out/collect_headline.log ;;
  multi)    timeout -k 10 1100 python3 tools/collect_profile.py 200 4096 fast 4 track config4_multi > gpurun_out/collect_multi.log 2>&1 || { tail -20 gpurun_out/collect_multi.log; exit 1; }; tail -12 gpurun_out/collect_multi.log ;;
  circle)   timeout -k 10 1100 python3 tools/collect_profile.py 500 1024 nidc 1 circle config1_circle > gpurun_out/collect_circle.log 2>&1 || { tail -20 gpurun_out/collect_circle.log; exit 1; }; tail -12 gpurun_out/collect_circle.log ;;
esac
