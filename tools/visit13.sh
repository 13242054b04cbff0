#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 1100 python3 tools/collect_profile.py 500 1024 nidc 1 circle config1_circle > gpurun_out/collect_circle.log 2>&1 || { tail -20 gpurun_out/collect_circle.log; exit 1; }; tail -12 gpurun_out/collect_circle.log
