#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 1100 python3 tools/collect_profile.py 200 4096 fast 4 track config4_multi > gpurun_out/collect_multi.log 2>&1 || { tail -20 gpurun_out/collect_multi.log; exit 1; }; tail -12 gpurun_out/collect_multi.log
