#!/bin/bash
# round-3 evidence: headline profile (bench line with cpu_baseline, kernel trace, SQ counters, FETCH/WRITE), then the same for
# BASELINE configs[4] (4-car worlds, ftgp_step_kernel<true>) and configs[1] (1024 envs, circle, nidc)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 1100 python3 tools/collect_profile.py 500 > gpurun_out/collect_headline.log 2>&1 || { tail -20 gpurun_out/collect_headline.log; exit 1; }; tail -15 gpurun_out/collect_headline.log
