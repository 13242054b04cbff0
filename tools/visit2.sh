#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 300 python3 tools/interleave_probe.py > gpurun_out/interleave_probe.log 2>&1 || exit 1; cat gpurun_out/interleave_probe.log
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > gpurun_out/bench20.log 2>&1 || exit 1; tail -1 gpurun_out/bench20.log
