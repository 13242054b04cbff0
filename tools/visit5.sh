#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 600 python3 tools/interleave_probe.py > gpurun_out/interleave_probe.log 2>&1 || { tail -5 gpurun_out/interleave_probe.log; exit 1; }; cat gpurun_out/interleave_probe.log
