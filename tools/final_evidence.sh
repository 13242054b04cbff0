#!/bin/bash
# final evidence of a round on one box: phase stamps + workgroup finish spread (diagnostic -DFTGP_STAMPS build), then the three profiles
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 300 bash tools/stamps.sh fast > gpurun_out/stamps.log 2>&1 || exit 1; cat gpurun_out/stamps.log
timeout -k 10 300 python3 tools/wg_spread.py gpurun_out/libftgp_stamps.so fast 4096 1 100 500 > gpurun_out/wg_spread.log 2>&1 || exit 1; cat gpurun_out/wg_spread.log
for w in headline multi circle; do bash tools/profile_round.sh $w || exit 1; done
