#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
QUICK_SHORT=1 QUICK_CASES=0,1 timeout -k 10 900 bash tools/ab.sh "r48:" "r40:-DFTGP_REFILL=40" "r44:-DFTGP_REFILL=44" "r52:-DFTGP_REFILL=52" "r56:-DFTGP_REFILL=56" > gpurun_out/ab_refill.log 2>&1 || exit 1; cat gpurun_out/ab_refill.log
