#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
QUICK_SHORT=1 QUICK_CASES=0,1,2 timeout -k 10 900 bash tools/ab.sh "nofair:-DFTGP_NO_FAIR" "fair13:-DFTGP_FAIR_SHIFT=13" "fair14:-DFTGP_FAIR_SHIFT=14" "fair15:-DFTGP_FAIR_SHIFT=15" "fair16:-DFTGP_FAIR_SHIFT=16" "fair18:-DFTGP_FAIR_SHIFT=18" > gpurun_out/ab_fair.log 2>&1 || exit 1; cat gpurun_out/ab_fair.log
STAMPS_FLAGS="" timeout -k 10 300 bash tools/stamps.sh fast > gpurun_out/stamps.log 2>&1 || exit 1
timeout -k 10 300 python3 tools/wg_spread.py gpurun_out/libftgp_stamps.so fast 4096 1 100 500 > gpurun_out/wg_spread.log 2>&1 || exit 1; cat gpurun_out/wg_spread.log
