#!/bin/bash
# One GPU-box visit: parity tests -> bench -> rocprofv3 kernel trace.  A step killed by its timeout ends the visit.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {  # name timeout cmd...
    local name=$1 t=$2; shift 2
    echo "== $name" | tee -a gpurun_out/round.log
    timeout -k 10 "$t" "$@" > "gpurun_out/$name.log" 2>&1
    local rc=$?
    echo "$name rc=$rc" | tee -a gpurun_out/round.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "$name was killed by its timeout: stopping" | tee -a gpurun_out/round.log; exit 1; fi
    return $rc
}
: > gpurun_out/round.log
rocminfo 2>/dev/null | grep -m1 -E "gfx9[0-9a-z]+" >> gpurun_out/round.log
nproc >> gpurun_out/round.log
WHAT=${1:-all}
if [ "$WHAT" = all ] || [ "$WHAT" = test ]; then
    run pytest_gpu 1000 python -m pytest tests -m gpu -x -q ${PYTEST_ARGS:-}
    tail -25 gpurun_out/pytest_gpu.log
fi
if [ "$WHAT" = all ] || [ "$WHAT" = bench ]; then
    run bench 500 python bench.py --steps ${BENCH_STEPS:-500} --warmup 50
    tail -3 gpurun_out/bench.log
fi
if [ "$WHAT" = all ] || [ "$WHAT" = prof ]; then
    rm -rf gpurun_out/prof
    run rocprof 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps ${BENCH_STEPS:-500} --warmup 50 --no-cpu-baseline
    find gpurun_out/prof -name "*kernel_stats*.csv" | head -1 | xargs -r head -8
fi
exit 0
