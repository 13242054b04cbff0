#!/bin/bash
# One GPU-box visit: parity tests -> smoke -> soak -> the driver's bench shape.  A step killed by its timeout ends the visit.
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
    tail -5 gpurun_out/pytest_gpu.log
    run smoke 300 python -c "import __graft_entry__ as g; g.smoke()"
    tail -2 gpurun_out/smoke.log
fi
if [ "$WHAT" = all ] || [ "$WHAT" = soak ]; then
    run soak 1000 python3 tools/soak.py ${SOAK_STEPS:-4000} 128
    tail -18 gpurun_out/soak.log
fi
if [ "$WHAT" = all ] || [ "$WHAT" = bench ]; then
    run bench20 500 python bench.py --steps 20 --warmup 5
    tail -1 gpurun_out/bench20.log | cut -c1-400
fi
exit 0
