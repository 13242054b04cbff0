#!/bin/bash
# Same-box sweep of the quad task list's tail (FTGP_QUAD_TAIL = how many of a car's quads go out as a pair and two single groups) on the headline row,
# beside FTGP_NO_QUADS=1 (pairs only):  tools/quad_tail_sweep.sh out.log [rounds] [lib.so]
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
out=gpurun_out/$1; rounds=${2:-3}; lib=${3:-ft_grandprix_amd/lib/libftgp.so}
echo "# kernel_source_sha=$(python3 tools/evidence.py sha) $(date '+%Y-%m-%d %H:%M:%S') quad_tail_sweep: $lib" > $out
for i in $(seq $rounds); do
  echo "pairs only (FTGP_NO_QUADS=1): $(FTGP_NO_QUADS=1 QUICK_CASES=${QUICK_CASES:-0} QUICK_SHORT=1 python3 tools/quick_perf.py $lib 2>&1)" >> $out
  for qt in 0 1 2 3 4; do
    echo "FTGP_QUAD_TAIL=$qt: $(FTGP_QUAD_TAIL=$qt QUICK_CASES=${QUICK_CASES:-0} QUICK_SHORT=1 python3 tools/quick_perf.py $lib 2>&1)" >> $out
  done
done
cat $out
