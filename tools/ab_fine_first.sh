#!/bin/bash
# The first look-up of every ray in 64-sector planes of its own, the rest of the march in the batch's coarse planes (round 4 experiment, not
# adopted): the previous commit's library, the experiment's with one plane set (same code) and with FTGP_FINE_FIRST=1.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
out=gpurun_out/ab_fine_first.log
echo "# kernel_source_sha=$(python3 tools/evidence.py sha) $(date '+%Y-%m-%d %H:%M:%S') ab_fine_first: prev = the commit before; off = one plane set; on = FTGP_FINE_FIRST=1, first look-up in 64-sector planes" > $out
for i in 1 2 3; do
  QUICK_SPAWN=1 python3 tools/quick_perf.py ft_grandprix_amd/lib/libftgp_prev.so | sed 's/^libftgp_prev.so/prev/' >> $out 2>&1
  QUICK_SPAWN=1 python3 tools/quick_perf.py | sed 's/^libftgp.so/off /' >> $out 2>&1
  FTGP_FINE_FIRST=1 QUICK_SPAWN=1 python3 tools/quick_perf.py | sed 's/^libftgp.so/on  /' >> $out 2>&1
done
cat $out
