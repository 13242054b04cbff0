#!/bin/bash
# Same-box sweep of the box field's direction-sector count (FTGP_SECTORS_RT, a handle's choice at ftgp_create) on the throughput rows of tools/quick_perf.py:
#   tools/sectors_rt_sweep.sh out.log [rounds] [lib.so]
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
out=gpurun_out/$1; rounds=${2:-2}; lib=${3:-ft_grandprix_amd/lib/libftgp.so}
echo "# kernel_source_sha=$(python3 tools/evidence.py sha) $(date '+%Y-%m-%d %H:%M:%S') sectors_rt_sweep: $lib" > $out
for i in $(seq $rounds); do
  echo "default: $(QUICK_SPAWN=1 QUICK_SHORT=1 python3 tools/quick_perf.py $lib 2>&1)" >> $out
  for n in 8 16 32 64; do
    echo "FTGP_SECTORS_RT=$n: $(FTGP_SECTORS_RT=$n QUICK_SPAWN=1 QUICK_SHORT=1 python3 tools/quick_perf.py $lib 2>&1)" >> $out
  done
done
cat $out
