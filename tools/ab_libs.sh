#!/bin/bash
# Same-box A/B of prebuilt libraries (e.g. the previous commit's, built beside the shipped one): quick_perf rows interleaved, N rounds.
#   ab_libs.sh out.log rounds lib1.so lib2.so ...
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
out=gpurun_out/$1; rounds=$2; shift 2
echo "# kernel_source_sha=$(python3 tools/evidence.py sha) $(date '+%Y-%m-%d %H:%M:%S') $(basename $out): $*" > $out
for i in $(seq $rounds); do for l in "$@"; do QUICK_SPAWN=1 QUICK_SHORT=1 python3 tools/quick_perf.py $l >> $out 2>&1; done; done
cat $out
