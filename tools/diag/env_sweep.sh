#!/bin/bash
# Same-box sweep of create-time switches on the headline row (tools/quick_perf.py): tools/diag/env_sweep.sh out.log rounds "VAR=val ..." ...   ("" = defaults)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
out=gpurun_out/$1; rounds=$2; shift 2
echo "# kernel_source_sha=$(python3 tools/evidence.py sha) $(date '+%Y-%m-%d %H:%M:%S') env_sweep" > $out
for i in $(seq $rounds); do for sw in "$@"; do
  echo "[${sw:-defaults}] $(env $sw QUICK_CASES=${QUICK_CASES:-0} QUICK_SHORT=1 python3 tools/quick_perf.py 2>&1)" >> $out
done; done
cat $out
