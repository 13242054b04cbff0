#!/bin/bash
# Same-box A/B of libraries in FAKELIDAR mode (4096 envs x 1080 rays, fast, 300-step launches): tools/diag/fake_ab.sh out.log rounds lib.so ...
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
out=gpurun_out/$1; rounds=$2; shift 2
echo "# kernel_source_sha=$(python3 tools/evidence.py sha) $(date '+%Y-%m-%d %H:%M:%S') fake_ab: $*" > $out
for i in $(seq $rounds); do for l in "$@"; do
  echo "$(basename $l) FAKELIDAR 4096x1080 fast: $(FTGP_LIB=$l FTGP_PROF_LIDAR=fakelidar python3 tools/prof_case.py 4096 1080 fast 300 2>&1 | tail -1)" >> $out
done; done
cat $out
