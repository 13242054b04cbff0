#!/bin/bash
# direction sectors of the box field: tools/ab_sectors.sh 64 128 ...  (same box, two rounds; the shipped library first)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -w -mllvm -amdgpu-atomic-optimizer-strategy=None -fno-slp-vectorize"
libs=(ft_grandprix_amd/lib/libftgp.so)
for n in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -DFTGP_SECTORS=$n -o gpurun_out/libftgp_s$n.so ft_grandprix_amd/csrc/ftgp_api.hip -ldl || exit 1
  libs+=(gpurun_out/libftgp_s$n.so)
done
rm -f gpurun_out/ab_sectors.log
for i in 1 2; do QUICK_SHORT=1 timeout -k 10 900 python3 tools/quick_perf.py "${libs[@]}" >> gpurun_out/ab_sectors.log 2>&1 || exit 1; done
cat gpurun_out/ab_sectors.log
