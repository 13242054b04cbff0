#!/bin/bash
# 32 vs 64 direction sectors of the box field (and the shipped library of the previous commit as `base`), same box, two rounds
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -w -mllvm -amdgpu-atomic-optimizer-strategy=None -fno-slp-vectorize"
/opt/rocm/bin/hipcc $FLAGS -DFTGP_SECTORS=64 -o gpurun_out/libftgp_s64.so ft_grandprix_amd/csrc/ftgp_api.hip -ldl || exit 1
/opt/rocm/bin/hipcc $FLAGS -DFTGP_SECTORS=32 -o gpurun_out/libftgp_s32.so ft_grandprix_amd/csrc/ftgp_api.hip -ldl || exit 1
libs=(gpurun_out/libftgp_s32.so gpurun_out/libftgp_s64.so)
[ -f ft_grandprix_amd/lib/libftgp_base.so ] && libs=(ft_grandprix_amd/lib/libftgp_base.so "${libs[@]}")
rm -f gpurun_out/ab_sectors.log
for i in 1 2; do QUICK_SHORT=1 timeout -k 10 900 python3 tools/quick_perf.py "${libs[@]}" >> gpurun_out/ab_sectors.log 2>&1 || exit 1; done
cat gpurun_out/ab_sectors.log
