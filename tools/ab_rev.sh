#!/bin/bash
# A/B of two source trees on the GPU box: tools/ab_rev.sh <dir with csrc/ of the other revision> [rounds]
# (stage the other revision first:  for f in ...; do git show REV:ft_grandprix_amd/csrc/$f > ab_prev/csrc/$f; done)
set -e
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -w -mllvm -amdgpu-atomic-optimizer-strategy=None -fno-slp-vectorize -Iinclude"
/opt/rocm/bin/hipcc $FLAGS -o gpurun_out/libftgp_prev.so $1/csrc/ftgp_api.hip -ldl
/opt/rocm/bin/hipcc $FLAGS -o gpurun_out/libftgp_cur.so ft_grandprix_amd/csrc/ftgp_api.hip -ldl
for i in $(seq 1 ${2:-2}); do python3 tools/quick_perf.py gpurun_out/libftgp_prev.so gpurun_out/libftgp_cur.so; done
