#!/bin/bash
# A/B on the GPU box: tools/ab.sh "<tag>:<-D flags>" ... -> builds gpurun_out/libftgp_<tag>.so and times the throughput configurations
set -e
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
libs=()
for spec in "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -w -mllvm -amdgpu-atomic-optimizer-strategy=None -fno-slp-vectorize $flags -o gpurun_out/libftgp_$tag.so ft_grandprix_amd/csrc/ftgp_api.hip -ldl
  libs+=(gpurun_out/libftgp_$tag.so)
done
python3 tools/quick_perf.py "${libs[@]}"
