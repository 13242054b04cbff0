#!/bin/bash
# diagnostic builds (never shipped): libftgp_<tag>.so with one phase compiled out; timing only
set -e
cd "$(dirname "$0")/.."
for tag in K1 K2; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -w -mllvm -amdgpu-atomic-optimizer-strategy=None -fno-slp-vectorize -DFTGP_DIAG -DFTGP_ABLATE_$tag -o gpurun_out/libftgp_no$tag.so ft_grandprix_amd/csrc/ftgp_api.hip -ldl
done
