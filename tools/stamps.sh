#!/bin/bash
# phase stamps on the GPU box: builds the -DFTGP_DIAG -DFTGP_STAMPS diagnostic library (never shipped) and prints tools/stamps.py for it
set -e
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -w -mllvm -amdgpu-atomic-optimizer-strategy=None -fno-slp-vectorize -DFTGP_DIAG -DFTGP_STAMPS $STAMPS_FLAGS -o gpurun_out/libftgp_stamps.so ft_grandprix_amd/csrc/ftgp_api.hip -ldl
python3 tools/stamps.py gpurun_out/libftgp_stamps.so "$@"
