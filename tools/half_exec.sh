#!/bin/bash
# Does a wave64 vector instruction whose upper 32 lanes are disabled issue in one pass instead of two?  16 fillers per march
# iteration with full exec vs with exec_hi = 0 (diagnostic builds, never shipped).
set -e
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -w -mllvm -amdgpu-atomic-optimizer-strategy=None -fno-slp-vectorize"
build() { /opt/rocm/bin/hipcc $FLAGS "${@:2}" -o gpurun_out/libftgp_$1.so ft_grandprix_amd/csrc/ftgp_api.hip -ldl; }
build hx_base
build hx_full_add -DFTGP_PAD_VALU=16 '-DFTGP_PAD_ASM="v_add_u32 %0, %0, %3"'
build hx_half_add -DFTGP_PAD_VALU=16 -DFTGP_PAD_HALF '-DFTGP_PAD_ASM="v_add_u32 %0, %0, %3"'
build hx_full_cnd -DFTGP_PAD_VALU=16 '-DFTGP_PAD_ASM="v_cndmask_b32_e64 %0, %0, %3, %4"'
build hx_half_cnd -DFTGP_PAD_VALU=16 -DFTGP_PAD_HALF '-DFTGP_PAD_ASM="v_cndmask_b32_e64 %0, %0, %3, %4"'
QUICK_CASES=${QUICK_CASES:-0} python3 tools/quick_perf.py gpurun_out/libftgp_hx_base.so gpurun_out/libftgp_hx_full_add.so gpurun_out/libftgp_hx_half_add.so gpurun_out/libftgp_hx_full_cnd.so gpurun_out/libftgp_hx_half_cnd.so
