#!/bin/bash
# Does a wave64 vector instruction whose upper (or lower) 32 lanes are disabled issue in one pass instead of two?  Diagnostic builds
# (never shipped) with 16 filler instructions per march iteration: all lanes enabled / lanes 32..63 disabled / lanes 0..31 disabled,
# for a 2-cycle instruction (v_add_u32), a 4-cycle one (v_cndmask_b32) and a binary32 fma.  exec is saved, changed and restored
# inside ONE asm statement (see the comment at FTGP_PAD_EXEC in ftgp_kernels.hip for what went wrong in round 3).
set -e
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -w -mllvm -amdgpu-atomic-optimizer-strategy=None -fno-slp-vectorize"
build() { /opt/rocm/bin/hipcc $FLAGS "${@:2}" -o gpurun_out/libftgp_$1.so ft_grandprix_amd/csrc/ftgp_api.hip -ldl; }
build hx_base
libs="gpurun_out/libftgp_hx_base.so"
for op in add cnd fma; do
  case $op in
    add) A='-DFTGP_PAD_EXEC_ASM(r)="v_add_u32 " r ", " r ", %3"' ;;
    cnd) A='-DFTGP_PAD_EXEC_ASM(r)="v_cndmask_b32_e64 " r ", " r ", %3, %4"' ;;
    fma) A='-DFTGP_PAD_EXEC_ASM(r)="v_fma_f32 " r ", " r ", " r ", %3"' ;;
  esac
  for m in 0 1 2; do
    build hx_${op}_$m -DFTGP_PAD_EXEC=$m "$A"
    libs="$libs gpurun_out/libftgp_hx_${op}_$m.so"
  done
done
QUICK_CASES=${QUICK_CASES:-0} timeout -k 10 400 python3 tools/quick_perf.py $libs
