#!/bin/bash
# What one more vector instruction of a kind costs in the march loop (GPU box): builds the library with 16 independent filler
# instructions per iteration (-DFTGP_PAD_VALU=16 -DFTGP_PAD_ASM=...; diagnostic, never shipped) and times the headline
# configuration.  Operands: %0 a scratch VGPR, %1 an SGPR pair (write-only), %2 a scratch VGPR pair, %3 the lane index, %4 a 64-bit SGPR mask.
set -e
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -w -mllvm -amdgpu-atomic-optimizer-strategy=None -fno-slp-vectorize"
build() { /opt/rocm/bin/hipcc $FLAGS "${@:2}" -o gpurun_out/libftgp_$1.so ft_grandprix_amd/csrc/ftgp_api.hip -ldl; }
build base
names=(base); 
while read -r tag insn; do
  [ -z "$tag" ] && continue
  build $tag -DFTGP_PAD_VALU=16 "-DFTGP_PAD_ASM=\"$insn\""
  names+=($tag)
done <<'LIST'
add_u32 v_add_u32 %0, %0, %3
xor v_xor_b32 %0, %0, %3
and v_and_b32 %0, %0, %3
lshrrev v_lshrrev_b32 %0, 8, %0
mov v_mov_b32 %0, %3
add_f32 v_add_f32 %0, %0, %3
mul_f32 v_mul_f32 %0, %0, %3
min_f32 v_min_f32 %0, %0, %3
fma_f32 v_fma_f32 %0, %0, %3, %3
mul_i24 v_mul_i32_i24 %0, %0, %3
mad_i24 v_mad_i32_i24 %0, %0, %3, %3
lshl_add v_lshl_add_u32 %0, %0, 1, %3
add3 v_add3_u32 %0, %0, %3, %3
bfe v_bfe_u32 %0, %0, 8, 8
med3_i32 v_med3_i32 %0, %0, %3, %3
cndmask_s v_cndmask_b32_e64 %0, %0, %3, %4
cndmask_vcc v_cndmask_b32_e32 %0, %0, %3, vcc
cmp_vcc v_cmp_lt_i32_e32 vcc, %0, %3
cmp_s v_cmp_lt_i32_e64 %1, %0, %3
cvt_f32_i32 v_cvt_f32_i32 %0, %0
cvt_flr v_cvt_flr_i32_f32 %0, %0
cvt_ubyte0 v_cvt_f32_ubyte0 %0, %0
fract v_fract_f32 %0, %0
floor v_floor_f32 %0, %0
sdwa_add v_add_u32_sdwa %0, %0, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0
rcp v_rcp_f32 %0, %0
mbcnt v_mbcnt_lo_u32_b32 %0, %3, %0
mul_lo v_mul_lo_u32 %0, %0, %3
fma_f64 v_fma_f64 %2, %2, %2, %2
mul_f64 v_mul_f64 %2, %2, %2
add_f64 v_add_f64 %2, %2, %2
LIST
python3 - "${names[@]}" <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from ft_grandprix_amd import capi
from ft_grandprix_amd.track import load_track
t = load_track("track"); base = None
for tag in sys.argv[1:]:
    lib = capi.CLib(f"gpurun_out/libftgp_{tag}.so", "ftgp_")
    with capi.Env(lib, t, n_envs=4096, n_rays=1080, spawn_mode=1, seed=1234) as e:
        e.rollout("fast", 100); e.last_kernel_ms(); best = 1e9
        for _ in range(3):
            e.rollout("fast", 300); best = min(best, e.last_kernel_ms())
    us = best * 1e3 / 300
    if base is None: base = us
    # 16 fillers x 67.4 wave-iterations x 4 cars per SIMD and step
    print(f"{tag:12s} {us:7.2f} us/step  +{us - base:5.2f} us = {(us - base) * 1e-6 * 2.3e9 / (16 * 67.4 * 4):5.2f} cycles per instruction (at 2.3 GHz)", flush=True)
PY
