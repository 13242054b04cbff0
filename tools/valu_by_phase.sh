#!/bin/bash
# Vector / scalar instructions per car-step by phase of the step kernel: SQ_INSTS_VALU / SQ_INSTS_SALU of the 500-step headline
# launch for the shipped kernel with the `fast` driver (A) and the null driver (B: A - B = K5 + window flush), and for diagnostic
# builds with K1 (C), the sweep (D) or both (E: what is left is the step loop itself) compiled out, null driver.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -w -mllvm -amdgpu-atomic-optimizer-strategy=None -fno-slp-vectorize"
/opt/rocm/bin/hipcc $FLAGS -DFTGP_DIAG -DFTGP_ABLATE_K1 -o gpurun_out/libftgp_noK1.so ft_grandprix_amd/csrc/ftgp_api.hip -ldl || exit 1
/opt/rocm/bin/hipcc $FLAGS -DFTGP_DIAG -DFTGP_ABLATE_K2 -o gpurun_out/libftgp_noK2.so ft_grandprix_amd/csrc/ftgp_api.hip -ldl || exit 1
/opt/rocm/bin/hipcc $FLAGS -DFTGP_DIAG -DFTGP_ABLATE_K1 -DFTGP_ABLATE_K2 -o gpurun_out/libftgp_noK1K2.so ft_grandprix_amd/csrc/ftgp_api.hip -ldl || exit 1
{
echo "A: shipped, fast";             bash tools/pmc.sh pA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -- 4096 1080 fast 500
echo "B: shipped, lobotomy";         bash tools/pmc.sh pB SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -- 4096 1080 lobotomy 500
echo "C: no K1, lobotomy";           FTGP_LIB=gpurun_out/libftgp_noK1.so bash tools/pmc.sh pC SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -- 4096 1080 lobotomy 500
echo "D: no sweep, lobotomy";        FTGP_LIB=gpurun_out/libftgp_noK2.so bash tools/pmc.sh pD SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -- 4096 1080 lobotomy 500
echo "E: no K1, no sweep, lobotomy"; FTGP_LIB=gpurun_out/libftgp_noK1K2.so bash tools/pmc.sh pE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -- 4096 1080 lobotomy 500
echo "F: no sweep, fast";            FTGP_LIB=gpurun_out/libftgp_noK2.so bash tools/pmc.sh pF SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -- 4096 1080 fast 500
echo "(divide by 4096 x 500 = 2 048 000 car-steps)"
} > gpurun_out/valu_by_phase.log 2>&1
cat gpurun_out/valu_by_phase.log
