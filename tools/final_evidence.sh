#!/bin/bash
# final evidence of a round on one box: phase stamps + workgroup finish spread (diagnostic -DFTGP_DIAG -DFTGP_STAMPS build), then the three profiles
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 300 bash tools/stamps.sh fast > gpurun_out/stamps.log 2>&1 || exit 1; cat gpurun_out/stamps.log
# entry / exit times only (-DFTGP_WG_TIMES): the phase stamps' atomics at the end of a launch would add about a millisecond to every launch
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -w -mllvm -amdgpu-atomic-optimizer-strategy=None -fno-slp-vectorize -DFTGP_DIAG -DFTGP_WG_TIMES -o gpurun_out/libftgp_wgtimes.so ft_grandprix_amd/csrc/ftgp_api.hip -ldl || exit 1
timeout -k 10 300 python3 tools/wg_spread.py gpurun_out/libftgp_wgtimes.so fast 4096 1 20 100 500 > gpurun_out/wg_spread.log 2>&1 || exit 1; cat gpurun_out/wg_spread.log
for w in headline multi circle; do bash tools/profile_round.sh $w || exit 1; done
