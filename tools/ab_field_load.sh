#!/bin/bash
# Cache-policy bits on the march's field load (global_load_ushort): default vs nt / sc0 / sc1 / sc0 sc1, same box, two rounds.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -w -mllvm -amdgpu-atomic-optimizer-strategy=None -fno-slp-vectorize"
libs=(ft_grandprix_amd/lib/libftgp.so)
for m in nt sc0 sc1; do
  /opt/rocm/bin/hipcc $FLAGS "-DFTGP_FIELD_LOAD_MOD=\" $m\"" -o gpurun_out/libftgp_ld_$m.so ft_grandprix_amd/csrc/ftgp_api.hip -ldl || exit 1
  libs+=(gpurun_out/libftgp_ld_$m.so)
done
rm -f gpurun_out/ab_field_load.log
for i in 1 2; do QUICK_SHORT=1 QUICK_CASES=0,1,2 timeout -k 10 600 python3 tools/quick_perf.py "${libs[@]}" >> gpurun_out/ab_field_load.log 2>&1 || exit 1; done
cat gpurun_out/ab_field_load.log
