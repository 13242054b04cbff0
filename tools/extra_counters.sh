#!/bin/bash
# The round's secondary evidence on one box: instruction mix, vector-memory path and L2 counters of the headline step kernel
# (own --pmc passes, kernel-trace only), and the phase ablation (K1 / K2 compiled out, drivers, 8 rays).
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
STAMP="# kernel_source_sha=$(python3 tools/evidence.py sha) $(date '+%Y-%m-%d %H:%M:%S')"
{ echo "$STAMP valu_mix: instruction mix of the headline step kernel (4096 envs, fast, 500 steps)"
bash tools/pmc.sh m1 SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 -- 4096 1080 fast 500
bash tools/pmc.sh m2 SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INST_CYCLES_SALU -- 4096 1080 fast 500
} > gpurun_out/valu_mix.log 2>&1 || exit 1
{ echo "$STAMP vmem_path: vector-memory path and L2 counters of the headline step kernel"
bash tools/pmc.sh t1 GRBM_GUI_ACTIVE TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum -- 4096 1080 fast 500
bash tools/pmc.sh t2 TA_FLAT_READ_WAVEFRONTS_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum -- 4096 1080 fast 500
bash tools/pmc.sh t3 TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum -- 4096 1080 fast 500
bash tools/l2_counters.sh final
} > gpurun_out/vmem_path.log 2>&1 || exit 1
bash tools/ablate_build.sh > gpurun_out/ablate_build.log 2>&1 || exit 1
{ echo "$STAMP ablate_phases: step time with K1 / the sweep compiled out (timing only)"; timeout -k 10 600 python3 tools/ablate_phases.py; } > gpurun_out/ablate_phases.log 2>&1 || exit 1
cat gpurun_out/valu_mix.log gpurun_out/vmem_path.log gpurun_out/ablate_phases.log
