#!/bin/bash
# Copy the evidence of one GPU-box visit (tools/profile_round.sh headline, tools/round4_evidence.sh, tools/round4_ab.sh, tools/launch_host.sh,
# tools/soak.py) from gpurun_out/ into profiles/roundN.  tools/evidence.py refuses every file that was measured on other kernel sources
# than the tree's.   publish_round.sh [profiles/round4]
cd "$(dirname "$0")/.."
dst=${1:-profiles/round5}
P=gpurun_out/profile
python3 tools/evidence.py publish "$dst" \
  $P/bench.json $P/bench_driver_shape.json $P/bench_config4_multi.json $P/bench_config1_circle.json \
  $P/kernel_stats.csv $P/kernel_stats_config4_multi.csv $P/kernel_stats_config1_circle.csv $P/kernel_stats_driver_shape_20steps.csv \
  $P/sq_latest.json $P/sq_config4_multi.json $P/sq_config1_circle.json \
  $P/traffic_latest.json $P/traffic_config4_multi.json $P/traffic_config1_circle.json \
  gpurun_out/valu_mix.log gpurun_out/vmem_path.log gpurun_out/ablate_phases.log gpurun_out/valu_by_phase.log \
  gpurun_out/ab_group_order.log gpurun_out/ab_pairs.log gpurun_out/ab_sectors.log gpurun_out/ab_occupancy.log \
  gpurun_out/launch_fixed.log gpurun_out/soak_r4.log
