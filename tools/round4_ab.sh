#!/bin/bash
# The same-box A/B comparisons DESIGN.md section 6 quotes for round 4, re-taken on the shipped library (every variant is an environment
# switch of ftgp_create, none needs another build): group order, opposite-ray pairs, direction sectors, occupancy, env-mate masks.
# Kernel time per step (HIP events), best of three 300-step launches each (tools/quick_perf.py); multi-car rows in bench.py's spawn rule.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export QUICK_SPAWN=1
hdr() { echo "# kernel_source_sha=$(python3 tools/evidence.py sha) $(date '+%Y-%m-%d %H:%M:%S') $1"; }
q() { python3 tools/quick_perf.py | sed "s/^libftgp.so/$(printf '%-34s' "$1")/"; }
{ hdr "ab_group_order: tasks in index order (car by car) vs long-first across the workgroup's cars"
  for i in 1 2; do FTGP_GROUP_ORDER_PLAIN=1 q "index order"; q "long-first (shipped)"; done; } > gpurun_out/ab_group_order.log 2>&1
{ hdr "ab_pairs: single groups vs pairs of opposite groups (tail = cheapest pairs sent as single groups)"
  for i in 1 2; do FTGP_NO_PAIRS=1 q "single groups"; FTGP_PAIR_TAIL=0 q "pairs, tail 0"; q "pairs, tail 2 (shipped)"; FTGP_PAIR_TAIL=4 q "pairs, tail 4"; done; } > gpurun_out/ab_pairs.log 2>&1
{ hdr "ab_sectors: direction sectors of the box field (shipped: 16 for >= 2048 cars, 64 below)"
  for i in 1 2; do for s in 64 32 16 8; do FTGP_SECTORS_RT=$s q "$s sectors"; done; done; } > gpurun_out/ab_sectors.log 2>&1
{ hdr "ab_occupancy: waves per workgroup (two workgroups per CU): 16 / 12 / 8 = 8 / 6 / 4 waves per SIMD"
  for w in 16 12 8; do FTGP_WAVES_PER_BLOCK=$w QUICK_CASES=0,1 q "$w waves per workgroup"; done; } > gpurun_out/ab_occupancy.log 2>&1
cat gpurun_out/ab_group_order.log gpurun_out/ab_pairs.log gpurun_out/ab_sectors.log gpurun_out/ab_occupancy.log
