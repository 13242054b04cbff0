#!/bin/bash
# What does one more SCALAR-side instruction per march iteration cost?  Diagnostic builds (never shipped) with 8 fillers of one kind per
# iteration, timed with cycle counters on the headline workload (tools/ab_pmc.sh): scalar ALU, nop, waitcnt, an untaken branch.
cd "${GRAFT_REPO_ROOT:-.}"
B="-DFTGP_SWEEP_V1 -DFTGP_NO_GUARD"
bash tools/ab_pmc.sh "s_base:$B" "s_add:$B -DFTGP_PAD_SALU=8" "s_nop:$B -DFTGP_PAD_SALU=8 -DFTGP_PAD_SALU_KIND=1" "s_and64:$B -DFTGP_PAD_SALU=8 -DFTGP_PAD_SALU_KIND=2" \
  "s_wait:$B -DFTGP_PAD_SALU=8 -DFTGP_PAD_SALU_KIND=3" "s_branch:$B -DFTGP_PAD_SALU=8 -DFTGP_PAD_SALU_KIND=4" "s_mov:$B -DFTGP_PAD_SALU=8 -DFTGP_PAD_SALU_KIND=5"
