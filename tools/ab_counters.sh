#!/bin/bash
# Same-box A/B of prebuilt libraries with counters (one --pmc pass each, kernel-trace only): instructions per car-step by kind, the launch's
# shader cycles (cycles compare across boxes, times do not), what the waves wait for.
#   tools/ab_counters.sh out.log lib1.so lib2.so ...        (CASE="envs rays policy steps [cars] [track]" for another configuration)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/$1; shift
CASE="${CASE:-4096 1080 fast 300}"
set -- "$@"
echo "# kernel_source_sha=$(python3 tools/evidence.py sha) $(date '+%Y-%m-%d %H:%M:%S') ab_counters: CASE=$CASE libs: $*" > $out
for lib in "$@"; do
  tag=$(basename $lib .so)
  FTGP_LIB=$lib bash tools/pmc.sh abc_$tag SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD -- $CASE > gpurun_out/abc_$tag.txt 2>&1
  python3 - $tag $CASE >> $out <<'PY'
import re, sys
tag = sys.argv[1]; envs, rays, policy, steps = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5]); cars = int(sys.argv[6]) if len(sys.argv) > 6 else 1
c = {}
for line in open(f"gpurun_out/abc_{tag}.txt"):
    m = re.match(r"^([A-Za-z_0-9]+) ([0-9.e+]+)$", line.strip())
    if m: c[m.group(1)] = float(m.group(2))
ms = [l for l in open(f"gpurun_out/pmc_abc_{tag}.log").read().splitlines() if l.startswith("kernel ms")]
n = envs * cars * steps
if "SQ_INSTS_VALU" not in c: print(tag, "no counters:", open(f"gpurun_out/abc_{tag}.txt").read()[-300:]); raise SystemExit
print(f"{tag:24s} per car-step: valu {c['SQ_INSTS_VALU'] / n:7.1f} salu {c['SQ_INSTS_SALU'] / n:7.1f} branch {c['SQ_INSTS_BRANCH'] / n:6.1f} vmem_rd {c['SQ_INSTS_VMEM_RD'] / n:6.1f} | "
      f"shader cycles per step {c['GRBM_GUI_ACTIVE'] / 8 / steps:8.0f} | waves wait {c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']:.3f} of their life, for issue {c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']:.3f} | {ms[-1] if ms else ''}")
PY
done
cat $out
