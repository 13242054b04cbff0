#!/bin/bash
# What a launch costs beyond its steps, on one box: events on the dispatch packet or recorded around it, with and without the fused
# metrics record, without a driver, and on a quarter of the envs.  (A polled wait instead of the blocked one was 7 us slower: removed.)
# Output: gpurun_out/launch_fixed.log (stamped with the kernel sources' hash).
set -e
cd "$(dirname "$0")/.."
out=gpurun_out/launch_fixed.log
mkdir -p gpurun_out
python3 tools/launch_fixed.py fast 4096 > $out 2>&1
FTGP_NO_HOST_SUM=1 python3 tools/launch_fixed.py fast 4096 >> $out 2>&1
FTGP_WAIT_SPIN=1 python3 tools/launch_fixed.py fast 4096 >> $out 2>&1
FTGP_LAUNCH_PLAIN=1 python3 tools/launch_fixed.py fast 4096 >> $out 2>&1
FTGP_LAUNCH_PLAIN=1 FTGP_NO_FUSED_METRICS=1 python3 tools/launch_fixed.py fast 4096 >> $out 2>&1
python3 tools/launch_fixed.py lobotomy 4096 >> $out 2>&1
python3 tools/launch_fixed.py fast 1024 >> $out 2>&1
python3 tools/evidence.py stamp $out
cat $out
