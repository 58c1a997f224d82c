#!/bin/bash
# profiling round + lane-occupancy counters of the traversal kernel.  usage: tools/r03_prof.sh <tag>
set -o pipefail
TAG=${1:-r03x}
mkdir -p gpurun_out
RT_TRACE_DEBUG=1 timeout -k 10 300 python3 -u bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_dbg.json 2> gpurun_out/${TAG}_dbg.txt || { tail -5 gpurun_out/${TAG}_dbg.txt; exit 1; }
grep "^\[rt\]" gpurun_out/${TAG}_dbg.txt | head -12
bash tools/profile_round.sh ${TAG} || exit 1
cat gpurun_out/${TAG}_pmc_summary.txt | head -60
