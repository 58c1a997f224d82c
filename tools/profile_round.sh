#!/bin/bash
# One profiling round on the GPU box: the bench line, a rocprofv3 kernel trace of the same bench command, and the counter passes.
# usage: tools/profile_round.sh <tag>     (outputs under gpurun_out/<tag>_*)
set -e
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
timeout -k 10 400 python3 -u bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -1 gpurun_out/${TAG}_bench.json | cut -c1-400
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/${TAG}_kt" -o bench -- python3 "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-frames-in-flight > "$R/gpurun_out/${TAG}_kt.log" 2>&1
cd "$R"
DB=$(find gpurun_out/${TAG}_kt -name "*.db" | head -1)
python3 tools/rocpd_summary.py "$DB" gpurun_out/${TAG}_kernel_stats.md
head -8 gpurun_out/${TAG}_kernel_stats.md | cut -c1-200
bash tools/pmc_passes.sh ${TAG}_pmc
python3 tools/pmc_summary.py gpurun_out/${TAG}_pmc > gpurun_out/${TAG}_pmc_summary.txt
