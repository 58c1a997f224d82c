#!/bin/bash
# one counter pass (instruction counts, lanes, waits) over the config-2 driver.  usage: tools/pmc_quick.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-pmc1}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD --output-format csv -d "$OUT/sq1" -o p -- python3 "$R/tools/trace_driver.py" 2 > "$OUT/sq1.log" 2>&1 || { tail -3 "$OUT/sq1.log"; exit 1; }
timeout -s KILL 150 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_INSTS_BRANCH --output-format csv -d "$OUT/sq2" -o p -- python3 "$R/tools/trace_driver.py" 2 > "$OUT/sq2.log" 2>&1 || { tail -3 "$OUT/sq2.log"; exit 1; }
cd $R && python3 tools/pmc_summary.py "$OUT" | awk '/== k_trace/{p=1} p&&/^==/&&!/k_trace/{p=0} p'
