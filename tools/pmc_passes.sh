#!/bin/bash
# Hardware-counter passes over tools/trace_driver.py: one rocprofv3 run per pass (slot limits: 8 SQ, 4 TCC, 4 TCP, 2 TA, 2 GRBM),
# counters only (no trace domains).  usage: tools/pmc_passes.sh <out_dir_under_gpurun_out>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/${1:-pmc}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o p -- python3 "$R/tools/trace_driver.py" 2 > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$OUT/$name.log"; return 1; }
  echo "pass $name ok"
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD &&
run sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_INSTS_BRANCH &&
run tcp TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES GRBM_GUI_ACTIVE &&
run tcc TCC_HIT TCC_MISS TCC_EA0_RDREQ TCC_EA0_RDREQ_32B &&
run fetch FETCH_SIZE &&
run write WRITE_SIZE
find "$OUT" -name "*.csv" | head -20
