#!/bin/bash
# scratch: sweep of the traversal thresholds (tools/trace_driver.py prints the frame's kernel time)
for t in 8 12 16 20 24; do
  for r in 8 12 16 24; do
    echo -n "tri_min $t refill $r: "
    RT_TRI_MIN_LANES=$t RT_REFILL_MIN_IDLE=$r RT_SIZES=800x480,1600x960 timeout -k 10 120 python tools/trace_driver.py 6 2>&1 | grep "^size" | sed 's/.*ms_kernels \([0-9.]*\) ms_trace(sum) \([0-9.]*\).*/\1 \2/' | tr '\n' ' '
    echo
  done
done
