for t in 1 0.3 0.1 0.01 0.001; do echo "prune_scale $t"; RT_COUNT=1 RT_PRUNE_SCALE=$t RT_STREAMS=1 RT_SIZES=800x480 timeout -k 10 100 python3 tools/trace_driver.py 3 2>&1 | tail -2 || exit 1; done
