for t in 8 16 24 32 40; do echo "tri_min $t"; RT_TRI_MIN_LANES=$t RT_STREAMS=1 RT_SIZES=200x120,800x480,1600x960 timeout -k 10 100 python3 tools/trace_driver.py 3 2>&1 | tail -3 || exit 1; done
