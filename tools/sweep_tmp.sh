for t in 1536 1024 768 512 384; do echo "small_cap $t"; RT_SMALL_CAP=$t RT_STREAMS=1 RT_SIZES=800x480 timeout -k 10 100 python3 tools/trace_driver.py 5 2>&1 | grep -E "^size" || exit 1; done
for t in 1536 1024 768; do echo "small_cap $t n<3M"; RT_SMALL_N=3000000 RT_SMALL_CAP=$t RT_STREAMS=1 RT_SIZES=800x480 timeout -k 10 100 python3 tools/trace_driver.py 5 2>&1 | grep -E "^size" || exit 1; done
