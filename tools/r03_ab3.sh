set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 900 python3 -u -m pytest tests/ -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/r03_ab_tests.log 2>&1 || { tail -30 gpurun_out/r03_ab_tests.log; exit 1; }
tail -2 gpurun_out/r03_ab_tests.log
RT_STREAMS=0,2,3 RT_SIZES=800x480,1600x960 timeout -k 10 300 python3 -u tools/trace_driver.py 8 2>&1 | grep "^size"
