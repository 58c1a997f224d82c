set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_parity.py tests/test_gpu_features.py tests/test_gpu_boundary.py -x -q -m gpu --timeout 180 --timeout-method thread > gpurun_out/r03f_tests.log 2>&1 || { tail -30 gpurun_out/r03f_tests.log; exit 1; }
tail -3 gpurun_out/r03f_tests.log
RT_NO_FUSE=1 RT_STREAMS=0,2 timeout -k 10 300 python3 -u tools/trace_driver.py 6 2>&1 | grep "^size"
RT_STREAMS=0,2,3 timeout -k 10 300 python3 -u tools/trace_driver.py 6 2>&1 | grep "^size"
RT_TIMELINE=1 timeout -k 10 300 python3 -u tools/trace_driver.py 3 > gpurun_out/r03f_tl.log 2>&1
tail -9 gpurun_out/r03f_tl.log
