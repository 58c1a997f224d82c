set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_parity.py tests/test_gpu_features.py tests/test_gpu_traversal.py tests/test_gpu_graze.py -x -q -m gpu --timeout 180 --timeout-method thread > gpurun_out/r03_ab_tests.log 2>&1 || { tail -30 gpurun_out/r03_ab_tests.log; exit 1; }
tail -2 gpurun_out/r03_ab_tests.log
RT_SIZES=800x480,1600x960 timeout -k 10 300 python3 -u tools/trace_driver.py 8 2>&1 | grep "^size"
