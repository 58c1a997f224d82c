set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 900 python3 -u -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k defining_size -s --timeout 800 --timeout-method thread > gpurun_out/r03_c5_test.log 2>&1 || { tail -40 gpurun_out/r03_c5_test.log; exit 1; }
tail -8 gpurun_out/r03_c5_test.log
