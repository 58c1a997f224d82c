set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 1000 python3 -u -m pytest tests/ -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/r03_full_tests.log 2>&1 || { tail -40 gpurun_out/r03_full_tests.log; exit 1; }
tail -2 gpurun_out/r03_full_tests.log
timeout -k 10 300 python3 -u bench.py --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
