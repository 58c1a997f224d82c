set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
RT_TRACE_DEBUG=1 timeout -k 10 200 python3 -u tools/trace_driver.py 1 2>&1 | grep -E "trace lanes closest|counted|^size" | head -3
RT_NO_CONE=1 RT_COUNT=1 timeout -k 10 200 python3 -u tools/trace_driver.py 4 2>&1 | grep -E "counted|^size" | head -3
