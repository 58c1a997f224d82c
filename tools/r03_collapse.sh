set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for M in 0 1 2; do echo "collapse mode $M"; RT_COLLAPSE_MODE=$M RT_COUNT=1 timeout -k 10 300 python3 -u tools/trace_driver.py 8 2>&1 | grep -E "counted|^size"; done
