set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for TW in 0 1 2 4; do echo "tail window $TW"; RT_TAIL_WINDOW=$TW RT_SIZES=800x480,1600x960 timeout -k 10 300 python3 -u tools/trace_driver.py 8 2>&1 | grep "^size"; done
RT_TAIL_WINDOW=0 RT_STREAMS=2 timeout -k 10 300 python3 -u tools/trace_driver.py 8 2>&1 | grep "^size"
RT_TAIL_WINDOW=1 RT_STREAMS=2 timeout -k 10 300 python3 -u tools/trace_driver.py 8 2>&1 | grep "^size"
