set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
RT_STREAMS=0,2 RT_SIZES=800x480,1600x960 timeout -k 10 300 python3 -u tools/trace_driver.py 8 2>&1 | grep "^size"
