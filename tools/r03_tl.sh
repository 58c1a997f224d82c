set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
RT_TIMELINE=1 timeout -k 10 300 python3 -u tools/trace_driver.py 3 > gpurun_out/r03d_tl.log 2>&1
tail -12 gpurun_out/r03d_tl.log
