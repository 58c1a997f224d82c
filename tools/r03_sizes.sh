set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
RT_SIZES=400x240,800x480,1600x960,3200x1920 timeout -k 10 300 python3 -u tools/trace_driver.py 3 > gpurun_out/r03c_sizes.log 2>&1
cat gpurun_out/r03c_sizes.log
