set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
RT_STREAMS=0,2,3,4 timeout -k 10 300 python3 -u tools/trace_driver.py 6 > gpurun_out/r03e_streams.log 2>&1
grep "^size" gpurun_out/r03e_streams.log
cd /tmp && export TMPDIR=/tmp
RT_STREAMS=2 timeout -k 10 300 rocprofv3 --kernel-trace -d "$R/gpurun_out/r03e_kt2" -o drv -- python3 "$R/tools/trace_driver.py" 3 > "$R/gpurun_out/r03e_kt2.log" 2>&1
DB=$(find $R/gpurun_out/r03e_kt2 -name "*.db" | head -1)
python3 $R/tools/rocpd_timeline.py "$DB" 70 > $R/gpurun_out/r03e_timeline2.txt
