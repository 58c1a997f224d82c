set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 300 python3 -u bench.py --no-cpu-baseline > gpurun_out/r03a_bench.json 2> gpurun_out/r03a_bench.err
tail -1 gpurun_out/r03a_bench.json | cut -c1-300
RT_TRACE_DEBUG=1 timeout -k 10 200 python3 -u tools/trace_driver.py 3 > gpurun_out/r03a_dbg.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/r03a_kt" -o drv -- python3 "$R/tools/trace_driver.py" 3 > "$R/gpurun_out/r03a_kt.log" 2>&1
cd $R
DB=$(find gpurun_out/r03a_kt -name "*.db" | head -1)
python3 tools/rocpd_timeline.py "$DB" 40 > gpurun_out/r03a_timeline.txt
tail -5 gpurun_out/r03a_dbg.log
