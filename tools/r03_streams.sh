set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for NS in 2 3; do
RT_STREAMS=$NS timeout -k 10 300 rocprofv3 --kernel-trace -d "$R/gpurun_out/r03b_kt$NS" -o drv -- python3 "$R/tools/trace_driver.py" 3 > "$R/gpurun_out/r03b_kt$NS.log" 2>&1
DB=$(find $R/gpurun_out/r03b_kt$NS -name "*.db" | head -1)
python3 $R/tools/rocpd_timeline.py "$DB" 70 > $R/gpurun_out/r03b_timeline$NS.txt
tail -1 $R/gpurun_out/r03b_kt$NS.log
done
