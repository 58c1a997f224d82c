set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
RT_SIZES=1600x960 timeout -k 10 300 rocprofv3 --kernel-trace -d "$R/gpurun_out/r03i_kt" -o drv -- python3 "$R/tools/trace_driver.py" 3 > "$R/gpurun_out/r03i_kt.log" 2>&1
DB=$(find $R/gpurun_out/r03i_kt -name "*.db" | head -1)
python3 $R/tools/rocpd_timeline.py "$DB" 45 > $R/gpurun_out/r03i_timeline.txt
