set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
cp ray-tracer_amd/csrc/librt_mi355.so /tmp/librt_orig.so
echo base; RT_SIZES=800x480,1600x960 timeout -k 10 300 python3 -u tools/trace_driver.py 8 2>&1 | grep "^size"
for V in "$@"; do cp tools/probe/librt_$V.so ray-tracer_amd/csrc/librt_mi355.so; echo "variant $V"; RT_SIZES=800x480,1600x960 timeout -k 10 300 python3 -u tools/trace_driver.py 8 2>&1 | grep "^size"; done
cp /tmp/librt_orig.so ray-tracer_amd/csrc/librt_mi355.so
