set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
cp ray-tracer_amd/csrc/librt_mi355.so /tmp/librt_orig.so
echo base; timeout -k 10 300 python3 -u tools/trace_driver.py 8 2>&1 | grep "^size"
for N in 4 8; do cp tools/probe/librt_probe$N.so ray-tracer_amd/csrc/librt_mi355.so; echo "probe $N x 8 v_nop"; timeout -k 10 300 python3 -u tools/trace_driver.py 8 2>&1 | grep "^size"; done
cp /tmp/librt_orig.so ray-tracer_amd/csrc/librt_mi355.so
