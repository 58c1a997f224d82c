set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_ply.py tests/test_zz_no_alloc.py tests/test_gpu_scene_export.py -x -q -m gpu --timeout 300 --timeout-method thread 2>&1 | tail -5
