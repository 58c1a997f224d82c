#!/bin/bash
# shadow rays on a side stream (default) against one fused launch per level: frame hashes and times at two sizes, then the sensitive tests
set -o pipefail
for f in 0 1; do RT_SHADOW_STREAM=$f RT_SIZES=800x480,1600x960 timeout -k 10 300 python3 -u tools/trace_driver.py 8 2>&1 | grep -E "^size|^frame" || exit 1; done
timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-frames-in-flight 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('Mrays/s', d['value'], 'ms/frame', d['ms_per_step'], 'kernels', d['config']['kernels_ms_per_frame'], 'trace ms', d['roofline']['trace_ms_per_frame'], 'launches', d['roofline']['launches_per_frame'])" || exit 1
timeout -k 10 900 python -u -m pytest tests/test_gpu_graze.py tests/test_gpu_parity.py tests/test_gpu_boundary.py tests/test_gpu_features.py -x -q --timeout 300 --timeout-method thread -m gpu > gpurun_out/r03_side_tests.log 2>&1
rc=$?
tail -5 gpurun_out/r03_side_tests.log
exit $rc
