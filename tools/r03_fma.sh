#!/bin/bash
# release build after a traversal change: bench line, then the exactness tests that are most sensitive to the box test
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u bench.py --no-cpu-baseline 2>gpurun_out/r03_fma_bench.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('Mrays/s', d['value'], 'ms/frame', d['ms_per_step'], 'kernels', d['config']['kernels_ms_per_frame'], d['roofline']['per_ray'], 'trace ms', d['roofline']['trace_ms_per_frame'])" || exit 1
timeout -k 10 900 python -u -m pytest tests/test_gpu_graze.py tests/test_gpu_parity.py tests/test_gpu_boundary.py -x -q --timeout 300 --timeout-method thread -m gpu > gpurun_out/r03_fma_tests.log 2>&1
rc=$?
tail -8 gpurun_out/r03_fma_tests.log
exit $rc
