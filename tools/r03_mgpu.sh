#!/bin/bash
# new tests of this change on the GPU: multi-GPU C-ABI (one rank), torch nccl path (world size 1), no-alloc overlap guard
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_mgpu.py tests/test_zz_no_alloc.py tests/test_gpu_multirank.py -x -v --timeout 300 --timeout-method thread -m gpu > gpurun_out/r03_mgpu.log 2>&1
rc=$?
tail -25 gpurun_out/r03_mgpu.log
exit $rc
