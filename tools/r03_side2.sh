#!/bin/bash
set -o pipefail
timeout -k 10 600 python -u -m pytest tests/test_gpu_boundary.py tests/test_abi.py tests/test_cpp_host.py -x -q --timeout 300 --timeout-method thread -m gpu > gpurun_out/r03_side_tests.log 2>&1
rc=$?
tail -15 gpurun_out/r03_side_tests.log
exit $rc
