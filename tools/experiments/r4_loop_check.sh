#!/bin/bash
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_cpp_adapters.py tests/test_sfm_loop_gpu.py tests/test_tri_gpu.py -x -q -m gpu > gpurun_out/r4_loop_tests.log 2>&1
rc=$?
tail -12 gpurun_out/r4_loop_tests.log
[ $rc = 0 ] || exit $rc
SFM_DEBUG=1 timeout -k 10 300 python tools/sfm_loop_rate.py 100 600 6000 10 2>&1 | tail -12
timeout -k 10 600 python tools/sfm_loop_rate.py 500 600 30000 10 2>&1 | tail -4
