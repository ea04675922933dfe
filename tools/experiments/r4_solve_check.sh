#!/bin/bash
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_solve_gpu.py tests/test_score_gpu.py tests/test_twoview_cpp.py tests/test_sfm_loop_gpu.py -x -q -m gpu > gpurun_out/r4_solve_tests.log 2>&1
rc=$?
tail -25 gpurun_out/r4_solve_tests.log
[ $rc = 0 ] || exit $rc
bash tools/prof_solve.sh r4
cat gpurun_out/solve_rate_r4.json
