#!/bin/bash
# rates and kernel stats of the robust-estimator building blocks: tools/prof_solve.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}" || exit 1
timeout -k 10 300 python3 tests/rate_solve.py > gpurun_out/solve_rate_$1.json 2> gpurun_out/solve_rate_$1.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_solve_$1 -- python3 tests/rate_solve.py > gpurun_out/prof_solve_$1.log 2>&1 || exit 1
python3 tools/kernel_stats.py gpurun_out/prof_solve_$1 1 | tee gpurun_out/prof_solve_$1.txt
