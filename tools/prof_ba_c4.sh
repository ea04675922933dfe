#!/bin/bash
# kernel stats of config-4 (500 cameras / 100k landmarks / 1M observations) RefineBA solves: tools/prof_ba_c4.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c4_$1 -- python3 tools/ba_only.py 3 500 100000 > gpurun_out/prof_c4_$1.log 2>&1 || exit 1
n=$(grep "inner iterations" gpurun_out/prof_c4_$1.log | awk '{print $3}')
python3 tools/kernel_stats.py gpurun_out/prof_c4_$1 $n | tee gpurun_out/prof_c4_$1.txt
