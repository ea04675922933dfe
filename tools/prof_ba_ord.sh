#!/bin/bash
# kernel stats of RefineBA solves under one elimination ordering: tools/prof_ba_ord.sh <tag> <ordering> [cams landmarks seed]
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}" || exit 1
export EACHAM_BA_ORDERING=$2
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ba_$1 -- python3 tools/ba_only.py 5 ${3:-200} ${4:-50000} ${5:-12345} > gpurun_out/prof_ba_$1.log 2>&1 || exit 1
n=$(grep "inner iterations" gpurun_out/prof_ba_$1.log | awk '{print $3}')
python3 tools/kernel_stats.py gpurun_out/prof_ba_$1 $n | tee gpurun_out/prof_ba_$1.txt
