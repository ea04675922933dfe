#!/bin/bash
# kernel trace of the S200 matching step alone (no BA, no sub-lines): tools/prof_match.sh <tag> [dim]
tag=${1:?usage: tools/prof_match.sh <tag> [dim]}
dim=${2:-256}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_match -- python3 bench.py --steps 4 --warmup 2 --dim $dim --cpu-pairs 0 --ba-solves 0 --lines none > gpurun_out/${tag}_match.log 2>&1 || exit 1
python3 tools/kernel_stats.py gpurun_out/${tag}_match 6 > gpurun_out/${tag}_match_stats.txt
cat gpurun_out/${tag}_match_stats.txt
tail -1 gpurun_out/${tag}_match.log | cut -c1-400
