#!/bin/bash
# What changed in the second half of round 5, in one gpurun call: tools/prof_round5c.sh <tag>
#   solver rates + kernel stats (tests/rate_solve.py alone and under rocprofv3), and three kernel TIMELINES from --kernel-trace runs:
#   one added frame of the incremental loop, one local-window RefineBA call, one eacham_ba_prepare of S200.
# then, in the container: copy gpurun_out/<tag>_*.txt / solve_rate_<tag>.json into profiles/ (profiles/README.md, Round 5).
tag=${1:?usage: tools/prof_round5c.sh <tag>}
here=$(dirname "$0")
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}" || exit 1
"$here"/prof_solve.sh ${tag} > /dev/null || exit 1
python3 tools/sfm_loop_rate.py > gpurun_out/${tag}_loop_rate.txt 2>&1 || exit 1     # (a warm-up pass: the profiled run is then warm too)
A=$(SFM_PREPARE_ONLY=1 python3 tools/sfm_loop_rate.py | tail -1)
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${tag}_loop -- $A > gpurun_out/${tag}_loop.log 2>&1 || exit 1
python3 tools/loop_timeline.py gpurun_out/${tag}_loop > gpurun_out/${tag}_loop_frame_timeline.txt || exit 1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${tag}_win -- python3 tools/ba_window_times.py > gpurun_out/${tag}_win.log 2>&1 || exit 1
python3 tools/win_timeline.py gpurun_out/${tag}_win > gpurun_out/${tag}_ba_window_timeline.txt || exit 1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${tag}_prep -- python3 tools/prep_only.py s200 > gpurun_out/${tag}_prep.log 2>&1 || exit 1
python3 tools/prep_timeline.py gpurun_out/${tag}_prep > gpurun_out/${tag}_prepare_timeline.txt || exit 1
python3 tools/ba_window_times.py > gpurun_out/${tag}_window_times.txt 2>&1 || exit 1
echo "prof_round5c $tag done"
