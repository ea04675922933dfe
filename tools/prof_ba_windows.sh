#!/bin/bash
# kernel stats of the local-window RefineBA calls (TUM stand-in): tools/prof_ba_windows.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_win_$1 -- python3 tools/ba_window_times.py > gpurun_out/prof_win_$1.log 2>&1 || exit 1
python3 tools/kernel_stats.py gpurun_out/prof_win_$1 $(grep -o "[0-9.]* inner iterations" gpurun_out/prof_win_$1.log | head -1 | awk "{printf \"%d\", \$1*40*2+0.5}") | tee gpurun_out/prof_win_$1.txt
