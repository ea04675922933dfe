#!/bin/bash
# S200 RefineBA kernel stats for several group sizes of the landmark-major Schur stage: tools/ba_grp_rows.sh <tag> <rows> ...
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}" || exit 1
tag=$1; shift
for r in "$@"; do
  export EACHAM_BA_GROUP_ROWS=$r
  echo "== rows $r"
  python3 tools/ba_check.py 2>&1 | tail -2
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gr_${tag}_$r -- python3 tools/ba_only.py 5 > gpurun_out/gr_${tag}_$r.log 2>&1 || exit 1
  n=$(grep "inner iterations" gpurun_out/gr_${tag}_$r.log | awk '{print $3}')
  python3 tools/kernel_stats.py gpurun_out/gr_${tag}_$r $n | grep -E "groups|total"
done
