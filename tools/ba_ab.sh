#!/bin/bash
# S200 RefineBA kernel stats with several builds of the library on the same box: tools/ba_ab.sh <tag> lib1.so lib2.so ...
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}" || exit 1
tag=$1; shift
i=0
for l in "$@"; do
  i=$((i+1))
  export EACHAM_HIP_LIB=$PWD/$l
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_${tag}_$i -- python3 tools/ba_only.py 5 > gpurun_out/ab_${tag}_$i.log 2>&1 || exit 1
  n=$(grep "inner iterations" gpurun_out/ab_${tag}_$i.log | awk '{print $3}')
  echo "== $l"
  python3 tools/kernel_stats.py gpurun_out/ab_${tag}_$i $n | tee -a gpurun_out/ab_${tag}.txt
done
