#!/bin/bash
# ba_schur_dense cut after each of its phases (timing only, wrong results): tools/experiments/ab_dense_knockouts.sh <tag>
# builds: (cd eacham_amd/csrc && for k in 1 2 3 4 5 6; do make exp NAME=wd$k SRC=ba.hip EXPFLAGS=-DEXP_WD_STOP=$k; done)
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}" || exit 1
tag=$1
export EACHAM_BA_SCHUR=dense
for l in eacham_amd/lib/libeacham_hip.so eacham_amd/lib/exp/libeacham_hip_wd1.so eacham_amd/lib/exp/libeacham_hip_wd2.so eacham_amd/lib/exp/libeacham_hip_wd3.so eacham_amd/lib/exp/libeacham_hip_wd4.so eacham_amd/lib/exp/libeacham_hip_wd5.so eacham_amd/lib/exp/libeacham_hip_wd6.so; do
  [ -f $l ] || continue
  export EACHAM_HIP_LIB=$PWD/$l
  n=$(basename $l .so)
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/wd_${tag}_$n -- python3 tools/ba_window_times.py > gpurun_out/wd_${tag}_$n.log 2>&1 || { echo "$n failed"; continue; }
  echo "== $n" | tee -a gpurun_out/wd_${tag}.txt
  python3 tools/kernel_stats.py gpurun_out/wd_${tag}_$n 1 | grep "ba_schur_dense\|ba_assemble_dense" | tee -a gpurun_out/wd_${tag}.txt
done
