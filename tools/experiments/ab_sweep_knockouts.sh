#!/bin/bash
# Knock-outs of match_sweep_kernel at 128-D and 256-D, same box (timing only: wrong results by construction; the parity gates of
# bench.py are expected to report bit_exact false for the knock-out builds): the full kernel, its MFMA chains + data movement without
# the top-2 epilogue (-DEXP_SWEEP_NO_EPI), its epilogue + data movement without the MFMAs (-DEXP_SWEEP_NO_MFMA).
#   make -C eacham_amd/csrc exp NAME=sweep_noepi SRC=matcher.hip EXPFLAGS=-DEXP_SWEEP_NO_EPI   (and sweep_nomfma)
#   gpurun tools/experiments/ab_sweep_knockouts.sh > gpurun_out/ab_sweep_ko.txt
cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}" || exit 1
for dim in 128 256; do
  for l in libeacham_hip.so exp/libeacham_hip_sweep_noepi.so exp/libeacham_hip_sweep_nomfma.so; do
    echo "== dim $dim  $l"
    EACHAM_HIP_LIB=$PWD/eacham_amd/lib/$l timeout -k 10 300 python bench.py --steps 5 --warmup 2 --dim $dim --lines none --cpu-pairs 0 --ba-solves 0 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('pairs/s %.0f  ms/step %.3f  sweep avg launch ms %.3f  sweep frac %.4f  frac_of_step %.4f' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], r['frac'], r['frac_of_step']))"
  done
done
