#!/bin/bash
# usage: tools/ab_bound_sweep.sh — the matching lines with the exact row sweep (EACHAM_MATCH_SWEEP_FORM=exact) and with the bound form +
# exact pass over the rows left open (=bound) at every dimension, same box, same data
for mode in exact bound; do
  echo "== $mode"
  export EACHAM_MATCH_SWEEP_FORM=$mode
  timeout -k 10 500 python bench.py --steps 3 --warmup 1 --cpu-pairs 0 --ba-solves 0 --lines s200_d128_i8,c2,c3_tum,c5_kitti,c5_kitti_long > gpurun_out/ab_bound_$mode.log 2>&1 || { tail -5 gpurun_out/ab_bound_$mode.log; exit 1; }
  python3 tools/bench_lines.py gpurun_out/ab_bound_$mode.log
  grep -o '"parity": {[^}]*}' gpurun_out/ab_bound_$mode.log | head -8 | cut -c1-120
done
