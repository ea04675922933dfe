#!/bin/bash
# same-box A/B of two library builds on the 128-D shapes: tools/experiments/ab_128.sh <libA> <libB>
for rep in 1 2; do
  for lib in "$@"; do
    echo -n "$lib s200 128-D: "
    EACHAM_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 10 --warmup 3 --dim 128 --cpu-pairs 0 --ba-solves 0 --lines none 2>&1 | tail -1 | grep -oE '"value": [0-9.]+|"frac": [0-9.]+' | tr '\n' ' '; echo
  done
done
for lib in "$@"; do
  echo -n "$lib lines: "
  EACHAM_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-pairs 0 --ba-solves 0 --lines c5_kitti,c5_kitti_long 2>&1 | grep -oE '"line": "[a-z0-9_]+"|"value": [0-9.]+' | head -6 | tr '\n' ' '; echo
done
