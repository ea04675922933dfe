#!/bin/bash
# A/B of library builds on the same box: tools/experiments/ab_sweep.sh <lib>...  (headline 256-D and 128-D rates, twice each)
for rep in 1 2; do
  for lib in "$@"; do
    for dim in 256 128; do
      echo -n "$lib dim $dim: "
      EACHAM_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 10 --warmup 3 --dim $dim --cpu-pairs 0 --ba-solves 0 --lines none 2>&1 | tail -1 | grep -oE '"value": [0-9.]+|"frac": [0-9.]+|avg_launch_ms": [0-9.]+|rror.*' | tr '\n' ' '; echo
    done
  done
done
