#!/bin/bash
# the second stream's priority (= its hardware queue): tools/experiments/ab_stream2.sh
for rep in 1 2; do
  for mode in same high low; do
    for dim in 256 128; do
      echo -n "stream2 $mode dim $dim: "
      EACHAM_STREAM2_PRIORITY=$mode timeout -k 10 300 python bench.py --steps 10 --warmup 3 --dim $dim --cpu-pairs 0 --ba-solves 0 --lines none 2>&1 | tail -1 | grep -oE '"value": [0-9.]+|"frac": [0-9.]+|avg_launch_ms": [0-9.]+|rror.*' | tr '\n' ' '; echo
    done
  done
done
