#!/bin/bash
for rep in 1 2; do
for mode in same high low; do
  echo -n "stream2 $mode: "
  EACHAM_STREAM2_PRIORITY=$mode timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-pairs 0 --ba-solves 0 --lines c2,c3_tum,c5_kitti 2>&1 | grep -oE '"line": "[a-z0-9_]+"|"value": [0-9.]+|"ms_per_step": [0-9.]+' | head -9 | tr '\n' ' '; echo
done
done
