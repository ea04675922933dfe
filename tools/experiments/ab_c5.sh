#!/bin/bash
for mode in same high; do
  for lines in c5_kitti c4_ba,c5_kitti c3_tum,c5_kitti c2,c5_kitti; do
    echo -n "stream2 $mode lines $lines: "
    EACHAM_STREAM2_PRIORITY=$mode timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-pairs 0 --ba-solves 0 --lines $lines 2>&1 | grep '"line": "c5_kitti"' | grep -oE '"value": [0-9.]+|"ms_per_step": [0-9.]+' | head -2 | tr '\n' ' '; echo
  done
done
