#!/bin/bash
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_match_gpu.py tests/test_configs_gpu.py -x -q -m gpu 2>&1 | tail -3 || exit 1
for dim in 256 128; do
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --dim $dim --cpu-pairs 0 --ba-solves 0 --lines none 2>&1 | tail -1 | grep -oE '"value": [0-9.]+|"frac": [0-9.]+|avg_launch_ms": [0-9.]+|rror.*' | tr '\n' ' '; echo
done
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-pairs 0 --ba-solves 0 --lines c2,c3_tum,c5_kitti 2>&1 | grep -oE '"line": "[a-z0-9_]+"|"value": [0-9.]+|"frac": [0-9.]+' | tr '\n' ' '; echo
