#!/bin/bash
# usage: tools/ab_full.sh lib1.so lib2.so ...  — full S200 matching bench with each library build
for l in "$@"; do
  echo "== $l"
  EACHAM_HIP_LIB=$PWD/$l timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-pairs 0 --ba-solves 0 2>&1 | tail -1 | grep -oE '"value": [0-9.]+|"frac": [0-9.]+|avg_launch_ms": [0-9.]+|finalize_ms_per_step": [0-9.]+|rror.*' | tr '\n' ' '; echo
done
