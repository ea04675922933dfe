#!/bin/bash
# usage: tools/ab_full.sh lib1.so lib2.so ... — full-size S200 matching (no BA, no sub-lines) with each library build,
# interleaved twice on the same box (the kernel is power-bound: boxes differ by several per cent)
for round in 1 2; do
for l in "$@"; do
  echo "== $l (round $round)"
  EACHAM_HIP_LIB=$PWD/$l timeout -k 10 300 python bench.py --steps 10 --warmup 3 --cpu-pairs 0 --ba-solves 0 --lines none 2>&1 | tail -1 | grep -oE '"value": [0-9.]+|"frac": [0-9.]+|avg_launch_ms": [0-9.]+|rror.*' | tr '\n' ' '; echo
done
done
