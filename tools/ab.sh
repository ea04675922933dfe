#!/bin/bash
# usage: tools/ab.sh lib1.so lib2.so ...  — reduced bench with each library build (same box, same data)
for l in "$@"; do
  echo "== $l"
  EACHAM_HIP_LIB=$PWD/$l timeout -k 10 300 python bench.py --steps 5 --warmup 2 --frames 64 --landmarks 16000 --cpu-pairs 0 --ba-solves 0 2>&1 | tail -1 | grep -oE '"value": [0-9.]+|"frac": [0-9.]+|avg_launch_ms": [0-9.]+|rror.*' | tr '\n' ' '; echo
done
