#!/bin/bash
# Knock-out measurement: the tile sweep with and without its column direction, S200 at 256-D and 128-D,
# interleaved twice on the same box. EACHAM_EXP_NO_COLTOP2 gives WRONG matches: timing only.
out=gpurun_out/r4_coltop.txt
: > $out
for round in 1 2; do
for dim in 256 128; do
for ko in 0 1; do
  if [ $ko = 1 ]; then export EACHAM_EXP_NO_COLTOP2=1; else unset EACHAM_EXP_NO_COLTOP2; fi
  echo "== dim $dim knockout $ko (round $round)" >> $out
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --dim $dim --cpu-pairs 0 --ba-solves 0 --lines none 2>&1 | tail -1 | grep -oE '"value": [0-9.]+|"frac": [0-9.]+|avg_launch_ms": [0-9.]+|rror.*' | tr '\n' ' ' >> $out; echo >> $out
done
done
done
cat $out
