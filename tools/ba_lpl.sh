#!/bin/bash
# wall time per LM inner iteration for every lanes-per-landmark setting of the step's tail kernels: tools/ba_lpl.sh [cams landmarks seed]
for l in 1 2 4 8; do
  echo "== EACHAM_BA_LPL_STEP=$l"
  EACHAM_BA_LPL_STEP=$l timeout -k 10 200 python tools/ba_orderings.py 10 ${1:-200} ${2:-50000} ${3:-12345} 2>&1 | grep "^auto" | cut -c1-160
done
