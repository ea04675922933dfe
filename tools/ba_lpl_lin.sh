#!/bin/bash
# wall time per LM inner iteration for every lanes-per-landmark setting of the linearisation: tools/ba_lpl_lin.sh [cams landmarks seed]
for l in 1 2 4 8; do
  echo "== EACHAM_BA_LPL_LIN=$l"
  EACHAM_BA_LPL_LIN=$l timeout -k 10 200 python tools/ba_orderings.py 10 ${1:-200} ${2:-50000} ${3:-12345} 2>&1 | grep "^auto" | cut -c1-170
done
