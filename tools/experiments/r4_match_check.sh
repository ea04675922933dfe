#!/bin/bash
# matcher parity (both column forms, see tests/conftest.py) + S200 rates at 256-D / 128-D for several batch budgets
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_match_gpu.py tests/test_configs_gpu.py tests/test_pipeline_gpu.py tests/test_graph_gpu.py -x -q -m gpu > gpurun_out/r4_match_tests.log 2>&1
rc=$?
tail -15 gpurun_out/r4_match_tests.log
[ $rc = 0 ] || exit $rc
out=gpurun_out/r4_match_rates.txt
: > $out
for dim in 256 128; do
for mb in 1024 512 256 128; do
  export EACHAM_MATCH_BUDGET_MB=$mb
  echo "== dim $dim budget $mb MB" >> $out
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --dim $dim --cpu-pairs 0 --ba-solves 0 --lines none 2>&1 | tail -1 | grep -oE '"value": [0-9.]+|"frac": [0-9.]+|avg_launch_ms": [0-9.]+|rror.*' | tr '\n' ' ' >> $out; echo >> $out
done
done
unset EACHAM_MATCH_BUDGET_MB
cat $out
bash tools/prof_match.sh r4b
