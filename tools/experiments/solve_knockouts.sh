#!/bin/bash
# kernel time of the solver kernels with one stage knocked out (WRONG results: timing only)
for l in "" dk1 pol0 jac1; do
  if [ -n "$l" ]; then export EACHAM_HIP_LIB=$PWD/eacham_amd/lib/exp/libeacham_hip_$l.so; else unset EACHAM_HIP_LIB; fi
  echo "== ${l:-product}"
  python3 tests/rate_solve.py 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print({k:(v['kernels_ms'], v['end_to_end_ms']) for k,v in d.items() if isinstance(v,dict)})"
done
