#!/bin/bash
# The candidate pass beside the sweep, same box: the headline line and the 128-D line under the switches
#   EACHAM_EXP_SWEEP_PRIO=n     s_setprio n for the sweep's waves
#   EACHAM_EXP_STREAM2_CUS=n    the second stream restricted to n CUs
# tools/experiments/ab_candidate_pass.sh > gpurun_out/ab_cand.txt
cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}" || exit 1
run() {
  echo "== $*"
  env "$@" timeout -k 10 300 python bench.py --steps 10 --warmup 3 --lines s200_d128_i8 --cpu-pairs 0 --ba-solves 0 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('headline %.0f pairs/s  ms/step %.3f  sweep frac %.4f  frac_of_step %.4f  sweep avg ms %.3f | d128 %.0f frac %.4f | stream2 %s' % (d['value'], d['ms_per_step'], r['frac'], r['frac_of_step'], r['avg_launch_ms'], d['lines']['s200_d128_i8']['value'], d['lines']['s200_d128_i8']['frac'], d['second_stream']))"
}
run A=0
run EACHAM_EXP_SWEEP_PRIO=1
run EACHAM_EXP_SWEEP_PRIO=3
run EACHAM_EXP_STREAM2_CUS=32
run EACHAM_EXP_STREAM2_CUS=64
run EACHAM_EXP_STREAM2_CUS=128
run EACHAM_EXP_STREAM2_CUS=64 EACHAM_EXP_SWEEP_PRIO=2
run A=1
