#!/bin/bash
# Everything profiles/r05_* is built from, in two gpurun calls (each under the 20-minute limit):
#   tools/prof_round5.sh <tag> a    bench under --kernel-trace --stats, FETCH_SIZE / WRITE_SIZE passes, SQ passes of the sweep at 256-D
#   tools/prof_round5.sh <tag> b    SQ passes at 128-D, BA kernel stats (S200, config 4, local windows), solver rates and stats
# then, in the container:  python3 tools/pmc_traffic_json.py <tag> r05 ; python3 tools/pmc_sq_json.py 64 r05 <tag>sq ;
#   python3 tools/pmc_sq_json.py 64 r05 <tag>sq128 match_sweep_kernel _d128 128 ; copy the *_stats files (profiles/README.md, Round 5)
# rocprofv3 passes are separate processes; --pmc is only ever combined with --kernel-trace.
tag=${1:?usage: tools/prof_round5.sh <tag> a|b}
here=$(dirname "$0")
if [ "$2" = "a" ]; then
  "$here"/prof.sh "$tag" || exit 1
  "$here"/pmc.sh ${tag}sq_a GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA || exit 1
  "$here"/pmc.sh ${tag}sq_b GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVE_CYCLES || exit 1
  "$here"/pmc.sh ${tag}sq_c GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY || exit 1
else
  export PMC_DIM=128
  "$here"/pmc.sh ${tag}sq128_a GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA || exit 1
  "$here"/pmc.sh ${tag}sq128_b GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVE_CYCLES || exit 1
  "$here"/pmc.sh ${tag}sq128_c GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY || exit 1
  unset PMC_DIM
  "$here"/prof_ba_ord.sh ${tag}_s200 auto > /dev/null || exit 1
  "$here"/prof_ba_ord.sh ${tag}_c4 auto 500 100000 4 > /dev/null || exit 1
  "$here"/prof_ba_windows.sh ${tag} > /dev/null || exit 1
  "$here"/prof_solve.sh ${tag} > /dev/null || exit 1
fi
echo "prof_round5 $tag $2 done"
