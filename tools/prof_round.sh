#!/bin/bash
# Everything profiles/<round>_* is built from, in one gpurun call: tools/prof_round.sh <tag>   (then, in the container:
#   python3 tools/pmc_traffic_json.py <tag> r03 ; python3 tools/pmc_sq_json.py 64 r03 <tag>sq ; copy the *_stats files)
# rocprofv3 passes are separate processes; --pmc is only ever combined with --kernel-trace.
tag=${1:?usage: tools/prof_round.sh <tag>}
here=$(dirname "$0")
"$here"/prof.sh "$tag" || exit 1
"$here"/pmc.sh ${tag}sq_a GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA || exit 1
"$here"/pmc.sh ${tag}sq_b GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVE_CYCLES || exit 1
"$here"/pmc.sh ${tag}sq_c GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY || exit 1
"$here"/prof_ba_ord.sh ${tag}_s200 auto > /dev/null || exit 1
"$here"/prof_ba_ord.sh ${tag}_c4 auto 500 100000 4 > /dev/null || exit 1
"$here"/prof_ba_windows.sh ${tag} > /dev/null || exit 1
echo "prof_round $tag done"
