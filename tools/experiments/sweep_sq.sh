#!/bin/bash
# SQ counter passes for the row sweep (match_sweep_kernel): tools/experiments/sweep_sq.sh <tag>
tag=${1:?tag}
here=$(dirname "$0")/..
"$here"/pmc.sh ${tag}sq_a GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA || exit 1
"$here"/pmc.sh ${tag}sq_b GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVE_CYCLES || exit 1
"$here"/pmc.sh ${tag}sq_c GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY || exit 1
"$here"/pmc.sh ${tag}sq_d GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT || exit 1
"$here"/pmc.sh ${tag}sq_e GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_LDS || exit 1
echo done
