#!/bin/bash
# the candidate pass's kernel (MFMA operands swapped, 27 VALU ops per 16 distances, top-2 values only) sweeping EVERY row of every pair:
# how fast is a whole sweep with that epilogue? (timing only)
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}" || exit 1
for dim in 256 128; do
EACHAM_EXP_ALL_CANDIDATES=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/allc_$dim -- python3 bench.py --steps 3 --warmup 1 --dim $dim --cpu-pairs 0 --ba-solves 0 --lines none > gpurun_out/allc_$dim.log 2>&1
python3 tools/kernel_stats.py gpurun_out/allc_$dim 4 | head -4
done
