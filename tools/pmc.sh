#!/bin/bash
# usage: [PMC_DIM=128] tools/pmc.sh <outdir> <counters...> -- runs the reduced bench under rocprofv3 --pmc
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}" || exit 1
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/$out -- python3 bench.py --steps 2 --warmup 1 --frames 64 --landmarks 16000 --dim ${PMC_DIM:-256} --cpu-pairs 0 --ba-solves 1 --lines none > gpurun_out/$out.log 2>&1
echo "exit $?"
