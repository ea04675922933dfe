#!/bin/bash
# Profiles the default bench (kernel trace + stats), then HBM traffic counters in separate --pmc passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass; --pmc is never combined with anything but --kernel-trace).
# usage: tools/prof.sh [round-tag]   (writes under gpurun_out/, then run tools/pmc_traffic_json.py etc. here)
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -- python3 bench.py --steps 2 --warmup 1 > gpurun_out/prof_bench.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 1 --warmup 1 --cpu-pairs 0 --ba-solves 1 --lines none > gpurun_out/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 1 --warmup 1 --cpu-pairs 0 --ba-solves 1 --lines none > gpurun_out/pmc_write.log 2>&1 || exit 1
tail -1 gpurun_out/prof_bench.log | cut -c1-300
