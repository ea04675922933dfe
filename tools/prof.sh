#!/bin/bash
# Profiles the default bench (kernel trace + stats), then HBM traffic counters in separate --pmc passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass; --pmc is never combined with anything but --kernel-trace).
# usage: tools/prof.sh <tag>   -> gpurun_out/<tag>_{bench,fetch,write}; then run tools/pmc_traffic_json.py <tag> r02 HERE
tag=${1:?usage: tools/prof.sh <tag>}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_bench -- python3 bench.py --steps 2 --warmup 1 --lines s200_d128_i8,s200_d256_f32,c2,c3_tum,c4_ba,c5_kitti,c5_kitti_long > gpurun_out/${tag}_bench.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_fetch -- python3 bench.py --steps 1 --warmup 1 --cpu-pairs 0 --ba-solves 1 --lines none > gpurun_out/${tag}_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_write -- python3 bench.py --steps 1 --warmup 1 --cpu-pairs 0 --ba-solves 1 --lines none > gpurun_out/${tag}_write.log 2>&1 || exit 1
grep "^{\"metric\"" gpurun_out/${tag}_bench.log | cut -c1-300
