#!/bin/bash
# whole GPU suite + the default bench + the two self-starting forms of N > 1 (rehearsed on the one GPU of the box)
set -o pipefail
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r4_gpu_suite.log 2>&1
rc=$?
tail -8 gpurun_out/r4_gpu_suite.log
[ $rc = 0 ] || exit $rc
timeout -k 10 600 python bench.py --steps 10 --warmup 3 > gpurun_out/r4_bench.log 2> gpurun_out/r4_bench.err || { tail -5 gpurun_out/r4_bench.err; exit 1; }
tail -1 gpurun_out/r4_bench.log | cut -c1-3000
timeout -k 10 300 python bench.py --single-process --gpus 1 --steps 5 --warmup 2 --ba-solves 5 > gpurun_out/r4_bench_sp.log 2>&1 || { tail -5 gpurun_out/r4_bench_sp.log; exit 1; }
tail -1 gpurun_out/r4_bench_sp.log | cut -c1-1500
timeout -k 10 400 python bench.py --gpus 2 --backend gloo --all-on-device 0 --steps 3 --warmup 1 --ba-solves 3 --cpu-pairs 0 --lines c5_kitti > gpurun_out/r4_bench_2r.log 2>&1 || { tail -8 gpurun_out/r4_bench_2r.log; exit 1; }
tail -1 gpurun_out/r4_bench_2r.log | cut -c1-1500
