#!/bin/bash
# usage: tools/prof.sh TAG  -- GPU tests, rocprof kernel stats, bench line
TAG=$1
export TMPDIR=/tmp; R=$PWD; mkdir -p gpurun_out/prof_$TAG
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/gpu_tests.log
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 300 --warmup 50 --no-cpu-baseline > $R/gpurun_out/prof_bench.log 2>&1; echo "rocprof rc=$?"
cd $R; timeout -k 10 300 python bench.py --steps 3000 --warmup 300 --no-cpu-baseline > gpurun_out/bench_$TAG.log 2>&1; tail -1 gpurun_out/bench_$TAG.log | cut -c1-220
