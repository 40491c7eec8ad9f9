#!/bin/bash
# usage: tools/prof_full.sh TAG -- the profile set committed under profiles/:
#   kernel stats of the default bench command, the default bench line, and two PMC passes
#   (FETCH_SIZE, WRITE_SIZE; counters in their own runs with --kernel-trace only).
TAG=$1
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/full_$TAG; mkdir -p $O
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline > $O/stats_bench.log 2>&1; echo "stats rc=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/fetch.log 2>&1; echo "fetch rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/write.log 2>&1; echo "write rc=$?"
cd $R
timeout -k 10 400 python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-200
python tools/pmc_summary.py $TAG
