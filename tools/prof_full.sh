#!/bin/bash
# usage: tools/prof_full.sh TAG -- the profile set committed under profiles/:
#   kernel stats of the default bench command, the default bench line, and two PMC passes
#   (FETCH_SIZE, WRITE_SIZE; counters in their own runs with --kernel-trace only).
TAG=$1
export TMPDIR=/tmp; R=$PWD; export PSM_BENCH_LOGDIR=$R/gpurun_out/full_$1/detail_prof; O=$R/gpurun_out/full_$TAG; mkdir -p $O
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline --no-extras > $O/stats_bench.log 2>&1; echo "stats rc=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras > $O/fetch.log 2>&1; echo "fetch rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras > $O/write.log 2>&1; echo "write rc=$?"
cd $R
export PSM_BENCH_LOGDIR=$O/detail_bench; timeout -k 10 400 python bench.py > $O/bench.log 2>$O/bench.err; tail -1 $O/bench.log | cut -c1-200

# convolutional path: kernel stats + bench lines
export PSM_BENCH_LOGDIR=$O/detail_prof
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/unet_stats -- python3 $R/bench.py --workload unet8_bf16 --steps 100 --warmup 10 --no-cpu-baseline > $O/unet_stats.log 2>&1; echo "unet stats rc=$?"
cd $R
export PSM_BENCH_LOGDIR=$O/detail_unet; timeout -k 10 300 python bench.py --workload unet > $O/bench_unet.log 2>&1; tail -1 $O/bench_unet.log | cut -c1-160
export PSM_BENCH_LOGDIR=$O/detail_unet8; timeout -k 10 300 python bench.py --workload unet8 --no-cpu-baseline > $O/bench_unet8.log 2>&1; tail -1 $O/bench_unet8.log | cut -c1-160
export PSM_BENCH_LOGDIR=$O/detail_unet8_bf16; timeout -k 10 300 python bench.py --workload unet8_bf16 --no-cpu-baseline > $O/bench_unet8_bf16.log 2>&1; tail -1 $O/bench_unet8_bf16.log | cut -c1-160
python tools/pmc_summary.py $TAG
# the box returns gpurun_out/ only (<= 64 MiB): the condensed files go there, the raw traces are dropped
mkdir -p $R/gpurun_out/profiles_$TAG
cp $R/profiles/${TAG}_* $R/gpurun_out/profiles_$TAG/ 2>/dev/null
rm -rf $O/stats $O/fetch $O/write $O/unet_stats
