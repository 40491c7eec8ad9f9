#!/bin/bash
# usage: tools/unet_hbm.sh TAG [workload = unet8_bf16] -- memory-side bytes per launch of the conv kernels at 8 cases per step (FETCH_SIZE and
# WRITE_SIZE in two separate --pmc passes with --kernel-trace only), condensed per kernel and grid into
# profiles/TAG_<workload>_hbm.csv with the achieved GB/s of each layer
TAG=$1; WL=${2:-unet8_bf16}
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/uh_$TAG; mkdir -p $O
cd /tmp
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/bench.py --workload $WL --steps 30 --warmup 5 --no-cpu-baseline > $O/fetch.log 2>&1; echo "fetch rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/bench.py --workload $WL --steps 30 --warmup 5 --no-cpu-baseline > $O/write.log 2>&1; echo "write rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --workload $WL --steps 30 --warmup 5 --no-cpu-baseline > $O/trace.log 2>&1; echo "trace rc=$?"
cd $R
python3 tools/unet_hbm_summary.py $TAG $WL
