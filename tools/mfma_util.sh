#!/bin/bash
# usage: tools/mfma_util.sh TAG -- MFMA utilisation per kernel from PMC counters (own pass, --kernel-trace only):
#   SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 1024 SIMDs), for the batched convolutional run and the default bench
TAG=$1
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/mfma_$TAG; mkdir -p $O
cd /tmp
timeout -k 10 300 rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $O/unet8 -- python3 $R/bench.py --workload unet8 --steps 30 --warmup 5 --no-cpu-baseline > $O/unet8.log 2>&1; echo "unet8 rc=$?"
timeout -k 10 300 rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $O/unet8b -- python3 $R/bench.py --workload unet8_bf16 --steps 30 --warmup 5 --no-cpu-baseline > $O/unet8b.log 2>&1; echo "unet8_bf16 rc=$?"
timeout -k 10 300 rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $O/unet64b -- python3 $R/bench.py --workload unet64_bf16 --steps 20 --warmup 5 --no-cpu-baseline > $O/unet64b.log 2>&1; echo "unet64_bf16 rc=$?"
timeout -k 10 300 rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $O/unet512b -- python3 $R/bench.py --workload unet512_bf16 --steps 30 --warmup 5 --no-cpu-baseline > $O/unet512b.log 2>&1; echo "unet512_bf16 rc=$?"
timeout -k 10 300 rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $O/pca -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --legs none > $O/pca.log 2>&1; echo "pca rc=$?"
timeout -k 10 300 rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $O/pca64 -- python3 $R/tools/configs_one.py 64 > $O/pca64.log 2>&1; echo "pca64 rc=$?"
cd $R
# the box returns gpurun_out/ only (<= 64 MiB): summarise there, keep the condensed file, drop the raw counter traces
python tools/mfma_summary.py $TAG
mkdir -p $R/gpurun_out/profiles_$TAG
cp $R/profiles/${TAG}_mfma_util.csv $R/gpurun_out/profiles_$TAG/
rm -rf $O/unet8 $O/unet8b $O/unet64b $O/unet512b $O/pca $O/pca64
