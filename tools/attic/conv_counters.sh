#!/bin/bash
# usage: tools/attic/conv_counters.sh TAG [workload] -- where the waves of the conv kernels spend their cycles (SQ counters,
# own passes with --kernel-trace only), for the batched convolutional run
TAG=$1; WL=${2:-unet8}
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/cc_$TAG; mkdir -p $O
cd /tmp
rocprofv3 -L > $O/counters_list.txt 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p1 -- python3 $R/bench.py --workload $WL --steps 20 --warmup 5 --no-cpu-baseline > $O/p1.log 2>&1; echo "p1 rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES --kernel-trace --output-format csv -d $O/p2 -- python3 $R/bench.py --workload $WL --steps 20 --warmup 5 --no-cpu-baseline > $O/p2.log 2>&1; echo "p2 rc=$?"
cd $R
python3 tools/attic/conv_counters_summary.py $TAG
