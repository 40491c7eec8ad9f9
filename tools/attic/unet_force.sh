#!/bin/bash
# usage: tools/attic/unet_force.sh N "force1" "force2" ... -- unet_bench with per-layer tile overrides (diagnostic)
N=$1; shift
mkdir -p gpurun_out; : > gpurun_out/force.log
for f in "$@"; do
  echo "=== FORCE=$f" >> gpurun_out/force.log
  PSM_UNET_FORCE="$f" timeout -k 10 100 python tools/attic/unet_bench.py 256 $N > gpurun_out/force_one.log 2>&1
  head -1 gpurun_out/force_one.log >> gpurun_out/force.log
  grep -E "enc|dec" gpurun_out/force_one.log | cut -c1-70 >> gpurun_out/force.log
done
cat gpurun_out/force.log
