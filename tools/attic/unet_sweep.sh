#!/bin/bash
# sweep of the conv-path planning knobs (diagnostic): fill target and deepest split-K
mkdir -p gpurun_out
for fill in 256 512 768; do for ks in 2 4 8; do
  for n in 1 8; do
    echo "fill=$fill ksmax=$ks n=$n: $(PSM_UNET_FILL=$fill PSM_UNET_KSPLIT_MAX=$ks timeout -k 10 100 python tools/attic/unet_bench.py 256 $n | head -1)"
  done
done; done > gpurun_out/sweep.log 2>&1
cat gpurun_out/sweep.log
