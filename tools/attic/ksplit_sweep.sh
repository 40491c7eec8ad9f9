#!/bin/bash
# usage: tools/attic/ksplit_sweep.sh -- conv path, deepest split-K allowed by the planner (PSM_UNET_KSPLIT_MAX) per workload
for wl in unet512_bf16 unet_bf16 unet; do
  for ks in 8 4 2 1; do
    PSM_UNET_KSPLIT_MAX=$ks python bench.py --workload $wl --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$wl ksplit_max=$ks', round(d['ms_per_step']*1e3,1), 'us', [round(l['avg_us'],1) for l in d['roofline']['launches']])"
  done
done
