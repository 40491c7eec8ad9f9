#!/bin/bash
# usage: tools/attic/conv_ab.sh ENVVAR -- every conv workload of bench.py with and without ENVVAR=1 (planner experiments)
for wl in unet unet_bf16 unet8 unet8_bf16 unet512_bf16; do
  for on in 0 1; do
    if [ $on = 1 ]; then export $1=1; else unset $1; fi
    python bench.py --workload $wl --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$wl $1=$on', round(d['ms_per_step']*1e3,1), 'us', [round(l['avg_us'],1) for l in d['roofline']['launches']])"
  done
done
