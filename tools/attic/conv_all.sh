#!/bin/bash
# usage: tools/attic/conv_all.sh -- every conv workload of bench.py (autotuned plan), one line each
for wl in unet unet_bf16 unet8 unet8_bf16 unet512_bf16; do
  python bench.py --workload $wl --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$wl', round(d['ms_per_step']*1e3,1), 'us', d['config'].get('planner') or '', [round(l['avg_us'],1) for l in d['roofline']['launches']])"
done
