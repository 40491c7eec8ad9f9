#!/bin/bash
# usage: tools/conv_arr_sweep.sh OUTDIR -- round 6, verdict item 1(b): the deep layers (enc2a .. dec2b, conv indices 4-13) with the larger
# per-wave register blocks (arrangements 2-4 of psm_unet.h: fewer ds_read_b128 per MFMA) against the shipped 8 x 16 px x 32 channel tile,
# at 64 and at 8 cases per step; parity of every forced plan first (tests/test_unet.py, bf16 layer-by-layer test).
OUT=${1:-gpurun_out/arr}; mkdir -p $OUT
force() { local arr=$1 nct=$2 s=""; for l in 4 5 6 7 8 9 10 11 12 13; do s="$s$l:$arr:$nct:1,"; done; echo "${s%,}"; }
for tag in "2 4" "3 2" "4 4"; do
  set -- $tag
  F=$(force $1 $2)
  PSM_UNET_FORCE="$F" timeout -k 10 300 python -m pytest tests/test_unet.py -q -x -m gpu -k "bf16_matches_bf16_oracle or straddle" > $OUT/parity_arr$1.log 2>&1 || { tail -5 $OUT/parity_arr$1.log; exit 1; }
  tail -1 $OUT/parity_arr$1.log
done
for n in 64 8; do
  timeout -k 10 200 python tools/unet_layers.py 256 $n bf16 > $OUT/base_$n.txt 2>&1 || exit 1
  for tag in "2 4" "3 2" "4 4"; do
    set -- $tag
    PSM_UNET_FORCE="$(force $1 $2)" timeout -k 10 200 python tools/unet_layers.py 256 $n bf16 > $OUT/arr$1_$n.txt 2>&1 || exit 1
  done
done
python - $OUT <<'PY'
import sys, re, os
out = sys.argv[1]
def rd(f):
    d = {}
    for ln in open(os.path.join(out, f)):
        m = re.match(r"\s+(\S+)\s+plan=\[(.*?)\]\s+([\d.]+) us", ln)
        if m: d[m.group(1)] = (float(m.group(3)), m.group(2))
    return d
for n in (64, 8):
    base = rd(f"base_{n}.txt"); alts = {a: rd(f"arr{a}_{n}.txt") for a in (2, 3, 4)}
    print(f"--- {n} cases: us per launch  shipped | arr2 (16x4ct, 8 reads / 16 MFMA) | arr3 (16x2ct, 6/8) | arr4 (8x4ct, 6/8)")
    for k, (us, plan) in base.items():
        print(f"{k:18s} [{plan:12s}] {us:7.2f} | " + " | ".join(f"{alts[a].get(k, (float('nan'), ''))[0]:7.2f} [{alts[a].get(k, (0, ''))[1]}]" for a in (2, 3, 4)))
PY
