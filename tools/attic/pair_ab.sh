#!/bin/bash
# usage: tools/attic/pair_ab.sh [SIZE CASES] -- the fused pairs of this build against a reference build (tools/_bin/libpsm_base.so, e.g. the previous
# round's library built from its commit) on ONE box, alternating runs: per-launch dispatch times of tools/unet_layers.py
SIZE=${1:-256}; CASES=${2:-8}
for rep in 1 2; do
  for lib in base new; do
    if [ $lib = base ]; then export PSM_LIB=$PWD/tools/_bin/libpsm_base.so; else unset PSM_LIB; fi
    python tools/unet_layers.py $SIZE $CASES bf16 2>&1 | awk -v tag="$lib$rep" '/ us |sum of launches/ { if ($0 ~ /sum of/) printf "%s  SUM %s\n", tag, $8; else printf "%s  %-18s %s us\n", tag, $1, $6 }'
  done
done
