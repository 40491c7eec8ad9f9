#!/bin/bash
# per-layer times of the deep layers with the round-4 tile (arrangement 2 = 8 rows x 4 channel tiles) forced, next to the planner's choice
# usage: tools/unet_arr2.sh SIZE CASES
S=${1:-256}; N=${2:-8}
echo "== planner"; python tools/unet_layers.py $S $N bf16 | grep -v "^  enc0\|^  enc1\|^  dec1\|^  dec0"
echo "== arrangement 2 on enc2b enc3b enc4b dec3a dec3b dec2a dec2b (split by rule)"
PSM_UNET_FORCE="5:2:4:1,7:2:4:1,9:2:4:1,10:2:4:1,11:2:4:1,12:2:4:1,13:2:4:1" python tools/unet_layers.py $S $N bf16 | grep -v "^  enc0\|^  enc1\|^  dec1\|^  dec0"
echo "== arrangement 2 + split 2 on dec3a dec2a"
PSM_UNET_FORCE="10:2:4:2,12:2:4:2" python tools/unet_layers.py $S $N bf16 | grep -v "^  enc0\|^  enc1\|^  dec1\|^  dec0"
echo "== autotuned"; python tools/unet_layers.py $S $N bf16 autotune | grep -v "^  enc0\|^  enc1\|^  dec1\|^  dec0"
