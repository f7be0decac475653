#!/bin/bash
# Round 5, item 3: which tile kernel the block's GEMMs want at the 8-GPU shape (b = 32 per GPU; also 64): the heuristic's choice
# against every forced kernel (144 = 256x144 ring, 257 / 258 = 256^2 four-wave one-shot / persistent, 256 = 256^2 eight-wave).
# Output: gpurun_out/r5_b32_tiles.txt
O=gpurun_out/r5_b32_tiles.txt
mkdir -p gpurun_out
{
for b in 32 64; do
  for rep in 1 2; do
    for t in 0 144 258 257 256; do
      echo "b=$b tile=$t"; REED_FORCE_TILE=$t timeout -k 10 120 python tools/gemm_table.py $b 50 2>/dev/null | tail -1
    done
  done
done
} > $O 2>&1
tail -12 $O
