#!/bin/bash
# round 4: the backward's other one-round kernels while a stand-in holds CUs (tools/_ab/kernels_under_hog.py); the attention
# backward with 1 / 2 / 4 / 8 / 16 workgroups per CU (REED_ATTN_BWD_GRID)
set -e
mkdir -p gpurun_out/r4Z
for g in 1 2 4 8 16; do
  echo "REED_ATTN_BWD_GRID=$g" >> gpurun_out/r4Z/khog_grid.txt
  REED_ATTN_BWD_GRID=$g HOG_NS=0,16,0,16 timeout -k 10 200 python tools/_ab/kernels_under_hog.py 256 2>&1 | grep -v amdgpu.ids | grep "b = \|CUs held\|attention backward  " >> gpurun_out/r4Z/khog_grid.txt
done
cat gpurun_out/r4Z/khog_grid.txt
timeout -k 10 200 python tools/_ab/kernels_under_hog.py 256 > gpurun_out/r4Z/khog256.txt 2>&1
cat gpurun_out/r4Z/khog256.txt
timeout -k 10 200 python tools/_ab/kernels_under_hog.py 32 > gpurun_out/r4Z/khog32.txt 2>&1
cat gpurun_out/r4Z/khog32.txt
