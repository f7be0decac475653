#!/bin/bash
# round 4: the CU-hog experiment (tools/_ab/hog.hip holds CUs for 50 ms of s_memrealtime) over counts and LDS sizes of the stand-in:
# pass 1 = b 64, nine counts x three LDS sizes (gpurun_out/r4Z/hog64_*.txt); pass 2 (this form) = b 256 and b 32 at 16 KiB
set -e
mkdir -p gpurun_out/r4Z
HOG_NS=0,8,16,32,64,0 timeout -k 10 200 python tools/_ab/wgrad_under_hog.py 256 > gpurun_out/r4Z/hog256.txt 2>&1
cat gpurun_out/r4Z/hog256.txt
HOG_NS=0,8,16,32,64,0 timeout -k 10 200 python tools/_ab/wgrad_under_hog.py 32 > gpurun_out/r4Z/hog32.txt 2>&1
cat gpurun_out/r4Z/hog32.txt
