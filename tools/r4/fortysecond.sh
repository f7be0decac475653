#!/bin/bash
# round 4: the grouped weight gradients while a stand-in holds 8 / 16 / 32 CUs (tools/_ab/hog.hip) — what a data-parallel backward
# does to a one-workgroup-per-CU static launch; then the gate's test and the GEMM tests
set -e
mkdir -p gpurun_out/r4Y
timeout -k 10 240 python tools/_ab/wgrad_under_hog.py 256 > gpurun_out/r4Y/hog256.txt 2>&1
timeout -k 10 120 python tools/_ab/wgrad_under_hog.py 32 > gpurun_out/r4Y/hog32.txt 2>&1
cat gpurun_out/r4Y/hog256.txt gpurun_out/r4Y/hog32.txt
timeout -k 10 500 python -m pytest tests/test_gemm_gpu.py -m gpu -x -q -k "wgrad" > gpurun_out/r4Y/tests.txt 2>&1; tail -3 gpurun_out/r4Y/tests.txt
