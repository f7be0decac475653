#!/bin/bash
# round 4: attention + GEMM GPU tests after the grid change of the attention backward beside a collective
set -e
mkdir -p gpurun_out/r4Z
timeout -k 10 900 python -m pytest tests/test_attention_gpu.py tests/test_cli_gpu.py -m gpu -x -q > gpurun_out/r4Z/tests_attn.txt 2>&1; tail -3 gpurun_out/r4Z/tests_attn.txt
