#!/bin/bash
# A/B on one box: double-buffered epilogue patches in the four-wave 256^2 kernel (variant library) vs the shipped one
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
REED_HIP_LIB=$R/tools/_ab/libreed_epidb.so timeout -k 10 500 python -m pytest tests/test_gemm_gpu.py -q -x -m gpu -k "epilogues or many_tiles or nn_dgrad or nt_bias" 2>&1 | tail -2
for v in epidb base epidb base; do
  if [ $v = epidb ]; then export REED_HIP_LIB=$R/tools/_ab/libreed_epidb.so; else unset REED_HIP_LIB; fi
  timeout -k 10 300 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-c3-leg --no-vae-leg --no-config-legs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); t = d['gemm_family_isolated']['table']
print('$v', d['value'], d['ms_per_step'], ' '.join(f\"{r['kernel'].split()[0][0]}{r['kernel'].split()[1][:4]}:{r['tflops']:.0f}\" for r in t[:8]))"
done
