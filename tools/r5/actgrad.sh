#!/bin/bash
# Round 5, item 1: the derivative-saving epilogues (EPI_GELU_G / EPI_SILU_G / EPI_MUL) — tests, then a same-box A/B of the step
# against the recomputing form (--save-act-grad 0), alternating.  Output: gpurun_out/r5_actgrad.txt
# (profiles/r5_actgrad.txt was made at commit 0a480b8 through an environment switch; since round 6 the same A/B runs through
# bench.py --save-act-grad {0,1}.)
set -o pipefail
O=gpurun_out/r5_actgrad.txt
mkdir -p gpurun_out
{
python -m pytest tests/test_gemm_gpu.py -x -q -k "test_epilogues or test_nn_dgrad or test_fused_epilogues or test_many_tiles" 2>&1 | tail -3 &&
python -m pytest tests/test_model_gpu.py -x -q -s -k "tiny_vs_reference or saved_activation or c2_xl2_trajectory or fp16_training_gradients or c4_xl2 or b2_alignment" 2>&1 | grep -v "^$" | tail -60 &&
python -m pytest tests/test_fp32_gpu.py -x -q 2>&1 | tail -3 &&
for rep in 1 2; do
  for v in 0 1; do
    echo "== b=256 --save-act-grad $v"
    python bench.py --save-act-grad $v --steps 10 --warmup 3 --no-cpu-baseline --no-config-legs --no-vae-leg --no-loss-vs-ref 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value',d['value'],'ms',d['ms_per_step'],'dom',d['roofline']['avg_ms_per_launch'],'c3',d.get('c3_per_gpu_leg',{}).get('images_per_sec_per_gpu'))
for r in d.get('gemm_family_isolated',{}).get('table',[]): print('   ',r)
"
  done
done
} > $O 2>&1
echo rc=$? >> $O
tail -5 $O
