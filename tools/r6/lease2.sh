#!/bin/bash
# Round 6, lease 2: (a) the re-dealt weight gradients per XCC (clocks), the b = 128 launch time seen in lease 1; (b) verdict item 4's
# gate (tools/r6/overlap_gate.py); (c) the b = 32 step under the side-stream choices (verdict item 3b).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6b
mkdir -p $O
cd $R
echo "[$(date +%T)] tests (attention: its grids follow the CU reserve now)"
timeout -k 10 900 python -m pytest tests/test_attention_gpu.py -x -q > $O/tests_attn.txt 2>&1 || { tail -30 $O/tests_attn.txt; exit 1; }
tail -2 $O/tests_attn.txt
echo "[$(date +%T)] stamps"
{ REED_HIP_LIB=tools/_ab/libreed_clk.so timeout -k 10 300 python tools/_ab/clk_tn_w4.py 256 && REED_HIP_LIB=tools/_ab/libreed_clk.so timeout -k 10 300 python tools/_ab/clk_tn_w4.py 128; } > $O/stamps.txt 2>&1 || { tail -20 $O/stamps.txt; exit 1; }
cat $O/stamps.txt
echo "[$(date +%T)] b = 128 alone, then after b = 256"
{ timeout -k 10 300 python tools/bench_wgrad_group.py 128 && timeout -k 10 300 python tools/bench_wgrad_group.py 64 128 && timeout -k 10 300 python tools/bench_wgrad_group.py 256 128; } > $O/b128.txt 2>&1
cat $O/b128.txt
echo "[$(date +%T)] overlap gate"
timeout -k 10 600 python tools/r6/overlap_gate.py > $O/overlap_gate.txt 2>&1 || { tail -20 $O/overlap_gate.txt; exit 1; }
cat $O/overlap_gate.txt
echo "[$(date +%T)] b = 32 step: side-stream choices"
for rep in 1 2; do
  for cfg in "" "REED_WGRAD_W4=0" "REED_WGRAD_STREAM=0" "REED_WGRAD_STREAM=0 REED_WGRAD_W4=0"; do
    echo "== ${cfg:-default}"
    env $cfg timeout -k 10 300 python bench.py --global-batch 32 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-table --no-c3-leg --no-vae-leg --no-config-legs --no-loss-vs-ref 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'])" || exit 1
  done
done > $O/b32_side.txt 2>&1
cat $O/b32_side.txt
echo "[$(date +%T)] done"
