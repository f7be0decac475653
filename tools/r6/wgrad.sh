#!/bin/bash
# Round 6, item 1: the grouped weight gradients re-dealt around tile rows per XCD (csrc/gemm256w.hip: TnGroupW) against round 5's
# build (tools/_ab/libreed_r5.so = `git archive 5cf404c reed_amd/csrc include` built with reed_amd/build.py's flags) on one box:
# tests, alternating launch times, in-kernel stamps (tools/_ab/libreed_clk.so = this tree with -DREED_CLK_PROBE), the PMC passes
# (FETCH_SIZE / WRITE_SIZE / GRBM_GUI_ACTIVE, separate passes), and the whole step alternating.
# usage (GPU box): bash tools/r6/wgrad.sh  ->  gpurun_out/r6w/*
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6w
mkdir -p $O
cd $R
echo "[$(date +%T)] tests"
timeout -k 10 900 python -m pytest tests/test_gemm_gpu.py -x -q -k "wgrad_group" > $O/tests.txt 2>&1 || { tail -40 $O/tests.txt; exit 1; }
tail -3 $O/tests.txt
echo "[$(date +%T)] launch times, alternating"
for rep in 1 2; do
  for lib in tools/_ab/libreed_r5.so ""; do
    echo "== lib=${lib:-current}"
    REED_HIP_LIB=$lib timeout -k 10 300 python tools/bench_wgrad_group.py 256 128 32 || exit 1
  done
done > $O/ab_launch.txt 2>&1
cat $O/ab_launch.txt
echo "[$(date +%T)] stamps"
{ REED_HIP_LIB=tools/_ab/libreed_clk.so timeout -k 10 300 python tools/_ab/clk_tn_w4.py 256 && REED_HIP_LIB=tools/_ab/libreed_clk.so timeout -k 10 300 python tools/_ab/clk_tn_w4.py 32; } > $O/stamps.txt 2>&1 || { tail -20 $O/stamps.txt; exit 1; }
cat $O/stamps.txt
echo "[$(date +%T)] pmc"
cd /tmp && export TMPDIR=/tmp
for tag in cur r5; do
  if [ $tag = r5 ]; then export REED_HIP_LIB=$R/tools/_ab/libreed_r5.so; else unset REED_HIP_LIB; fi
  for c in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    d=$O/pmc_${tag}_$(echo $c | cut -d' ' -f1)
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c -d $d --output-format csv -- python3 $R/tools/pmc_wgrad.py > $d.log 2>&1 || { tail -5 $d.log; exit 1; }
  done
done
unset REED_HIP_LIB
cd $R
for tag in cur r5; do
  echo "==== $tag"
  for c in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE SQ_BUSY_CYCLES; do python tools/pmc_summary.py $O/pmc_${tag}_$c ""; done
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$O/pmc_${tag}_GRBM_GUI_ACTIVE/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:70]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in acc.items():
    print("duration in the GRBM_GUI_ACTIVE pass:", k, sum(v) / len(v) / 1e3, "us", len(v))
PY
done > $O/pmc_wgrad.txt 2>&1
rm -rf $O/pmc_cur_* $O/pmc_r5_*
grep -v "^ *$" $O/pmc_wgrad.txt | grep -i "group\|reduce\|FETCH\|WRITE\|GRBM\|MFMA_BUSY\|duration\|====" | head -60
echo "[$(date +%T)] step, alternating"
for rep in 1 2; do
  for lib in tools/_ab/libreed_r5.so ""; do
    echo "== lib=${lib:-current}"
    REED_HIP_LIB=$lib timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-table --no-c3-leg --no-vae-leg --no-config-legs --no-loss-vs-ref 2>/dev/null | tail -1 | cut -c1-420 || exit 1
  done
done > $O/ab_step.txt 2>&1
cat $O/ab_step.txt
echo "[$(date +%T)] done"
