#!/bin/bash
# usage: TAG=r5 bash tools/profile.sh  — PMC passes of the dominant GEMMs (traffic, clock), kernel stats of the bench
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${TAG:-prof}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
echo "[$(date +%T)] pmc gemm"
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/gemm_a --output-format csv -- python3 $R/tools/pmc_gemm.py > $O/gemm_a.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/gemm_f --output-format csv -- python3 $R/tools/pmc_gemm.py > $O/gemm_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/gemm_w --output-format csv -- python3 $R/tools/pmc_gemm.py > $O/gemm_w.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $O/gemm_g --output-format csv -- python3 $R/tools/pmc_gemm.py > $O/gemm_g.log 2>&1
cd $R
for d in gemm_a gemm_f gemm_w gemm_g; do python tools/pmc_summary.py $O/$d ""; done > $O/pmc_gemm.txt 2>&1
python - <<PY >> $O/pmc_gemm.txt 2>&1
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$O/gemm_g/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:70]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in acc.items():
    print("duration in the GRBM_GUI_ACTIVE pass:", k, sum(v) / len(v) / 1e3, "us", len(v))
PY
rm -rf $O/gemm_a $O/gemm_f $O/gemm_w $O/gemm_g
cd /tmp
echo "[$(date +%T)] kernel trace b=256"
rocprofv3 --kernel-trace --stats -d $O/prof256 --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-table --no-c3-leg --no-vae-leg --no-config-legs --no-loss-vs-ref > $O/bench_n1_under_rocprof.json 2> $O/rocprof256.err
echo "[$(date +%T)] kernel trace b=32"
rocprofv3 --kernel-trace --stats -d $O/prof32 --output-format csv -- python3 $R/bench.py --steps 8 --warmup 2 --global-batch 32 --no-cpu-baseline --no-kernel-table --no-c3-leg --no-vae-leg --no-config-legs --no-loss-vs-ref > $O/bench_n1_b32_under_rocprof.json 2> $O/rocprof32.err
cd $R
for d in prof256 prof32; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); cp $f $O/${d}_kernel_stats.csv; t=$(find $O/$d -name "*kernel_trace.csv" | head -1); python tools/timeline.py $t 4 > $O/${d}_timeline.txt 2>&1; done
rm -rf $O/prof256 $O/prof32
cut -c1-160 $O/bench_n1_under_rocprof.json; cut -c1-160 $O/bench_n1_b32_under_rocprof.json
echo done
