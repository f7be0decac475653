#!/bin/bash
# round 4 evidence: rocprofv3 kernel stats of the bench (b = 256, b = 32) with the bench record of the SAME run, PMC passes of the
# shipped attention kernels and of the dominant GEMM (separate runs per counter group; the effective clock from GRBM_GUI_ACTIVE)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
echo "[$(date +%T)] kernel trace b=256"
rocprofv3 --kernel-trace --stats -d $O/prof256 --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-table --no-c3-leg --no-vae-leg --no-config-legs > $O/bench_n1_under_rocprof.json 2> $O/rocprof256.err
echo "[$(date +%T)] kernel trace b=32"
rocprofv3 --kernel-trace --stats -d $O/prof32 --output-format csv -- python3 $R/bench.py --steps 8 --warmup 2 --global-batch 32 --no-cpu-baseline --no-kernel-table --no-vae-leg --no-config-legs > $O/bench_n1_b32_under_rocprof.json 2> $O/rocprof32.err
cd $R
for d in prof256 prof32; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); cp $f $O/${d}_kernel_stats.csv; t=$(find $O/$d -name "*kernel_trace.csv" | head -1); python tools/timeline.py $t 4 > $O/${d}_timeline.txt 2>&1; done
rm -rf $O/prof256 $O/prof32
cut -c1-200 $O/bench_n1_under_rocprof.json
cd /tmp
echo "[$(date +%T)] pmc attention"
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_a --output-format csv -- python3 $R/tools/pmc_attn.py > $O/pmc_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS -d $O/pmc_b --output-format csv -- python3 $R/tools/pmc_attn.py > $O/pmc_b.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_f --output-format csv -- python3 $R/tools/pmc_attn.py > $O/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_w --output-format csv -- python3 $R/tools/pmc_attn.py > $O/pmc_w.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $O/pmc_g --output-format csv -- python3 $R/tools/pmc_attn.py > $O/pmc_g.log 2>&1
echo "[$(date +%T)] pmc gemm"
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/gemm_a --output-format csv -- python3 $R/tools/pmc_gemm.py > $O/gemm_a.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/gemm_f --output-format csv -- python3 $R/tools/pmc_gemm.py > $O/gemm_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/gemm_w --output-format csv -- python3 $R/tools/pmc_gemm.py > $O/gemm_w.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $O/gemm_g --output-format csv -- python3 $R/tools/pmc_gemm.py > $O/gemm_g.log 2>&1
cd $R
for d in pmc_a pmc_b pmc_f pmc_w pmc_g; do python tools/pmc_summary.py $O/$d attn; done > $O/pmc_attention.txt 2>&1
for d in gemm_a gemm_f gemm_w gemm_g; do python tools/pmc_summary.py $O/$d gemm; done > $O/pmc_gemm.txt 2>&1
# kernel durations of the PMC runs (for the clock: GRBM_GUI_ACTIVE / 8 / duration)
python - <<PY > $O/pmc_durations.txt 2>&1
import csv, glob, collections
for d in ("pmc_g", "gemm_g"):
    acc = collections.defaultdict(list)
    for f in glob.glob("$O/" + d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:70]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in acc.items():
        if "attn" in k or "gemm" in k:
            print(d, k, "mean ns", sum(v) / len(v), "n", len(v))
PY
cat $O/pmc_attention.txt | head -60; cat $O/pmc_durations.txt
rm -rf $O/pmc_a $O/pmc_b $O/pmc_f $O/pmc_w $O/pmc_g $O/gemm_a $O/gemm_f $O/gemm_w $O/gemm_g
echo done
