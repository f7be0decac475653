#!/bin/bash
# Round-2 measurement pass on the GPU box: full GPU tests, bench at b = 256 / 128 / 64 / 32, rocprofv3 kernel stats, PMC traffic of
# the dominant GEMM under two tile walks.  Outputs under gpurun_out/r2/ (copied into profiles/ afterwards).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r2
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q -p no:cacheprovider -s > $O/gputest.log 2>&1; echo "pytest rc=$?" | tee $O/gputest.rc; tail -3 $O/gputest.log
python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
for b in 128 64 32; do python bench.py --steps 10 --warmup 3 --global-batch $b --no-cpu-baseline > $O/bench_n1_b$b.json 2>> $O/bench_n1.err; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof256 --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-table > $O/bench_n1_under_rocprof.json 2> $O/rocprof256.err
rocprofv3 --kernel-trace --stats -d $O/prof32 --output-format csv -- python3 $R/bench.py --steps 8 --warmup 2 --global-batch 32 --no-cpu-baseline --no-kernel-table > $O/bench_n1_b32_under_rocprof.json 2> $O/rocprof32.err
for gm in 4 1; do
  REED_GEMM256_GM=$gm rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch_gm$gm --output-format csv -- python3 $R/tools/pmc_gemm.py > $O/pmc_fetch_gm$gm.log 2>&1
  REED_GEMM256_GM=$gm rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write_gm$gm --output-format csv -- python3 $R/tools/pmc_gemm.py > $O/pmc_write_gm$gm.log 2>&1
done
cd $R
for gm in 4 1; do echo "== GM=$gm"; python tools/pmc_summary.py $O/pmc_fetch_gm$gm gemm; python tools/pmc_summary.py $O/pmc_write_gm$gm gemm; done > $O/pmc_gemm_traffic.txt 2>&1
find $O -name "*kernel_stats.csv" | head
echo done
