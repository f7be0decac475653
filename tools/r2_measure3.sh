#!/bin/bash
# Round-2 third measurement pass (after the host-side fixes and the persistent four-wave GEMM): full GPU tests, bench at
# b = 256 / 128 / 64 / 32, rocprofv3 kernel stats at b = 256 and 32, PMC traffic of the dominant kernels.  Outputs under
# gpurun_out/r2c/ (copied into profiles/ afterwards).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r2c
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q -p no:cacheprovider -s > $O/gputest.log 2>&1; echo "pytest rc=$?" | tee $O/gputest.rc; tail -3 $O/gputest.log
python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
for b in 128 64 32; do python bench.py --steps 10 --warmup 3 --global-batch $b --no-cpu-baseline > $O/bench_n1_b$b.json 2>> $O/bench_n1.err; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof256 --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-table > $O/bench_n1_under_rocprof.json 2> $O/rocprof256.err
rocprofv3 --kernel-trace --stats -d $O/prof32 --output-format csv -- python3 $R/bench.py --steps 8 --warmup 2 --global-batch 32 --no-cpu-baseline --no-kernel-table > $O/bench_n1_b32_under_rocprof.json 2> $O/rocprof32.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch --output-format csv -- python3 $R/tools/pmc_gemm.py > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write --output-format csv -- python3 $R/tools/pmc_gemm.py > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d $O/pmc_mfma --output-format csv -- python3 $R/tools/pmc_gemm.py > $O/pmc_mfma.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/pmc_lds --output-format csv -- python3 $R/tools/pmc_gemm.py > $O/pmc_lds.log 2>&1
cd $R
for d in pmc_fetch pmc_write pmc_mfma pmc_lds; do python tools/pmc_summary.py $O/$d gemm; done > $O/pmc_gemm.txt 2>&1
find $O -name "*kernel_stats.csv" | head
echo done
