#!/bin/bash
# Round 3 evidence pass at HEAD: printed measurements of the new parity tests, bench at b = 256 / 128 / 64 / 32, rocprofv3 kernel
# stats at b = 256 and 32, PMC passes of the shipped attention kernels and of the dominant kernel.  Outputs: gpurun_out/r3m/
# (the summaries are copied into profiles/r3_*).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3m
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_fp32_gpu.py tests/test_vae_gpu.py "tests/test_model_gpu.py::test_long_horizon_heun_cfg_drift_xl2" "tests/test_model_gpu.py::test_long_horizon_heun_cfg_drift_s2" -q -s -m gpu 2>&1 | grep -E "drift|fp32\]|fp32 samplers|SD-VAE|HIP fp32|REF fp32|passed|failed" > $O/parity_prints.txt; cat $O/parity_prints.txt | cut -c1-260
for b in 128 64 32; do python bench.py --steps 10 --warmup 3 --global-batch $b --no-cpu-baseline --no-c3-leg > $O/bench_n1_b$b.json 2>> $O/bench.err; done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_n1.json 2>> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof256 --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-table --no-c3-leg > $O/bench_n1_under_rocprof.json 2> $O/rocprof256.err
rocprofv3 --kernel-trace --stats -d $O/prof32 --output-format csv -- python3 $R/bench.py --steps 8 --warmup 2 --global-batch 32 --no-cpu-baseline --no-kernel-table > $O/bench_n1_b32_under_rocprof.json 2> $O/rocprof32.err
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_a --output-format csv -- python3 $R/tools/pmc_attn.py > $O/pmc_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS -d $O/pmc_b --output-format csv -- python3 $R/tools/pmc_attn.py > $O/pmc_b.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_f --output-format csv -- python3 $R/tools/pmc_attn.py > $O/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_w --output-format csv -- python3 $R/tools/pmc_attn.py > $O/pmc_w.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmcg_f --output-format csv -- python3 $R/tools/pmc_gemm.py > $O/pmcg_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmcg_w --output-format csv -- python3 $R/tools/pmc_gemm.py > $O/pmcg_w.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d $O/pmcg_m --output-format csv -- python3 $R/tools/pmc_gemm.py > $O/pmcg_m.log 2>&1
cd $R
for d in prof256 prof32; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); cp $f $O/${d}_kernel_stats.csv; done
for d in pmc_a pmc_b pmc_f pmc_w; do python tools/pmc_summary.py $O/$d attn; done > $O/pmc_attention.txt 2>&1
for d in pmcg_f pmcg_w pmcg_m; do python tools/pmc_summary.py $O/$d gemm_tn; done > $O/pmc_gemm.txt 2>&1
rm -rf $O/prof256 $O/prof32 $O/pmc_a $O/pmc_b $O/pmc_f $O/pmc_w $O/pmcg_f $O/pmcg_w $O/pmcg_m
for f in $O/bench_n1.json $O/bench_n1_b128.json $O/bench_n1_b64.json $O/bench_n1_b32.json $O/bench_n1_under_rocprof.json $O/bench_n1_b32_under_rocprof.json; do python3 -c "
import json,sys
d=json.load(open(sys.argv[1])); r=d.get('roofline') or {}
print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['step_mfma_frac'], r.get('achieved'), r.get('frac'), r.get('avg_ms_per_launch'), (d.get('c3_per_gpu_leg') or {}).get('images_per_sec_per_gpu'))" $f; done
cat $O/pmc_gemm.txt | head -20
echo done
