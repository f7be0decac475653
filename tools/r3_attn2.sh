#!/bin/bash
# Round 3: where the persistent key-stationary backward's time goes — timing with parts skipped (REED_ATTN_KSP_DBG: 1 = no
# phase A, 2 = no phase B products, 3 = neither: loads, waits, barriers and stores only) and PMC passes of the shipped kernels.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3d
mkdir -p $O
cd $R
for d in 0 1 2 3; do echo "dbg=$d"; REED_ATTN_KSP_DBG=$d timeout -k 10 120 python tools/time_attn.py 256 | tee -a $O/time_dbg.txt; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_a --output-format csv -- python3 $R/tools/pmc_attn.py > $O/pmc_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS -d $O/pmc_b --output-format csv -- python3 $R/tools/pmc_attn.py > $O/pmc_b.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_f --output-format csv -- python3 $R/tools/pmc_attn.py > $O/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_w --output-format csv -- python3 $R/tools/pmc_attn.py > $O/pmc_w.log 2>&1
cd $R
for d in pmc_a pmc_b pmc_f pmc_w; do python tools/pmc_summary.py $O/$d attn; done > $O/pmc_attention.txt 2>&1
cat $O/pmc_attention.txt
rm -rf $O/pmc_a $O/pmc_b $O/pmc_f $O/pmc_w
echo done
