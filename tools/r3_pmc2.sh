#!/bin/bash
# PMC passes (separate runs per counter group) of the weight-gradient kernel's LDS side and of the implicit-GEMM convolution
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for w in gemm conv; do
  rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/${w}_a --output-format csv -- python3 $R/tools/pmc_$w.py > $O/${w}_a.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU -d $O/${w}_b --output-format csv -- python3 $R/tools/pmc_$w.py > $O/${w}_b.log 2>&1
done
cd $R
(for w in gemm conv; do for g in a b; do python tools/pmc_summary.py $O/${w}_$g $( [ $w = gemm ] && echo gemm_tn_group || echo conv3x3 ); done; done) > $R/gpurun_out/pmc2_summary.txt 2>&1
cat $O/gemm_a.log | tail -3; rm -rf $O
cat $R/gpurun_out/pmc2_summary.txt
