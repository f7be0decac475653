#!/bin/bash
# round 4, fifteenth lease: cache policy of the epilogues' stores / operand loads (nt, sc1, sc0+nt+sc1): GEMM table A/B
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4r
mkdir -p $O
cd $R
for rep in 1 2; do
  for lib in "" tools/_ab/libreed_stnt.so tools/_ab/libreed_stsc1.so tools/_ab/libreed_stntsc.so tools/_ab/libreed_stldnt.so; do
    echo "lib=${lib:-product}" | tee -a $O/policy_ab.txt; REED_HIP_LIB=$lib timeout -k 10 200 python tools/gemm_table.py 256 20 | tee -a $O/policy_ab.txt || exit 1
  done
done
echo done
