#!/bin/bash
# round 4, fourth lease: static wave priority in the attention kernels; the tile kernels on the fat-epilogue GEMM shapes
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4d
mkdir -p $O
cd $R
for d in 0 64 128; do
  echo "fwd prio dbg=$d"; REED_ATTN_FWD_DBG=$((d + 256)) timeout -k 10 120 python tools/time_attn.py 256 | cut -c1-60 | tee -a $O/fwd_prio.txt || exit 1
done
for d in 0 8 16 0 8; do
  echo "bwd prio dbg=$d"; REED_ATTN_KSP_DBG=$d timeout -k 10 120 python tools/time_attn.py 256 | cut -c60-400 | tee -a $O/bwd_prio.txt || exit 1
done
for t in 0 128 256 144 0; do
  echo "force tile $t:"; REED_FORCE_TILE=$t timeout -k 10 200 python tools/gemm_table.py 256 20 | tee -a $O/force_tile.txt || exit 1
done
echo done
