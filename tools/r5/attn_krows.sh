#!/bin/bash
# Round 5, item 2 (iii): the ring backward's K tiles on 160-byte rows — tests, same-box A/B of the kernel against the build with
# 144-byte K rows (tools/_ab/libreed_k144.so = the previous commit), and the LDS bank-conflict counters of both.
# Output: gpurun_out/r5_attn_krows.txt
O=$PWD/gpurun_out/r5_attn_krows.txt
R=$PWD
mkdir -p gpurun_out
{
timeout -k 10 500 python -m pytest tests/test_attention_gpu.py -x -q 2>&1 | tail -3 || exit 1
for rep in 1 2 3; do
  for lib in tools/_ab/libreed_k144.so ""; do
    echo "lib=${lib:-product (K rows 160 B)}"; REED_HIP_LIB=$lib timeout -k 10 120 python tools/time_attn.py 32 256 2>/dev/null
  done
done
cd /tmp && export TMPDIR=/tmp
for lib in tools/_ab/libreed_k144.so ""; do
  tag=$([ -z "$lib" ] && echo k160 || echo k144)
  [ -n "$lib" ] && export REED_HIP_LIB=$R/$lib || unset REED_HIP_LIB
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT -d $R/gpurun_out/pmc_$tag --output-format csv -- python3 $R/tools/pmc_attn.py > /dev/null 2>&1
  echo "== LDS counters, $tag"; (cd $R && python tools/pmc_summary.py gpurun_out/pmc_$tag attn)
  rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $R/gpurun_out/pmcb_$tag --output-format csv -- python3 $R/tools/pmc_attn.py > /dev/null 2>&1
  echo "== busy counters, $tag"; (cd $R && python tools/pmc_summary.py gpurun_out/pmcb_$tag attn)
  rm -rf $R/gpurun_out/pmc_$tag $R/gpurun_out/pmcb_$tag
done
} > $O 2>&1
tail -40 $O
