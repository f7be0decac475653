#!/bin/bash
# Round 6: counters of the 256x288 kernel beside the 256x144 and the 256^2 eight-/four-wave kernels on the same GEMM (b = 32: fc1 forward)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/a --output-format csv -- python3 $R/tools/r6/t288.py ${B:-32} > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS -d $O/l --output-format csv -- python3 $R/tools/r6/t288.py ${B:-32} > $O/l.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY -d $O/v --output-format csv -- python3 $R/tools/r6/t288.py ${B:-32} > $O/v.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f --output-format csv -- python3 $R/tools/r6/t288.py ${B:-32} > $O/f.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $O/g --output-format csv -- python3 $R/tools/r6/t288.py ${B:-32} > $O/g.log 2>&1
cd $R
for d in a l v f g; do python tools/pmc_summary.py $O/$d gemm; done > $O/pmc288.txt 2>&1
python - <<PY >> $O/pmc288.txt 2>&1
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$O/g/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:70]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in acc.items():
    print("duration in the GRBM_GUI_ACTIVE pass:", k, sum(v) / len(v) / 1e3, "us", len(v))
PY
rm -rf $O/a $O/l $O/v $O/f $O/g
tail -5 $O/a.log
cat $O/pmc288.txt
