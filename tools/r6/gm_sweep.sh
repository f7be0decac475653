#!/bin/bash
# Round 6, verdict item 5: another XCD deal of the persistent 256^2 forward / dgrad kernel against the clock.
# tools/_ab/libreed_gm.so = this tree with -DREED_TILE_GM_ENV (python tools/_ab/build_variant.py gm -DREED_TILE_GM_ENV)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6gm
mkdir -p $O
export REED_HIP_LIB=$R/tools/_ab/libreed_gm.so
cd $R
for rep in 1 2; do for gm in default 2 4 5 8 16 32; do
  if [ $gm = default ]; then unset REED_TILE_GM; else export REED_TILE_GM=$gm; fi
  timeout -k 10 120 python tools/r6/gm_workload.py 2>&1 | grep "GM="
done; done > $O/times.txt
cat $O/times.txt
cd /tmp && export TMPDIR=/tmp
for gm in default 4 8 16; do
  if [ $gm = default ]; then unset REED_TILE_GM; else export REED_TILE_GM=$gm; fi
  for c in FETCH_SIZE GRBM_GUI_ACTIVE; do
    N=3 timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_${gm}_$c --output-format csv -- python3 $R/tools/r6/gm_workload.py > $O/pmc_${gm}_$c.log 2>&1 || { tail -3 $O/pmc_${gm}_$c.log; exit 1; }
  done
done
cd $R
for gm in default 4 8 16; do
  echo "==== GM=$gm"
  python - <<PY
import csv, glob, collections
val = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "GRBM_GUI_ACTIVE"):
    acc = collections.defaultdict(list)
    for f in glob.glob("$O/pmc_${gm}_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm256w" in r["Kernel_Name"]:
                acc[r["Kernel_Name"][22:75]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        val[k][c] = sum(v) / len(v)
dur = collections.defaultdict(list)
for f in glob.glob("$O/pmc_${gm}_GRBM_GUI_ACTIVE/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm256w" in r["Kernel_Name"]:
            dur[r["Kernel_Name"][22:75]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k in val:
    d = sum(dur[k]) / len(dur[k]) / 1e3
    print(f"{k}: FETCH_SIZE x 2 = {2 * val[k].get('FETCH_SIZE', 0) / 1e6:.3f} GB, {d:.1f} us in the clock pass, effective clock {val[k].get('GRBM_GUI_ACTIVE', 0) / 8 / d / 1e3:.3f} GHz")
PY
done > $O/pmc.txt 2>&1
cat $O/pmc.txt
rm -rf $O/pmc_*_FETCH_SIZE $O/pmc_*_GRBM_GUI_ACTIVE
