#!/bin/bash
# round 4, forty-first lease: rocprofv3 kernel stats of the step with the four-wave weight gradients and with the two-workgroup kernel, same box
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4X
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for w4 in 1 0 1 0; do
  REED_WGRAD_W4=$w4 rocprofv3 --kernel-trace --stats -d $O/p --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-table --no-c3-leg --no-vae-leg --no-config-legs > $O/b.json 2> $O/e.txt
  f=$(find $O/p -name "*kernel_stats.csv" | head -1)
  echo "REED_WGRAD_W4=$w4: $(cut -c1-75 $O/b.json | grep -o '"value": [0-9.]*')" | tee -a $O/interaction.txt
  python3 - "$f" <<'PY' | tee -a $O/interaction.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print(f"   {r['Name'][:70]:70s} {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:8.1f} us")
PY
  rm -rf $O/p
done
echo done
