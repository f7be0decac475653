cd /root/repo
for b in 256 32; do
 for rep in 1 2; do
  for g in 0 1; do
   echo -n "b=$b opt_overlap=$g: "
   REED_OPT_OVERLAP=$g timeout -k 10 300 python bench.py --global-batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-table 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" || exit 1
  done
 done
done
