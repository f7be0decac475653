cd /root/repo
for p in 0 -1 0 -1; do
   echo -n "b=32 REED_WGRAD_PRIO=$p: "
   REED_WGRAD_PRIO=$p timeout -k 10 300 python bench.py --global-batch 32 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-table 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" || exit 1
done
