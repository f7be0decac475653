cd /root/repo
for b in 32 64 128; do
 for r in 1.18 1.26 1.34 1.18 1.26 1.34; do
   echo -n "b=$b rate=$r: "
   REED_GEMM256_RATE=$r timeout -k 10 300 python bench.py --global-batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-table 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" || exit 1
 done
done
