cd /root/repo
for b in 64 32; do
 for eta in 0.92 0.86 0.80 0.92 0.86 0.80; do
   echo -n "b=$b eta=$eta: "
   REED_GEMM144_ETA=$eta timeout -k 10 300 python bench.py --global-batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-table 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" || exit 1
 done
done
