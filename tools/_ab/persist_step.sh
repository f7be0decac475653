cd /root/repo
for b in 256 128; do
for rep in 1 2; do
for v in 0 1; do
   echo -n "b=$b REED_GEMM_PERSIST=$v: "
   REED_GEMM_PERSIST=$v timeout -k 10 300 python bench.py --global-batch $b --steps 12 --warmup 4 --no-cpu-baseline --no-kernel-table 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['final_loss'])" || exit 1
done
done
done
