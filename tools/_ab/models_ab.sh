cd /root/repo
for m in SiT-L/2 SiT-B/2 SiT-S/2; do
 for g in 0 1; do
   echo -n "$m group=$g: "
   REED_WGRAD_GROUP=$g timeout -k 10 300 python bench.py --model $m --global-batch 256 --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-table 2>gpurun_out/models.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['final_loss'])" || tail -3 gpurun_out/models.err
 done
done
