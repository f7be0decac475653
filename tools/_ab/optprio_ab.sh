cd /root/repo
python -c "
import torch
print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else None)
for p in (-1,0,1,2):
    try:
        s=torch.cuda.Stream(priority=p); print(p,'->',s.priority)
    except Exception as e: print(p,'error',e)
"
for b in 32 64; do
for rep in 1 2; do
for p in 0 1 -1; do
   echo -n "b=$b REED_OPT_PRIO=$p: "
   REED_OPT_PRIO=$p timeout -k 10 300 python bench.py --global-batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-table 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" || exit 1
done
done
done
