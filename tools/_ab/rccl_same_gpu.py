"""Does RCCL accept two ranks on one device?  (the answer decides what a one-GPU box can rehearse)"""
import os, sys, torch, torch.distributed as dist
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    x = torch.full((1024,), float(rank + 1), device="cuda")
    dist.all_reduce(x)
    torch.cuda.synchronize()
    print("rank", rank, "all_reduce ->", float(x[0]), flush=True)
    dist.destroy_process_group()
except Exception as e:
    print("rank", rank, "FAILED:", repr(e)[:600], flush=True)
    sys.exit(1)
