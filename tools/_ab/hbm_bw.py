import torch
dev = torch.device("cuda")
for mb in (256, 1024, 4096):
    n = mb * 1024 * 1024 // 4
    a = torch.randn(n, device=dev); b = torch.empty_like(a)
    for _ in range(3): b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"copy {mb} MB: {ms*1e3:.1f} us, {2*n*4/ms/1e9:.2f} TB/s (read+write)")
    e0.record()
    for _ in range(20): a.mul_(1.0001)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"in-place scale {mb} MB: {ms*1e3:.1f} us, {2*n*4/ms/1e9:.2f} TB/s")
    e0.record()
    for _ in range(20): s = a.sum()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"sum (read only) {mb} MB: {ms*1e3:.1f} us, {n*4/ms/1e9:.2f} TB/s")
