"""What HBM gives a plain streaming kernel on this box (torch copy / fill / sum of 2 GiB): the ceiling the row kernels are held against."""
import torch
dev = torch.device("cuda:0")
n = 1 << 29
a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
b = torch.empty_like(a)
def t(fn, bytes_, name):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{name}: {ms:.3f} ms, {bytes_ / ms / 1e9:.2f} TB/s")
t(lambda: b.copy_(a), 8 * n, "copy (4 B read + 4 B write)")
t(lambda: b.fill_(1.0), 4 * n, "fill (4 B write)")
t(lambda: a.sum(), 4 * n, "sum (4 B read)")
t(lambda: torch.add(a, b, out=b), 12 * n, "add (8 B read + 4 B write)")
