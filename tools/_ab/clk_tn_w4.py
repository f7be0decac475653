import ctypes, os, statistics, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from reed_amd import _lib, ops
b = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda")
D, Hm, T = 1152, 4608, 256
shapes = [(D, Hm), (Hm, D), (D, D), (3 * D, D)]
M = b * T
probs = []
for n_out, k_in in shapes:
    dy = (torch.randn(M, n_out, device=dev) * 0.05).to(torch.bfloat16)
    x = (torch.randn(M, k_in, device=dev) * 0.05).to(torch.bfloat16)
    out = torch.zeros(n_out * k_in + n_out, device=dev)
    probs.append((dy, x, out[:n_out * k_in].view(n_out, k_in), out[n_out * k_in:], n_out, k_in))
flop = sum(2.0 * M * n * k for n, k in shapes)
L = _lib.load("bf16")
rd = L.reed_clk_probe_read; rd.restype = ctypes.c_int; rd.argtypes = [ctypes.c_void_p, ctypes.c_int]
ops.wgrad_group(probs, M); torch.cuda.synchronize()
t0 = time.perf_counter()
while time.perf_counter() - t0 < 2.0:
    for _ in range(10): ops.wgrad_group(probs, M)
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ops.wgrad_group(probs, M)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
n = 4096
buf = (ctypes.c_ulonglong * (8 * n))()
assert rd(buf, 8 * n) == 0
W = [[buf[8 * i + j] for j in range(8)] for i in range(n) if buf[8 * i + 1] > 0]
for mode in (0, 1, 2, 3, 4, 5):
    w = [x for x in W if x[3] == mode]
    if w:
        print(f"MODE {mode}: {len(w)} records, clock {statistics.median([x[0]/x[1]*0.1 for x in w]):.3f} GHz, {statistics.median([x[0]/x[2] for x in w]):.1f} cycles per K-tile of 64 (MFMA floor 2048 full tile), loop {statistics.median([x[1] for x in w])/100:.1f} us")
print(f"REED_WGRAD_W4={os.environ.get('REED_WGRAD_W4')}: {ms:.4f} ms per launch, {flop/ms/1e9:.1f} TF")
# timeline of the last launch: loop start / end of every recorded tile on the 100 MHz clock
t0 = min(x[4] for x in W)
ends = sorted((x[5] - t0) / 100.0 for x in W)
print(f"timeline: {len(W)} tiles recorded; last K loop ends {ends[-1]:.0f} us after the first one starts; K-loop start times (us): "
      + ", ".join(f"mode {m}: " + " ".join(f"{v:.0f}" for v in sorted((x[4] - t0) / 100.0 for x in W if x[3] == m)[::max(1, len([x for x in W if x[3] == m]) // 8)]) for m in (0, 1, 2, 3, 4, 5) if any(x[3] == m for x in W)))
print("K-loop end times (us), every 16th: " + " ".join(f"{v:.0f}" for v in ends[::16]))

# static items (records 1000 + item index): per mode, cycles per K-tile and the items' K-tile counts
items = [[buf[8 * i + j] for j in range(8)] for i in range(1000, n) if buf[8 * i + 1] > 0]
for mode in (1, 2, 3, 4, 5):
    w = [x for x in items if x[3] == mode]
    if w:
        print(f"static items MODE {mode}: {len(w)} items, {statistics.median([x[0]/x[2] for x in w]):.1f} cycles per K-tile, K-tiles per item {sorted(int(x[2]) for x in w)[:40]}, loop us {sorted(int(x[1]/100) for x in w)[:40]}")
if items:
    t0 = min(x[4] for x in W)
    print("static items: K-loop end times (us): " + " ".join(f"{(x[5]-t0)/100:.0f}" for x in sorted(items, key=lambda x: x[5])[::4]))

# full tiles by XCD (records 2000 + blockIdx): loop time per XCC
full = [[buf[8 * i + j] for j in range(8)] for i in range(2000, min(n, 2000 + 300)) if buf[8 * i + 1] > 0 and buf[8 * i + 3] == 0]
byx = {}
for x in full:
    byx.setdefault(int(x[6]), []).append(x[1] / 100.0)
for k in sorted(byx):
    v = sorted(byx[k])
    print(f"full tiles on XCC {k}: {len(v)} tiles, K loop {v[0]:.0f} .. {v[len(v)//2]:.0f} .. {v[-1]:.0f} us (blockIdx % 8 = {sorted(set(int(x[7]) % 8 for x in full if int(x[6]) == k))})")
