"""In-kernel stamps of the grouped weight-gradient launch (diagnostic build: python tools/_ab/build_variant.py clk -DREED_CLK_PROBE;
REED_HIP_LIB=tools/_ab/libreed_clk.so python tools/_ab/clk_tn_w4.py [b ...]): per item MODE (0 = full 256^2 tile, 6 = 384x128, 7 = 128x384,
8 = bias only) the clock, the cycles per K-tile and the K loop's length; per XCC its clock, which blockIdx classes it ran and when its
items' K loops end (are they in step?).  Several sizes run one after the other in ONE process (KEEP=1: the earlier sizes' operands
stay allocated)."""
import ctypes, os, statistics, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from reed_amd import _lib, ops
dev = torch.device("cuda")
D, Hm, T = 1152, 4608, 256
shapes = [(D, Hm), (Hm, D), (D, D), (3 * D, D)]
L = _lib.load("bf16")
rd = L.reed_clk_probe_read; rd.restype = ctypes.c_int; rd.argtypes = [ctypes.c_void_p, ctypes.c_int]
med = statistics.median
keep = []


def probe(b):
    M = b * T
    probs = []
    for n_out, k_in in shapes:
        dy = (torch.randn(M, n_out, device=dev) * 0.05).to(torch.bfloat16)
        x = (torch.randn(M, k_in, device=dev) * 0.05).to(torch.bfloat16)
        out = torch.zeros(n_out * k_in + n_out, device=dev)
        probs.append((dy, x, out[:n_out * k_in].view(n_out, k_in), out[n_out * k_in:], n_out, k_in))
    if os.environ.get("KEEP") == "1":
        keep.append(probs)
    flop = sum(2.0 * M * n * k for n, k in shapes)
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
    W = [[buf[8 * i + j] for j in range(8)] for i in range(2000, 2000 + 256) if buf[8 * i + 1] > 0]
    print(f"b={b} REED_WGRAD_W4={os.environ.get('REED_WGRAD_W4')}: {ms:.4f} ms per launch, {flop/ms/1e9:.1f} TF; {len(W)} items recorded")
    t00 = min(x[4] for x in W)
    for mode in sorted(set(int(x[3]) for x in W)):
        w = [x for x in W if x[3] == mode]
        lp = sorted(x[1] / 100.0 for x in w)
        print(f"MODE {mode}: {len(w)} items, clock {med([x[0]/x[1]*0.1 for x in w]):.3f} GHz, {med([x[0]/x[2] for x in w]):.1f} cycles per K-tile of 64 "
              f"(MFMA floor 2048 for a full tile), K loop {lp[0]:.0f} .. {lp[len(lp)//2]:.0f} .. {lp[-1]:.0f} us")
    for k in sorted(set(int(x[6]) for x in W)):
        w = [x for x in W if int(x[6]) == k]
        parts = []
        for mode in sorted(set(int(x[3]) for x in w)):
            e = sorted((x[5] - t00) / 100.0 for x in w if x[3] == mode)
            parts.append(f"mode {mode} x{len(e)}: {e[0]:.0f}..{e[len(e)//2]:.0f}..{e[-1]:.0f}")
        f0 = [x for x in w if x[3] == 0]
        st = sorted((x[4] - t00) / 100.0 for x in w)
        print(f"XCC {k}: {len(w)} items (blockIdx % 8 = {sorted(set(int(x[7]) % 8 for x in w))}), K loops start {st[0]:.0f}..{st[-1]:.0f} us, full tiles: "
              f"clock {med([x[0]/x[1]*0.1 for x in f0]):.3f} GHz, {med([x[0]/x[2] for x in f0]):.0f} cycles per K-tile; K loops end (us after the first start) " + "; ".join(parts))


for b in [int(v) for v in sys.argv[1:]] or [256]:
    probe(b)
