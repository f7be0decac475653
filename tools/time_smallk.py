#!/usr/bin/env python
"""Time the small-K weight gradients (final layer, patch embed) at b=256 and check them against torch."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reed_amd import ops
dev = torch.device("cuda"); M, D = 65536, 1152
hb = (torch.randn(M, D, device=dev)).to(torch.bfloat16); dl = torch.randn(M, 32, device=dev).to(torch.bfloat16)
dx = torch.randn(M, D, device=dev); xb = torch.randn(M, 16, device=dev).to(torch.bfloat16)
ws = torch.empty(ops.smallk_ws_floats(D, 32), device=dev)
gw = torch.zeros(32 * D, device=dev); gb = torch.zeros(D, device=dev); gs = torch.zeros(32, device=dev)
def t(fn, it=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it
print("final  (bf16 wide, KS=32): %.3f ms" % t(lambda: ops.smallk_wgrad(hb, False, dl, ws, gw, None, gs, M, D, 32, 1, False)))
print("pembed (f32 wide,  KS=16): %.3f ms" % t(lambda: ops.smallk_wgrad(dx, True, xb, ws, gw, gb, None, M, D, 16, 0, False)))
ref = hb.float().t() @ dl.float()
ops.smallk_wgrad(hb, False, dl, ws, gw, None, gs, M, D, 32, 1, False)
print("max err final:", (gw.view(32, D).t() - ref).abs().max().item(), "ref max", ref.abs().max().item())
