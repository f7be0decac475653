#!/usr/bin/env python
"""Tiny workload for rocprofv3 --pmc passes: 3 launches of the implicit-GEMM 3x3 convolution (csrc/conv.hip) at the SD-VAE decoder's
largest layer shapes (batch 8): 128 -> 128 channels at 256 x 256, and 512 -> 512 at 64 x 64 with the fused nearest x2 upsampling."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reed_amd import ops
dev = torch.device("cuda")
prev = ops.use("fp16")
for (B, H, C, N, up) in ((8, 256, 128, 128, False), (8, 64, 512, 512, True)):
    a = (torch.randn(B, H, H, C, device=dev)).to(torch.float16)
    w = (torch.randn(N, 9 * C, device=dev) / (9 * C) ** 0.5).to(torch.float16)
    bias = torch.randn(N, device=dev)
    Ho = H * 2 if up else H
    out = torch.empty(B, Ho, Ho, N, device=dev)
    for _ in range(3):
        ops.conv3x3(a, w, bias, out, N, B, H, H, C, N, upsample=up)
torch.cuda.synchronize()
print("done")
