#!/usr/bin/env python
"""The block's grouped weight gradients while n CUs are held by something else (tools/_ab/hog.hip: a stand-in for RCCL's channels
on a one-GPU box): gemm_tn.hip's two-workgroups-per-CU launch (force_tile 128) against gemm256w.hip's static one-workgroup-per-CU
form, and the default under ops.set_concurrent_comm(True) (must be the former).  ms per launch over 8 launches that start 3 ms
after the hog.  usage: [HOG_NS=0,8,16,32] [HOG_LDS=16384] python tools/_ab/wgrad_under_hog.py [b]"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from reed_amd import ops  # noqa: E402

_here = os.path.dirname(os.path.abspath(__file__))
if not os.path.exists(os.path.join(_here, "libhog.so")):   # built artefacts are not in history
    import subprocess
    subprocess.check_call(["hipcc", "-O2", "--offload-arch=gfx950", "-shared", "-fPIC", os.path.join(_here, "hog.hip"), "-o",
                           os.path.join(_here, "libhog.so")])
hog = ctypes.CDLL(os.path.join(_here, "libhog.so"))
hog.hog_launch.argtypes = [ctypes.c_int, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda")
b = int(sys.argv[1]) if len(sys.argv) > 1 else 256
D, Hm, T = 1152, 4608, 256
M = b * T
probs = []
for n_out, k_in in [(D, Hm), (Hm, D), (D, D), (3 * D, D)]:
    dy = (torch.randn(M, n_out, device=dev) * 0.05).to(torch.bfloat16)
    x = (torch.randn(M, k_in, device=dev) * 0.05).to(torch.bfloat16)
    out = torch.zeros(n_out * k_in + n_out, device=dev)
    probs.append((dy, x, out[:n_out * k_in].view(n_out, k_in), out[n_out * k_in:], n_out, k_in))
sink = torch.zeros(4, dtype=torch.int32, device=dev)
side = torch.cuda.Stream()
NL = 8
print(f"b = {b}: ms per grouped launch ({NL} launches, started 3 ms after the hog)")
print("CUs held | two workgroups per CU (gemm_tn.hip) | static, one per CU (gemm256w.hip) | default beside a collective")
LDS = int(os.environ.get("HOG_LDS", "16384"))
print(f"hog: 256 threads, {LDS} B of LDS per workgroup, 50 ms")
for nh in [int(v) for v in os.environ.get("HOG_NS", "0,8,16,32").split(",")]:
    row = []
    for tile, comm in ((128, False), (0, False), (0, True)):
        ops.gemm_force_tile(tile)
        ops.set_concurrent_comm(comm)
        for _ in range(3):
            ops.wgrad_group(probs, M)
        torch.cuda.synchronize()
        if nh:
            rc = hog.hog_launch(nh, 5000000, LDS, sink.data_ptr(), side.cuda_stream)   # 50 ms
            assert rc == 0, rc
            time.sleep(0.003)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(NL):
            ops.wgrad_group(probs, M)
        e1.record()
        torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) / NL)
    ops.gemm_force_tile(0)
    ops.set_concurrent_comm(False)
    print(f"{nh:8d} | {row[0]:8.3f} | {row[1]:8.3f} | {row[2]:8.3f}", flush=True)
