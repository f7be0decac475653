#!/usr/bin/env python
"""The backward's other one-round kernels while n CUs are held by something else (tools/_ab/hog.hip, see wgrad_under_hog.py): the
attention backward and forward (persistent: one workgroup per CU, items in a static stride) and an NN dgrad GEMM in its persistent
and one-shot forms (ops.set_concurrent_comm).  us per launch over 8 launches that start 3 ms after the hog.
usage: [HOG_NS=0,8,16,32] [HOG_LDS=16384] python tools/_ab/kernels_under_hog.py [b]"""
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
T, H, hd = 256, 16, 72
D = H * hd
M = b * T
qkv = (torch.randn(M, 3 * D, device=dev) * 0.5).to(torch.bfloat16)
o = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
do = torch.randn(M, D, device=dev).to(torch.bfloat16)
dqkv = torch.empty(M, 3 * D, dtype=torch.bfloat16, device=dev)
lse = torch.empty(b, H, T, device=dev)
ws = torch.empty(ops.attention_bwd_ws_floats(b, T, H), device=dev)
w = (torch.randn(3 * D, D, device=dev) / D ** 0.5).to(torch.bfloat16)
dx = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
ops.attention_fwd(qkv, o, lse, b, T, H, hd)


def gemm(comm):
    ops.set_concurrent_comm(comm)
    ops.gemm(ops.NN, ops.EPI_BF16, dqkv, w, M, D, 3 * D, dx, 3 * D, D, D)
    ops.set_concurrent_comm(False)


def attn_bwd(comm):
    ops.set_concurrent_comm(comm)
    ops.attention_bwd(qkv, o, do, lse, dqkv, b, T, H, hd, ws=ws)
    ops.set_concurrent_comm(False)


cases = [("attention backward", lambda: attn_bwd(False)),
         ("attention backward (beside a collective)", lambda: attn_bwd(True)),
         ("attention forward", lambda: ops.attention_fwd(qkv, o, lse, b, T, H, hd)),
         ("dgrad qkv, persistent", lambda: gemm(False)),
         ("dgrad qkv, one-shot (beside a collective)", lambda: gemm(True))]
sink = torch.zeros(4, dtype=torch.int32, device=dev)
side = torch.cuda.Stream()
NL = 8
LDS = int(os.environ.get("HOG_LDS", "16384"))
ns = [int(v) for v in os.environ.get("HOG_NS", "0,8,16,32").split(",")]
print(f"b = {b}: us per launch ({NL} launches, started 3 ms after the hog: 256 threads, {LDS} B of LDS per workgroup, 50 ms)")
print(f"{'CUs held':44s}" + "".join(f"{n:9d}" for n in ns))
for name, fn in cases:
    row = []
    for nh in ns:
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        if nh:
            rc = hog.hog_launch(nh, 5000000, LDS, sink.data_ptr(), side.cuda_stream)
            assert rc == 0, rc
            time.sleep(0.003)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(NL):
            fn()
        e1.record()
        torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) / NL * 1e3)
    print(f"{name:44s}" + "".join(f"{v:9.1f}" for v in row), flush=True)
