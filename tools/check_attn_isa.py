#!/usr/bin/env python
"""attn_fwd256p_kernel's and attn_fwd256v_kernel's waits are counted (s_waitcnt vmcnt(N)): every instantiation must issue exactly 20 LDS-DMA pieces and 7
stores before its item loop and 15 + 7 per item, in the order the waits assume, and must not touch scratch (a spill is a vector-memory
instruction the counts do not know).  Compiles csrc/attention.hip to ISA and counts (the compiler merges identical stores and
could one day split or fuse others).  usage: python tools/check_attn_isa.py"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(tempfile.mkdtemp(), "attn.s")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"), "-I",
                "/opt/rocm/include", "-x", "hip", "--cuda-device-only", "-S", os.path.join(ROOT, "reed_amd/csrc/attention.hip"),
                "-o", out] + sys.argv[1:], check=True, stderr=subprocess.DEVNULL)
s = open(out).read()
parts = re.split(r"\n(_Z\w+):[^\n]*\n", s)
bad = 0
for i in range(1, len(parts), 2):
    if "attn_fwd256p_kernel" not in parts[i] and "attn_fwd256v_kernel" not in parts[i]:
        continue
    body = parts[i + 1].split("s_endpgm")[0]
    dma, st = len(re.findall(r"buffer_load_dwordx4[^\n]* lds", body)), len(re.findall(r"buffer_store", body))
    waits = re.findall(r"vmcnt\((\d+)\)", body)
    want = (35, 14, ["17", "17", "12", "22", "0"])
    if "attn_fwd256v_kernel" in parts[i]:   # round 6 (V double-buffered): 15 pieces + 8 dropped stores in front of the item loop, 15 + 7 per
        want = (30, 22, ["13", "17", "25", "0", "0"])   # item, the last item's 7 stores behind it
    ok = (dma, st, waits) == want and "scratch_" not in body
    bad += not ok
    print(("ok  " if ok else "BAD ") + parts[i][:60], "dma", dma, "stores", st, "waits", waits)
sys.exit(1 if bad else 0)
