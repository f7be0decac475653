#!/usr/bin/env python
"""attn_fwd256p_kernel's and attn_fwd256v_kernel's waits are counted (s_waitcnt vmcnt(N)): every instantiation must issue exactly 20 LDS-DMA pieces and 7
stores before its item loop and 15 + 7 per item, in the order the waits assume, and must not touch scratch (a spill is a vector-memory
instruction the counts do not know).  Compiles csrc/attention.hip to ISA and counts (the compiler merges identical stores and
could one day split or fuse others).  attn_bwd_ring_kernel: the wave's own rows of the next item (V fragments, lse, delta) are loaded by
inline asm into the registers the item loop carries, and waited for by a counted wait much later — a temporary + copy in between would
read a register the load has not written yet: the loads inside the item loop must write exactly the registers the loads in front of
it write, and no scratch.  usage: python tools/check_attn_isa.py"""
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
    if "attn_bwd_ring_kernel" in parts[i]:
        body = parts[i + 1].split("s_endpgm")[0]
        lines = body.split("\n")
        first_bar = next(k for k, l in enumerate(lines) if "s_barrier" in l)
        dst = [(k, l.split()[1].rstrip(",")) for k, l in enumerate(lines) if l.strip().startswith(("global_load_dwordx4", "global_load_dword "))]
        pro, loop = [d for k, d in dst if k < first_bar], [d for k, d in dst if k > first_bar]
        ok = len(pro) > 0 and set(pro) == set(loop) and len(loop) % len(pro) == 0 and "scratch_" not in body
        bad += not ok
        print(("ok  " if ok else "BAD ") + parts[i][:60], "own-row loads in front of the loop", len(pro), "inside", len(loop), "same registers", set(pro) == set(loop))
        continue
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
