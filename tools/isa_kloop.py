#!/usr/bin/env python
"""Instruction mix and order of the MFMA loops of one kernel in a `hipcc -S --cuda-device-only` dump: per loop (label .. backward
branch) the counts of MFMAs, LDS reads, LDS-DMAs, waits, barriers, other vector / scalar instructions, and the sequence of those
events (M = MFMA, r = transposing LDS read, R = other LDS read, D = LDS-DMA, w(..) = s_waitcnt, |B| = s_barrier).
usage: python tools/isa_kloop.py file.s kernel_name_substring"""
import re
import sys

s = open(sys.argv[1]).read()
m = re.search(r"\n(_Z[^\n:]*" + re.escape(sys.argv[2]) + r"[^\n:]*):[^\n]*\n", s)
body = s[m.end():]
body = body[:body.index(".Lfunc_end")]
lines = body.split("\n")
labels = {l.split(":")[0]: k for k, l in enumerate(lines) if re.match(r"^\.LBB\d+_\d+:", l)}
P_VALU = re.compile(r"^\s+v_(?!mfma)")
P_SALU = re.compile(r"^\s+s_(?!waitcnt|barrier|cbranch|nop)")
for k, l in enumerate(lines):
    mm = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
    if not (mm and mm.group(1) in labels and labels[mm.group(1)] < k):
        continue
    a = labels[mm.group(1)]
    seg = lines[a:k + 1]
    nm = sum("v_mfma" in x for x in seg)
    if nm < 16:
        continue
    cnt = lambda p: sum(bool(re.search(p, x)) for x in seg)  # noqa: E731
    print("loop %s: %d lines, mfma %d, ds_read_tr %d, ds_read other %d, dma %d, s_waitcnt %d, s_barrier %d, valu %d, salu %d, s_nop %d" % (
        mm.group(1), k - a, nm, cnt("ds_read_b64_tr"), cnt(r"ds_read_(?!b64_tr)"), cnt(r"buffer_load.* lds"), cnt("s_waitcnt"),
        cnt("s_barrier"), sum(bool(P_VALU.search(x)) for x in seg), sum(bool(P_SALU.search(x)) for x in seg), cnt("s_nop")))
    seq = []
    for x in seg:
        if "v_mfma" in x: seq.append("M")
        elif "ds_read_b64_tr" in x: seq.append("r")
        elif "ds_read" in x: seq.append("R")
        elif " lds" in x and "buffer_load" in x: seq.append("D")
        elif "s_waitcnt" in x: seq.append("w(" + x.split("s_waitcnt")[1].strip() + ")")
        elif "s_barrier" in x: seq.append("|B|")
    print("".join(seq))
