#!/usr/bin/env python
"""Scalar-memory loads inside MFMA loops: an `s_load` there is a kernel argument (or other constant) the compiler reloads every
iteration — typically because an asm statement with a "memory" clobber sits in the loop — followed by an `s_waitcnt lgkmcnt(0)` on
the wave's critical path.  usage: python tools/isa_sload_scan.py file.s"""
import re
import sys

s = open(sys.argv[1]).read()
for m in re.finditer(r"\n(_Z[^\n:]*):[^\n]*\n", s):
    name = m.group(1)
    body = s[m.end():]
    end = body.find(".Lfunc_end")
    if end < 0:
        continue
    lines = body[:end].split("\n")
    labels = {l.split(":")[0]: k for k, l in enumerate(lines) if re.match(r"^\.LBB\d+_\d+:", l)}
    for k, l in enumerate(lines):
        mm = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < k:
            seg = lines[labels[mm.group(1)]:k + 1]
            nm = sum("v_mfma" in x for x in seg)
            nl = sum(bool(re.search(r"\ts_load_|\ts_buffer_load", x)) for x in seg)
            if nm >= 8 and nl and len(seg) < 3000:
                print(f"{name[:70]:70s} loop {mm.group(1):10s} {len(seg):5d} lines, {nm:3d} MFMAs, {nl} scalar loads")
