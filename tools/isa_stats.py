"""Per-kernel ISA statistics of a hipcc -S dump: scratch traffic, vmcnt(0) drains, transposing reads, copies.
Usage: python tools/isa_stats.py file.s [name-filter]"""
import re
import sys

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
parts = re.split(r"\n(_Z\w+):[^\n]*\n", s)
for i in range(1, len(parts), 2):
    name, body = parts[i], parts[i + 1].split("s_endpgm")[0]
    if flt not in name:
        continue
    loops = re.findall(r"s_cbranch_\w+ (\.LBB\d+_\d+)", body)
    c = lambda pat: len(re.findall(pat, body))
    n_scr, n_vm0 = c(r"scratch_(load|store)"), c(r"vmcnt\(0\)")
    print(f"{name[:70]:70s} scratch={n_scr:3d} vmcnt0={n_vm0:3d} "
          f"tr={c('ds_read_b64_tr_b16'):4d} b128={c('ds_read_b128'):4d} mfma={c('v_mfma'):4d} "
          f"v_mov={c('v_mov_b32'):4d} lines={body.count(chr(10))}")
