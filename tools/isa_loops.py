"""For each kernel of a hipcc -S dump: where do scratch (spill) accesses sit relative to the MFMA code, how many
vector-memory instructions does it issue, which vmcnt immediates does it use. usage: isa_loops.py file.s [filter]"""
import re
import sys

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
parts = re.split(r"\n(_Z\w+):[^\n]*\n", s)
for i in range(1, len(parts), 2):
    name, body = parts[i], parts[i + 1].split("s_endpgm")[0]
    if flt not in name:
        continue
    lines = body.split("\n")
    mf = [k for k, l in enumerate(lines) if "v_mfma" in l]
    sc = [k for k, l in enumerate(lines) if "scratch_" in l]
    inside = [k for k in sc if mf and mf[0] < k < mf[-1]]
    # scratch ops whose neighbours (within 40 lines both sides) are MFMAs = in the K loop proper
    hot = [k for k in inside if any("v_mfma" in l for l in lines[max(0, k - 40):k]) and any("v_mfma" in l for l in lines[k:k + 40])]
    st = len(re.findall(r"buffer_store|global_store", body))
    ld = len(re.findall(r"buffer_load_dword(x\d)? v|global_load", body))
    dma = len(re.findall(r"buffer_load_dwordx4 v\d+, s\[\d+:\d+\], s\d+ offen lds|offen lds", body))
    vm = sorted(set(re.findall(r"vmcnt\((\d+)\)", body)), key=int)
    m = re.search(r"ILi(\d)ELi(\d)E", name)
    print(f"{m.groups() if m else name[:40]} mfma={len(mf)} scratch total={len(sc)} between-mfma={len(inside)} hot={len(hot)} "
          f"stores={st} loads={ld} dma={dma} vmcnt={','.join(vm)}")
