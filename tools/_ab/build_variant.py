#!/usr/bin/env python
"""Build a variant of libreed_hip.so with extra compiler flags into tools/_ab/ (same-box A/B through REED_HIP_LIB).
usage: python tools/_ab/build_variant.py NAME -DFOO=1 ...   ->  tools/_ab/libreed_NAME.so"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from reed_amd import build as B  # noqa: E402

name, flags = sys.argv[1], sys.argv[2:]
here = os.path.dirname(os.path.abspath(__file__))
B.OBJ = os.path.join(here, "_obj_" + name)
B.LIB = os.path.join(here, f"libreed_{name}.so")
B.LIB_F16 = os.path.join(here, f"libreed_{name}_f16.so")
B.CFLAGS = B.CFLAGS + flags
os.makedirs(B.OBJ, exist_ok=True)
srcs = B._sources()
import subprocess
from concurrent.futures import ThreadPoolExecutor
with ThreadPoolExecutor(8) as ex:
    res = list(ex.map(B._compile, [(s, "", []) for s in srcs]))
cmd = [B.HIPCC, "-shared", "-fPIC", f"--offload-arch={B.ARCH}", "-o", B.LIB] + [o for o, _ in res] + \
      ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-Bsymbolic"]
subprocess.run(cmd, check=True)
print(B.LIB)
