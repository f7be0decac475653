"""Build libreed_hip.so (gfx950) in-tree with hipcc. No torch extension machinery: the library
is a plain C-ABI shared object (include/reed_hip.h) loaded with ctypes."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libreed_hip.so")
# the same sources with IEEE-half operands (-DREED_FP16, csrc/common.hpp): the sampling path at TF32's mantissa
LIB_F16 = os.path.join(HERE, "libreed_hip_f16.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
CFLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wno-unused-result",
          "-I", os.path.join(HERE, "..", "include"), "-I", "/opt/rocm/include"]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp"))]
    hdrs.append(os.path.join(HERE, "..", "include", "reed_hip.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(job):
    src, f16 = job
    obj = os.path.join(OBJ, src + (".f16.o" if f16 else ".o"))
    srcp = os.path.join(CSRC, src)
    if os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(srcp), _deps_mtime()):
        return obj, False
    cmd = [HIPCC] + CFLAGS + (["-DREED_FP16"] if f16 else []) + ["-x", "hip", "-c", srcp, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    return obj, True


def build(verbose=True, jobs=None):
    os.makedirs(OBJ, exist_ok=True)
    srcs = _sources()
    jobs = jobs or 8
    with ThreadPoolExecutor(jobs) as ex:
        res = list(ex.map(_compile, [(s, f) for f in (False, True) for s in srcs]))
    n = len(srcs)
    for lib, part in ((LIB, res[:n]), (LIB_F16, res[n:])):
        objs = [o for o, _ in part]
        rebuilt = any(c for _, c in part)
        if rebuilt or not os.path.exists(lib):
            cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib] + objs + \
                  ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-Bsymbolic"]   # both libraries export
            r = subprocess.run(cmd, capture_output=True, text=True)                            # the same names
            if r.returncode != 0:
                raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[reed_amd.build] {lib} ({'rebuilt' if rebuilt else 'up to date'}; {n} sources)")
    return LIB


if __name__ == "__main__":
    build()
    sys.exit(0)
