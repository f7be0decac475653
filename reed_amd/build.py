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
# the fp32-operand build (-DREED_FP32: `bf16` is float, v_mfma_f32_32x32x2_f32): --mixed-precision no / generate.py --no-tf32
LIB_F32 = os.path.join(HERE, "libreed_hip_f32.so")
# (variant tag, extra flags, library): every variant compiles the shared sources; the MFMA-tuned 16-bit kernels are left out of
# the fp32 build, whose own GEMM / attention sources are left out of the 16-bit builds
VARIANTS = (("", [], LIB), ("f16", ["-DREED_FP16"], LIB_F16), ("f32", ["-DREED_FP32"], LIB_F32))
ONLY_16BIT = {"gemm.hip", "gemm256.hip", "gemm256w.hip", "gemm144.hip", "gemm288.hip", "gemm_skinny.hip", "gemm_tn.hip", "attention.hip", "conv.hip"}
ONLY_F32 = {"gemm_f32.hip", "attention_f32.hip"}
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
CFLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wno-unused-result",
          "-I", os.path.join(HERE, "..", "include"), "-I", "/opt/rocm/include"]


def _sources(tag=""):
    skip = ONLY_16BIT if tag == "f32" else ONLY_F32
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")) and f not in skip)


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp"))]
    hdrs.append(os.path.join(HERE, "..", "include", "reed_hip.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(job):
    src, tag, flags = job
    obj = os.path.join(OBJ, src + (f".{tag}.o" if tag else ".o"))
    srcp = os.path.join(CSRC, src)
    if os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(srcp), _deps_mtime()):
        return obj, False
    cmd = [HIPCC] + CFLAGS + flags + ["-x", "hip", "-c", srcp, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    return obj, True


def build(verbose=True, jobs=None):
    os.makedirs(OBJ, exist_ok=True)
    jobs = jobs or 8
    work = [(s_, tag, flags) for tag, flags, _ in VARIANTS for s_ in _sources(tag)]
    with ThreadPoolExecutor(jobs) as ex:
        res = dict(zip([(s_, tag) for s_, tag, _ in work], ex.map(_compile, work)))
    for tag, _, lib in VARIANTS:
        part = [res[(s_, tag)] for s_ in _sources(tag)]
        objs = [o for o, _ in part]
        rebuilt = any(c for _, c in part)
        if rebuilt or not os.path.exists(lib):
            cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib] + objs + \
                  ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-Bsymbolic"]   # the libraries export
            r = subprocess.run(cmd, capture_output=True, text=True)                            # the same names
            if r.returncode != 0:
                raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[reed_amd.build] {lib} ({'rebuilt' if rebuilt else 'up to date'}; {len(objs)} sources)")
    return LIB


if __name__ == "__main__":
    build()
    sys.exit(0)
