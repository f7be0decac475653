"""ctypes binding of libreed_hip.so (C ABI in include/reed_hip.h).

The product path has no fallback: if the shared object is missing or a call fails, this raises.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# REED_HIP_LIB=<path of another build's libreed_hip.so>: same-box A/B of two builds of the library (tools/ab_lib.sh); its
# siblings <stem>_f16.so / <stem>_f32.so are taken where they exist.  The product loads the in-tree libraries.
LIB_PATH = os.environ.get("REED_HIP_LIB") or os.path.join(_HERE, "libreed_hip.so")


def _sibling(suffix):
    alt = LIB_PATH[:-3] + suffix + ".so" if LIB_PATH.endswith(".so") else ""
    if alt and os.path.exists(alt):
        return alt
    if os.environ.get("REED_HIP_LIB"):   # an A/B build without this sibling: say so once — the two precisions then come from two builds
        import sys
        print(f"[reed_amd] REED_HIP_LIB={LIB_PATH} has no {os.path.basename(alt)} beside it: the in-tree libreed_hip{suffix}.so is used "
              f"for that operand type", file=sys.stderr)
    return os.path.join(_HERE, "libreed_hip" + suffix + ".so")


# the same sources built with IEEE-half operands (csrc/common.hpp, -DREED_FP16): the sampling path
LIB_PATH_F16 = _sibling("_f16")
# the fp32-operand build (-DREED_FP32): --mixed-precision no / generate.py --no-tf32
LIB_PATH_F32 = _sibling("_f32")
PRECISIONS = ("bf16", "fp16", "fp32")
HEADER_PATH = os.path.join(_HERE, "..", "include", "reed_hip.h")

_lib = None
_libs = {}

_CT = {
    "int": ctypes.c_int, "int64_t": ctypes.c_int64, "float": ctypes.c_float,
    "double": ctypes.c_double,
}


def parse_header(path=HEADER_PATH):
    """Return {name: (restype, [(ctype, argname), ...])} for every prototype in reed_hip.h."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    protos = {}
    for m in re.finditer(r"\b(int64_t|int|const char\*)\s+(reed_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        alist = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a:
                    alist.append((ctypes.c_void_p, a.split("*")[-1].strip()))
                else:
                    toks = a.replace("const ", "").split()
                    alist.append((_CT[toks[0]], toks[-1]))
        protos[name] = (ctypes.c_char_p if "char" in ret else (ctypes.c_int64 if ret == "int64_t" else ctypes.c_int), alist)
    return protos


def load(precision="bf16"):
    """precision "bf16": libreed_hip.so (training and everything else); "fp16": libreed_hip_f16.so (sampling at TF32's
    mantissa, --mixed-precision fp16); "fp32": libreed_hip_f32.so (--mixed-precision no, generate.py --no-tf32)."""
    global _lib
    if precision in _libs:
        return _libs[precision]
    path = {"bf16": LIB_PATH, "fp16": LIB_PATH_F16, "fp32": LIB_PATH_F32}[precision]
    if not os.path.exists(path):
        raise RuntimeError(
            f"reed_amd: {path} not found. Build it with `python -m reed_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU or PyTorch fallback for the hot path.")
    # PyTorch-ROCm wheels bundle their own libamdhip64 / librccl. Import torch FIRST so that our NEEDED entries
    # resolve (by SONAME) to the runtime torch already loaded: one HIP runtime and one RCCL per process. Loading
    # /opt/rocm's copies first makes torch fail later with "no ROCm-capable device is detected".
    import torch  # noqa: F401
    lib = ctypes.CDLL(path)
    missing = []
    for name, (restype, args) in parse_header().items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype = restype
        fn.argtypes = [t for t, _ in args]
    lib._reed_missing = missing  # tests assert this is empty; calling a missing symbol raises
    if not missing and lib.reed_half_kind() != {"bf16": 0, "fp16": 1, "fp32": 2}[precision]:
        raise RuntimeError(f"reed_amd: {path} was not built for {precision} operands")
    _libs[precision] = lib
    if precision == "bf16":
        _lib = lib
    return lib


def loaded():
    """The builds this process has loaded so far (knobs that every build carries are set in each of them)."""
    return dict(_libs)


def check(rc, what="", lib=None):
    if rc != 0:
        msg = (lib or load()).reed_last_error()
        raise RuntimeError(f"reed_hip {what} failed (code {rc}): {msg.decode() if msg else ''}")
