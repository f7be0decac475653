"""SD-VAE decode rate: the HIP path of reed_amd/vae.py per operand type against the same module on torch's
operators (MIOpen convolutions), published sd-vae-ft configuration, 32x32 latents -> 256x256 images, random weights."""
import argparse
import sys
import time

import torch

sys.path.insert(0, ".")
from reed_amd import ops, vae as rvae  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
dec = rvae.SDVAEDecoder()
for p in dec.parameters():
    p.data.normal_(0, 0.02)
dec = dec.to(dev)
z = torch.randn(a.batch, 4, 32, 32, device=dev)


def timed(fn):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / a.reps


flop = 622.3e9      # contraction flop per 256 x 256 image of this configuration (padded GEMM shapes, counted once with the GEMM probe)
for prec in ("fp32", "fp16", "bf16"):
    t = timed(lambda: dec.decode(z, precision=prec))
    print(f"HIP {prec}: {t * 1e3 / a.batch:8.2f} ms / image, {a.batch / t:8.1f} images/s, {flop * a.batch / t / 1e12:7.1f} TFLOP/s")
with torch.no_grad():
    t = timed(lambda: dec.decode_torch(z))
print(f"torch operators (MIOpen), fp32: {t * 1e3 / a.batch:8.2f} ms / image, {a.batch / t:8.1f} images/s")
