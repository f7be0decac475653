#!/bin/bash
# per-kernel time of the fp16-operand VAE decode (batch 8, 32x32 latents)
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
cat > /tmp/vae_run.py <<'PY'
import sys, torch
sys.path.insert(0, ".")
from reed_amd import vae as rvae
dev = torch.device("cuda:0")
torch.manual_seed(0)
dec = rvae.SDVAEDecoder()
for p in dec.parameters(): p.data.normal_(0, 0.02)
dec = dec.to(dev)
z = torch.randn(8, 4, 32, 32, device=dev)
prec = sys.argv[1]
for _ in range(3): dec.decode(z, precision=prec)
torch.cuda.synchronize()
PY
rm -rf gpurun_out/vaeprof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/vaeprof -o vae -- python3 /tmp/vae_run.py fp16 > gpurun_out/vaeprof.log 2>&1
find gpurun_out/vaeprof -name "*kernel_stats.csv" -exec cp {} gpurun_out/vae_fp16_kernel_stats.csv \;
