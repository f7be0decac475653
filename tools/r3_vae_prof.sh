#!/bin/bash
# SD-VAE decode (N4): rate per operand type + per-kernel time of the fp16-operand decode (batch 8, 32x32 latents) -> profiles/r3_vae_decode.txt
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cat > /tmp/vae_run.py <<'PY'
import sys, torch
sys.path.insert(0, sys.argv[2])
from reed_amd import vae as rvae
dev = torch.device("cuda:0")
torch.manual_seed(0)
dec = rvae.SDVAEDecoder()
for p in dec.parameters(): p.data.normal_(0, 0.02)
dec = dec.to(dev)
z = torch.randn(8, 4, 32, 32, device=dev)
for _ in range(3): dec.decode(z, precision=sys.argv[1])
torch.cuda.synchronize()
PY
cd $R && timeout -k 10 300 python tools/time_vae.py > $O/time_vae.log 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf $O/vaeprof
rocprofv3 --kernel-trace --stats --output-format csv -d $O/vaeprof -o vae -- python3 /tmp/vae_run.py fp16 $R > $O/vaeprof.log 2>&1
cd $R
find $O/vaeprof -name "*kernel_stats.csv" -exec cp {} $O/vae_fp16_kernel_stats.csv \;
rm -rf $O/vaeprof
tail -6 $O/time_vae.log
