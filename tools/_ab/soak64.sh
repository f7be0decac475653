cd /root/repo
for p in bf16 fp16; do
  rm -rf /tmp/soak_$p
  timeout -k 10 500 python -m reed_amd.train --exp-name soak --model SiT-XL/2 --output-dir /tmp/soak_$p --mixed-precision $p --batch-size 64 \
     --synthetic 4096 --num-workers 8 --diffusion-warm-up-steps 0 --report-to none --checkpointing-steps 100000 --enc-type dinov2-vit-l \
     --max-train-steps 500 --learning-rate 1e-4 --log-every 50 > /tmp/soak_$p.log 2>&1 || { tail -5 /tmp/soak_$p.log; exit 1; }
  python - <<PY
import json,glob,math
f=glob.glob("/tmp/soak_$p/*/metrics.jsonl")[0]
L=[json.loads(l) for l in open(f)]
print("$p", len(L), "records; loss", [round(r["training_denoising_loss"],4) for r in L], "img/s", [round(r["images_per_sec"]) for r in L][-3:], "all finite", all(math.isfinite(r["training_denoising_loss"]) and math.isfinite(r["grad_norm"]) for r in L))
PY
done
