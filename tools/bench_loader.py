#!/usr/bin/env python
"""Data path: items/s of one loader process on the reference's on-disk format (PNG decode + per-item
.npy opens, image/dataset.py:18-85) and on the packed memory-mapped form of the same items (reed_amd/dataset.py), with
real item sizes (256x256x3 PNG, f32 [8,32,32] moments, f32 [256,1024] precomputed features).
usage: python tools/bench_loader.py [n_items]"""
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reed_amd.dataset import CustomDataset, PackedDataset, pack_dataset  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
root = tempfile.mkdtemp(prefix="loaderbench_")
import PIL.Image  # noqa: E402
rng = np.random.default_rng(0)
labels = []
for i in range(n):
    sub = f"{i // 1000:05d}"
    for d in ("images", "vae-sd", "feat"):
        os.makedirs(os.path.join(root, d, sub), exist_ok=True)
    # smooth-ish image (random low-res upsampled) so the PNG is neither trivially nor maximally compressible
    img = np.kron(rng.integers(0, 256, (32, 32, 3), dtype=np.uint8), np.ones((8, 8, 1), dtype=np.uint8))
    PIL.Image.fromarray(img).save(os.path.join(root, "images", sub, f"img{i:08d}.png"), compress_level=0)
    np.save(os.path.join(root, "vae-sd", sub, f"img-mean-std-{i:08d}.npy"), rng.standard_normal((8, 32, 32)).astype(np.float32))
    np.save(os.path.join(root, "feat", sub, f"img{i:08d}.npy"), rng.standard_normal((256, 1024)).astype(np.float32))
    labels.append([f"{sub}/img-mean-std-{i:08d}.npy", int(i % 1000)])
json.dump({"labels": labels}, open(os.path.join(root, "vae-sd", "dataset.json"), "w"))


def rate(ds, order):
    t0 = time.perf_counter()
    s = 0
    for i in order:
        it = ds[int(i)]
        s += int(it[1].numel())
    return len(order) / (time.perf_counter() - t0)


order = rng.permutation(n)
res = {"n_items": n}
for tag, kw in (("features", dict(features_dirs=["feat"], need_images=False)), ("images", dict(need_images=True))):
    ds = CustomDataset(root, **kw)
    pk_dir = os.path.join(root, "packed_" + tag)
    t0 = time.perf_counter()
    pack_dataset(root, pk_dir, features_dirs=kw.get("features_dirs"), with_images=kw["need_images"])
    t_pack = time.perf_counter() - t0
    pk = PackedDataset(pk_dir)
    rate(ds, order[:16]); rate(pk, order[:16])
    res[tag] = {"files_items_per_s": round(rate(ds, order), 1), "packed_items_per_s": round(rate(pk, order), 1),
                "pack_seconds": round(t_pack, 2)}
print(json.dumps(res))
