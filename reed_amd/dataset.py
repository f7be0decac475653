"""CustomDataset — the reference's on-disk training format (image/dataset.py:18-85, written by
image/preprocessing/dataset_tools.py): <data>/images/**.png (uint8 RGB 256x256), <data>/vae-sd/**.npy (SD-VAE
moments f32 [8,32,32]) + vae-sd/dataset.json {"labels": [[fname, int], ...]}, optional <data>/<text_embeds_dir>/**.npy.
Item: (image u8[3,H,W], moments f32[8,h,w], label i64, text f32[Dt] or zeros_like(moments)).

Additive (not in the reference): `features_dirs=[...]` loads precomputed frozen-encoder patch features
<data>/<dir>/**.npy f32 [256, z] instead of running the encoder every step, and images are
optional when no on-the-fly encoder needs them.

`pack_dataset` / `PackedDataset`: at ~1000 images/s per GPU the reference format costs one PNG
decode and 2-3 small-file opens per image per step. Packing writes the same items, bit for bit and in the same
(sorted-filename) order, into one memory-mappable array per field; `PackedDataset[i]` returns exactly what
`CustomDataset[i]` returns.
"""
import json
import os

import numpy as np
import torch
from torch.utils.data import Dataset


class CustomDataset(Dataset):
    def __init__(self, data_dir, text_embeds_dir=None, features_dirs=None, need_images=True):
        self.images_dir = os.path.join(data_dir, "images")
        self.features_dir = os.path.join(data_dir, "vae-sd")
        self.need_images = need_images and os.path.isdir(self.images_dir)
        exts = {".png", ".jpg", ".jpeg", ".npy", ".bmp", ".webp"}

        def walk(root):
            return sorted(os.path.relpath(os.path.join(r, f), start=root) for r, _d, files in os.walk(root)
                          for f in files if os.path.splitext(f)[1].lower() in exts)

        self.image_fnames = walk(self.images_dir) if self.need_images else []
        self.feature_fnames = walk(self.features_dir)
        with open(os.path.join(self.features_dir, "dataset.json"), "rb") as f:
            labels = dict(json.load(f)["labels"])
        labels = np.array([labels[fn.replace("\\", "/")] for fn in self.feature_fnames])
        self.labels = labels.astype({1: np.int64, 2: np.float32}[labels.ndim])
        self.text_embeds_dir = text_embeds_dir
        if text_embeds_dir is not None:
            self.full_text_embeds_dir = os.path.join(data_dir, text_embeds_dir)
            assert os.path.exists(self.full_text_embeds_dir), f"Text embeds dir {self.full_text_embeds_dir} does not exist"
        self.z_dirs = [os.path.join(data_dir, d) for d in (features_dirs or [])]

    def __len__(self):
        if self.need_images:
            assert len(self.image_fnames) == len(self.feature_fnames), \
                "Number of feature files and label files should be same"
        return len(self.feature_fnames)

    @staticmethod
    def _stem(feature_fname):
        d, f = os.path.split(feature_fname)
        return os.path.join(d, os.path.splitext(f)[0].replace("img-mean-std-", "img"))

    def __getitem__(self, idx):
        ffn = self.feature_fnames[idx]
        features = np.load(os.path.join(self.features_dir, ffn))
        if self.need_images:
            ifn = self.image_fnames[idx]
            ext = os.path.splitext(ifn)[1].lower()
            if ext == ".npy":
                image = np.load(os.path.join(self.images_dir, ifn))
                image = image.reshape(-1, *image.shape[-2:])
            else:
                import PIL.Image
                image = np.array(PIL.Image.open(os.path.join(self.images_dir, ifn)))
                image = image.reshape(*image.shape[:2], -1).transpose(2, 0, 1)
            image = torch.from_numpy(image)
            stem = os.path.splitext(ifn)[0]
        else:
            image = torch.zeros(0, dtype=torch.uint8)
            stem = self._stem(ffn)
        if self.text_embeds_dir is not None:
            text = torch.from_numpy(np.load(os.path.join(self.full_text_embeds_dir, stem + ".npy")))
        else:
            text = torch.zeros_like(torch.from_numpy(features))
        out = (image, torch.from_numpy(features), torch.tensor(self.labels[idx]), text)
        if self.z_dirs:
            out = out + tuple(torch.from_numpy(np.load(os.path.join(d, stem + ".npy"))).float() for d in self.z_dirs)
        return out


PACK_META = "packed.json"


def pack_dataset(data_dir, out_dir, text_embeds_dir=None, features_dirs=None, with_images=True, log_every=0):
    """Write <out_dir>/{moments,labels[,images][,text][,z0,z1,...]}.npy + packed.json from the reference format."""
    ds = CustomDataset(data_dir, text_embeds_dir=text_embeds_dir, features_dirs=features_dirs, need_images=with_images)
    n = len(ds)
    if n == 0:
        raise ValueError(f"{data_dir}: empty dataset")
    os.makedirs(out_dir, exist_ok=True)
    first = ds[0]
    fields = ["images", "moments", "labels", "text"] + [f"z{j}" for j in range(len(first) - 4)]
    keep = {"images": with_images and ds.need_images, "text": text_embeds_dir is not None}
    arrays = {}
    for name, t in zip(fields, first):
        if not keep.get(name, True):
            continue
        arrays[name] = np.lib.format.open_memmap(os.path.join(out_dir, name + ".npy"), mode="w+",
                                                 dtype=t.numpy().dtype, shape=(n,) + tuple(t.shape))
    for i in range(n):
        item = first if i == 0 else ds[i]
        for name, t in zip(fields, item):
            if name in arrays:
                if tuple(t.shape) != arrays[name].shape[1:]:
                    raise ValueError(f"item {i}: field {name} has shape {tuple(t.shape)}, expected {arrays[name].shape[1:]}")
                arrays[name][i] = t.numpy()
        if log_every and (i + 1) % log_every == 0:
            print(f"[pack_dataset] {i + 1}/{n}", flush=True)
    for a in arrays.values():
        a.flush()
    meta = {"n": n, "fields": {k: {"dtype": str(v.dtype), "shape": list(v.shape[1:])} for k, v in arrays.items()},
            "source": os.path.abspath(data_dir), "text_embeds_dir": text_embeds_dir, "features_dirs": features_dirs or []}
    with open(os.path.join(out_dir, PACK_META), "w") as f:
        json.dump(meta, f, indent=1)
    return meta


class PackedDataset(Dataset):
    """Items of `CustomDataset` from memory-mapped arrays (`pack_dataset`): same tuple, same dtypes, same order."""

    def __init__(self, packed_dir):
        with open(os.path.join(packed_dir, PACK_META)) as f:
            self.meta = json.load(f)
        self.n = self.meta["n"]
        self.arr = {k: np.load(os.path.join(packed_dir, k + ".npy"), mmap_mode="r") for k in self.meta["fields"]}
        self.zkeys = sorted((k for k in self.arr if k.startswith("z")), key=lambda k: int(k[1:]))

    def __len__(self):
        return self.n

    def _get(self, k, idx):
        return torch.from_numpy(np.array(self.arr[k][idx]))   # copy out of the mapping

    def __getitem__(self, idx):
        moments = self._get("moments", idx)
        image = self._get("images", idx) if "images" in self.arr else torch.zeros(0, dtype=torch.uint8)
        text = self._get("text", idx) if "text" in self.arr else torch.zeros_like(moments)
        out = (image, moments, torch.tensor(self.arr["labels"][idx]), text)
        return out + tuple(self._get(k, idx).float() for k in self.zkeys)


class SyntheticLatents(Dataset):
    """Random ImageNet-256-shaped items (SURVEY.md §8d synthetic inputs): for plumbing tests and throughput runs.  Latents and
    labels are drawn per index; the encoder features (1 MB per item for a 256 x 1024 target: drawing them per item costs 3 ms of
    host time, i.e. 300 items/s per loader process) come from a pool of `pool` tensors per encoder drawn once per process."""

    def __init__(self, n, z_dims=(), z_types=(), num_classes=1000, seed=0, latent=32, pool=32):
        self.n, self.z_dims, self.z_types, self.nc, self.seed, self.latent = n, list(z_dims), list(z_types), num_classes, seed, latent
        self.pool = max(1, int(pool))
        self._zs = None

    def __len__(self):
        return self.n

    def _pool(self):
        if self._zs is None:
            T = (self.latent // 2) ** 2
            g = torch.Generator().manual_seed(self.seed * 7919 + 17)
            self._zs = [torch.randn(self.pool, T, z, generator=g) if k == "i" else torch.randn(self.pool, z, generator=g)
                        for z, k in zip(self.z_dims, self.z_types)]
        return self._zs

    def __getitem__(self, idx):
        g = torch.Generator().manual_seed(self.seed * 1000003 + idx)
        mean = torch.randn(4, self.latent, self.latent, generator=g) * 5.49
        moments = torch.cat([mean, torch.full_like(mean, 0.5)], 0)
        label = torch.randint(0, self.nc, (), generator=g)
        zs = tuple(z[idx % self.pool] for z in self._pool())
        return (torch.zeros(0, dtype=torch.uint8), moments, label, torch.zeros(0)) + zs


if __name__ == "__main__":   # python -m reed_amd.dataset pack <data_dir> <out_dir> [--text-embeds-dir D] [--features-dirs A B] [--no-images]
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("cmd", choices=["pack"])
    ap.add_argument("data_dir")
    ap.add_argument("out_dir")
    ap.add_argument("--text-embeds-dir", default=None)
    ap.add_argument("--features-dirs", nargs="*", default=None)
    ap.add_argument("--no-images", action="store_true")
    a = ap.parse_args()
    m = pack_dataset(a.data_dir, a.out_dir, a.text_embeds_dir, a.features_dirs, with_images=not a.no_images, log_every=10000)
    print(json.dumps(m))
