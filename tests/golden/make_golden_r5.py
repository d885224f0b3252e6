#!/usr/bin/env python3
"""Golden vectors for the raw-image (PIL) side of the embedders -- run only in the build container:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_r5.py

What the reference does to a PIL image before an encoder sees it is library code: timm's eval transform (pretrain/tinyvit_embedder.py:51-53,67-69),
transformers' CLIPProcessor (pretrain/clip_embedder.py:25,51-55) and a torchvision Compose (inference.py:74-85).  All three resample with Pillow.  This
script RUNS Pillow's ``Image.resize`` (12.2, in the image) for every resize and transformers' own ``CLIPImageProcessorPil`` (what ``CLIPProcessor`` calls
when torchvision is absent) for the CLIP pipeline; timm and torchvision are not installed, so their published transform code (Resize to the shortest edge /
CenterCrop / ToTensor / Normalize) is restated around Pillow's resize with torch's fp32 ops.  The script also asserts that the Pillow-based restatement of the
CLIP pipeline reproduces the processor's pixel_values bit for bit -- the check that the geometry rules used for the other two are read correctly.
Outputs: tests/golden/preprocess_pil.npz (synthetic input images + expected results, data only: the uint8 crop and the pixel_values in full for two images per
pipeline, a SHA-256 of the uint8 crop and a float64 checksum pair of the pixel_values for the others)."""
import math
import os

import numpy as np
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
IMAGENET = ((0.485, 0.456, 0.406), (0.229, 0.224, 0.225))
CLIP = ((0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711))


def synth(h, w, mode, seed):
    """smooth structure + texture + hard edges, so that both the antialiasing and the bicubic overshoot (clipping at 0 / 255) are exercised"""
    g = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    ch = 4 if mode == "RGBA" else (1 if mode == "L" else 3)
    a = np.zeros((h, w, ch))
    for c in range(ch):
        a[..., c] = 127 + 90 * np.sin(x / (7.0 + 3 * c) + c) * np.cos(y / (11.0 - 2 * c)) + 35 * g.standard_normal((h, w))
        a[(y.astype(int) // 23 + x.astype(int) // 31 + c) % 5 == 0, c] = 255 * ((c + seed) % 2)
    a = np.clip(np.rint(a), 0, 255).astype(np.uint8)
    return Image.fromarray(a[..., 0] if ch == 1 else a, mode)


def tv_resize_size(w, h, s):            # torchvision _compute_resized_output_size, size = int: shortest edge -> s
    return (s, int(s * h / w)) if w <= h else (int(s * w / h), s)


def tv_center_crop(im, c):              # torchvision F.center_crop on a PIL image at least as large as the crop
    w, h = im.size
    top, left = int(round((h - c) / 2.0)), int(round((w - c) / 2.0))
    return im.crop((left, top, left + c, top + c))


def to_tensor_normalize(im, mean, std):  # torchvision ToTensor + Normalize
    t = torch.from_numpy(np.asarray(im, np.uint8).copy()).permute(2, 0, 1).to(torch.float32).div(255)
    m, s = torch.tensor(mean, dtype=torch.float32).view(3, 1, 1), torch.tensor(std, dtype=torch.float32).view(3, 1, 1)
    return t.sub(m).div(s).numpy()


def timm_eval(im, img_size, crop_pct, crop_mode):
    """timm transforms_imagenet_eval, crop_mode center / squash, interpolation bicubic"""
    im = im.convert("RGB")
    scale = math.floor(img_size / crop_pct)
    size = (scale, scale) if crop_mode == "squash" else tv_resize_size(*im.size, scale)
    im = tv_center_crop(im.resize(size, Image.Resampling.BICUBIC), img_size)
    return np.asarray(im, np.uint8).copy(), to_tensor_normalize(im, *IMAGENET)


def inference_transform(im, size, mean, std):
    """inference.py:84: T.Resize(size) (bilinear on PIL images) -> T.CenterCrop(size) -> T.ToTensor() -> T.Normalize(mean, std); the caller converts to RGB (:90)"""
    im = im.convert("RGB")
    im = tv_center_crop(im.resize(tv_resize_size(*im.size, size), Image.Resampling.BILINEAR), size)
    return np.asarray(im, np.uint8).copy(), to_tensor_normalize(im, mean, std)


def clip_restated(im, size=224):
    im = im.convert("RGB")
    w, h = im.size
    ws, hs = tv_resize_size(w, h, size)            # transformers get_resize_output_image_size(shortest_edge): the same rule
    im = im.resize((ws, hs), Image.Resampling.BICUBIC)
    top, left = (hs - size) // 2, (ws - size) // 2
    return np.asarray(im.crop((left, top, left + size, top + size)), np.uint8).copy()


def sha(u8):                              # bit-exact comparison of a large uint8 result without storing it
    import hashlib
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(u8).tobytes()).digest(), np.uint8).copy()


def csum(pv):
    return np.asarray([pv.astype(np.float64).sum(), np.abs(pv.astype(np.float64)).sum()])


def main():
    from transformers import CLIPImageProcessorPil
    proc = CLIPImageProcessorPil()
    imgs = {"land": synth(301, 452, "RGB", 1), "port": synth(381, 233, "RGB", 2), "gray": synth(333, 250, "L", 3), "rgba": synth(257, 300, "RGBA", 4),
            "small": synth(226, 230, "RGB", 5), "sq": synth(224, 224, "RGB", 6), "wide": synth(240, 531, "RGB", 7)}
    full = ("land", "gray")                                    # pixel_values stored in full for these; a float64 checksum pair (sum, sum |.|) for the others
    out = {}
    for name, im in imgs.items():
        out[f"{name}.img"] = np.asarray(im, np.uint8)
        out[f"{name}.mode"] = np.asarray([ord(c) for c in im.mode], np.uint8)
        pv = proc(images=im, return_tensors="np")["pixel_values"][0].astype(np.float32)
        u8 = clip_restated(im)
        m, s = (np.asarray(v, np.float32).reshape(3, 1, 1) for v in CLIP)
        mine = ((u8.astype(np.float32).transpose(2, 0, 1) * np.float32(1 / 255)) - m) / s
        assert np.abs(mine - pv).max() <= 1e-6, (name, float(np.abs(mine - pv).max()))          # the restated geometry IS the processor's
        out[f"{name}.clip_u8" if name in full else f"{name}.clip_sha"] = u8 if name in full else sha(u8)
        out[f"{name}.clip_pv" if name in full else f"{name}.clip_pv_sum"] = pv if name in full else csum(pv)
        u8, pv = timm_eval(im, 224, 0.95, "center")
        out[f"{name}.timm224_u8" if name in full else f"{name}.timm224_sha"] = u8 if name in full else sha(u8)
        out[f"{name}.timm224_pv" if name in full else f"{name}.timm224_pv_sum"] = pv.astype(np.float32) if name in full else csum(pv)
    for name in ("land", "port"):                              # the 512 variant's "squash" mode and the inference transform: uint8 crops only (size)
        u8, _ = timm_eval(imgs[name], 512, 1.0, "squash")
        out[f"{name}.timm512_sha"] = sha(u8)
        u8, pv = inference_transform(imgs[name], 336, *CLIP)
        out[f"{name}.inf336_sha"] = sha(u8)
        out[f"{name}.inf336_pv_sum"] = csum(pv)
    u8, _ = timm_eval(imgs["small"], 384, 1.0, "center")       # the 384 variant: crop_pct 1.0, centre mode (an up-scale)
    out["small.timm384_sha"] = sha(u8)
    np.savez_compressed(os.path.join(HERE, "preprocess_pil.npz"), **out)
    print("wrote preprocess_pil.npz:", sorted(k for k in out if not k.endswith((".img", ".mode"))))


if __name__ == "__main__":
    main()
