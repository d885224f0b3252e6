#!/usr/bin/env python3
"""Golden vectors for the batch-conditioning kernel (gg_preprocess_bilinear) and prototype building (gg_segment_mean),
produced by EXECUTING the reference's own statements -- run only in the build container:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_preprocess.py

The conditioning code is inline in the reference's training loop (main_coordinator_idun_s3.py, from
`if target_dimensions is not None:` to `images = (images - mean_t) / std_t`), not a function: the block is read from the
reference file at run time, dedented and exec'd on our inputs (nothing of it is copied into this repository).  Prototype
building runs ProtoRefiner-side semantics (models/proto_refiner.py:461-517: running fp32 sum in member order / count) through
torch exactly as that loop does.  Outputs: tests/golden/preprocess.npz (inputs + expected outputs)."""
import os
import textwrap

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/main_coordinator_idun_s3.py"


def reference_block():
    lines = open(REF).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.strip() == "if target_dimensions is not None:")
    end = next(i for i, l in enumerate(lines) if i > start and l.strip() == "images = (images - mean_t) / std_t")
    return textwrap.dedent("\n".join(lines[start:end + 1]))


def run_reference(images, target_dimensions, norm_mean, norm_std):
    env = dict(images=images.clone(), target_dimensions=target_dimensions, norm_mean=norm_mean, norm_std=norm_std, device="cpu",
               F=F, torch=torch)
    exec(compile(reference_block(), REF, "exec"), env)
    return env["images"]


def main():
    g = torch.Generator().manual_seed(7)
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)          # main_coordinator_idun_s3.py:214-215 (TinyViT data config)
    cases = {}
    pano = torch.rand(2, 4, 3, 37, 53, generator=g)                    # panorama batch, down-scale to 32x32
    cases["pano_down"] = (pano, (32, 32), mean, std)
    single = torch.rand(3, 3, 20, 24, generator=g)                     # (B,C,H,W), up-scale, anisotropic
    cases["single_up"] = (single, (45, 31), mean, std)
    same = torch.rand(2, 3, 16, 16, generator=g)                       # no resize requested, normalise only
    cases["same_size"] = (same, None, mean, std)
    nonorm = torch.rand(1, 4, 3, 9, 11, generator=g)                   # resize only
    cases["resize_only"] = (nonorm, (14, 14), None, None)
    out = {}
    for name, (x, size, m, s) in cases.items():
        y = run_reference(x, size, m, s)
        out[name + ".x"] = x.numpy(); out[name + ".y"] = y.numpy()
        out[name + ".size"] = np.asarray(size if size is not None else (-1, -1)); out[name + ".norm"] = np.asarray([m is not None])
    # prototype building: running mean over member panoramas (each first averaged over its views), in member order
    emb = torch.randn(23, 4, 64, generator=g)
    pano_vec = torch.stack([e.mean(dim=0) for e in emb])               # `vec.mean(dim=0)` of a (V, D) embedding
    clusters = [[3, 5, 7], [], [0], [22, 1, 2, 4, 6, 8, 9, 10, 11, 12, 13], [14, 15]]
    protos = []
    for idxs in clusters:
        sum_cpu, count = None, 0
        for i in idxs:
            v = pano_vec[i].detach().to("cpu")
            sum_cpu = v.clone() if sum_cpu is None else sum_cpu.add_(v)
            count += 1
        protos.append((sum_cpu / count).contiguous() if count else torch.zeros(64))
    out["proto.emb"] = emb.numpy(); out["proto.pano_vec"] = pano_vec.numpy()
    out["proto.ptr"] = np.cumsum([0] + [len(c) for c in clusters]).astype(np.int64)
    out["proto.member"] = np.asarray([i for c in clusters for i in c], np.int64)
    out["proto.out"] = torch.stack(protos).numpy()
    np.savez_compressed(os.path.join(HERE, "preprocess.npz"), **out)
    print("wrote preprocess.npz", os.path.getsize(os.path.join(HERE, "preprocess.npz")) / 1024, "KB")


if __name__ == "__main__":
    main()
