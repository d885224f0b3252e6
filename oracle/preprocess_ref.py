"""CPU restatement (TEST INFRASTRUCTURE ONLY) of the batch conditioning the reference's training loop applies before the encoder
(main_coordinator_idun_s3.py:337-381) and of prototype building (models/proto_refiner.py:461-517).

Pinned: tests/golden/preprocess.npz is produced by running the reference's own statements (torch.nn.functional.interpolate +
its normalisation lines) in tests/golden/make_golden.py; this numpy version is checked against it in tests/test_oracle_geo.py."""
import numpy as np


def bilinear_resize(x: np.ndarray, size) -> np.ndarray:
    """F.interpolate(x, size=size, mode="bilinear", align_corners=False) for (..., H, W) float32 (ATen upsample_bilinear2d:
    src = scale*(dst+0.5)-0.5 clamped at 0, idx1 = min(idx0+1, in-1), rows then columns)  -- main_coordinator_idun_s3.py:343-360."""
    x = np.asarray(x, np.float32)
    hs, ws = x.shape[-2:]
    hd, wd = size
    if (hs, ws) == (hd, wd):
        return x.copy()

    def idx(n_in, n_out):
        scale = np.float32(n_in) / np.float32(n_out)
        src = np.maximum(scale * (np.arange(n_out, dtype=np.float32) + np.float32(0.5)) - np.float32(0.5), np.float32(0))
        i0 = np.minimum(src.astype(np.int64), n_in - 1)
        i1 = np.minimum(i0 + 1, n_in - 1)
        l1 = np.clip(src - i0.astype(np.float32), 0, 1).astype(np.float32)
        return i0, i1, (np.float32(1) - l1).astype(np.float32), l1

    y0, y1, ly0, ly1 = idx(hs, hd)
    x0, x1, lx0, lx1 = idx(ws, wd)
    top = x[..., y0, :][..., :, x0] * lx0 + x[..., y0, :][..., :, x1] * lx1
    bot = x[..., y1, :][..., :, x0] * lx0 + x[..., y1, :][..., :, x1] * lx1
    return (top * ly0[:, None] + bot * ly1[:, None]).astype(np.float32)


def prepare_batch(images: np.ndarray, target_dimensions=None, norm_mean=None, norm_std=None) -> np.ndarray:
    """main_coordinator_idun_s3.py:337-381 for float images (B,V,3,H,W) or (B,3,H,W): resize, then (x-mean)/std per channel."""
    x = np.asarray(images)
    u8 = x.dtype == np.uint8
    x = x.astype(np.float32)
    if target_dimensions is not None:
        x = bilinear_resize(x, target_dimensions)
    if u8:
        x = x / np.float32(255.0)
    if norm_mean is not None and norm_std is not None:
        m = np.asarray(norm_mean, np.float32).reshape(3, 1, 1)
        s = np.asarray(norm_std, np.float32).reshape(3, 1, 1)
        x = (x - m) / s
    return x.astype(np.float32)


def cluster_mean(embeddings: np.ndarray, ptr: np.ndarray, member: np.ndarray) -> np.ndarray:
    """models/proto_refiner.py:461-517: running fp32 sum over the members of a cluster in order, divided by the count."""
    emb = np.asarray(embeddings, np.float32)
    out = np.zeros((len(ptr) - 1, emb.shape[1]), np.float32)
    for k in range(len(ptr) - 1):
        s = np.zeros(emb.shape[1], np.float32)
        for m in member[ptr[k]:ptr[k + 1]]:
            s = (s + emb[m]).astype(np.float32)
        if ptr[k + 1] > ptr[k]:
            out[k] = s / np.float32(ptr[k + 1] - ptr[k])
    return out


def prototype_means(table: np.ndarray, latlon: np.ndarray, ptr: np.ndarray, member: np.ndarray) -> np.ndarray:
    """``Embeddings.generate_embeddings`` (models/proto_refiner.py:461-517) for every cluster of a CSR member list: members outside
    [0, len(latlon)) or with a non-finite (lat, lon) row are skipped (:469-474), each member's (V, D) embedding is averaged over
    its views (:483-485), clusters keep a running fp32 sum in list order divided by the number of VALID members (:492-515), a
    cluster without valid members is the zero vector (:500-512).  PINNED by tests/golden/proto_mean.npz."""
    table = np.asarray(table, np.float32)
    views = np.zeros(table.shape[:1] + table.shape[2:], np.float32)
    for v in range(table.shape[1]):                     # torch's mean over dim 0 of a (V, D) tensor: sequential row sum, / V
        views = (views + table[:, v]).astype(np.float32)
    views = (views / np.float32(table.shape[1])).astype(np.float32)
    out = np.zeros((len(ptr) - 1, table.shape[-1]), np.float32)
    for k in range(len(ptr) - 1):
        s, cnt = np.zeros(table.shape[-1], np.float32), 0
        for m in member[ptr[k]:ptr[k + 1]]:
            if m < 0 or m >= len(latlon) or not np.isfinite(latlon[m]).all():
                continue
            s = (s + views[m]).astype(np.float32)
            cnt += 1
        if cnt:
            out[k] = s / np.float32(cnt)
    return out
