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


# ------------------------------------------------------------------------------------------------- raw-image pipelines (PIL resampling)
# What the reference's raw-image entry points do to a PIL image before the encoder sees it:
#   pretrain/tinyvit_embedder.py:51-53,67-69  timm.data.create_transform(**resolve_model_data_config(model), is_training=False):
#       Resize(floor(img_size / crop_pct), bicubic) [crop_mode "squash": Resize((img_size, img_size))] -> CenterCrop(img_size) -> ToTensor -> Normalize
#   pretrain/clip_embedder.py:25,51-55        CLIPProcessor: convert RGB -> resize shortest edge 224 (PIL bicubic) -> center crop 224 -> * 1/255 -> normalize
#   inference.py:74-85                        T.Resize(size) (PIL bilinear, which Pillow applies with its support scaled = antialiased) -> CenterCrop -> ToTensor -> Normalize
# All three resample with Pillow's ImagingResample on 8-bit pixels (Pillow 12.2, src/libImaging/Resample.c; timm and torchvision are not in the image: their
# published transform code is restated here).  The resampler is restated below integer for integer: double-precision filter weights normalised per output
# pixel, converted to 22-bit fixed point, a horizontal pass into an 8-BIT intermediate image (rounded, clipped), then a vertical pass.
# PINNED by tests/golden/preprocess_pil.npz, which tests/golden/make_golden_r5.py produces by running Pillow's Image.resize and transformers'
# CLIPImageProcessorPil themselves (tests/test_oracle_geo.py::test_pil_pipelines_match_pillow_and_transformers).
PIL_BILINEAR, PIL_BICUBIC = 2, 3            # PIL.Image.Resampling codes
_PRECISION_BITS = 32 - 8 - 2


def _pil_filter(x: float, flt: int) -> float:
    if x < 0.0:
        x = -x
    if flt == PIL_BILINEAR:
        return 1.0 - x if x < 1.0 else 0.0
    a = -0.5
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_coeffs(in_size: int, out_size: int, flt: int):
    """precompute_coeffs + normalize_coeffs_8bpc of Resample.c for a full-image resize: (bounds (out, 2) int, kk (out, ksize) int32)."""
    support0 = 1.0 if flt == PIL_BILINEAR else 2.0
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = support0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int64)
    kk = np.zeros((out_size, ksize), np.int64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_pil_filter((x + xmin - center + 0.5) * ss, flt) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << _PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << _PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pil_pass(img: np.ndarray, bounds: np.ndarray, kk: np.ndarray) -> np.ndarray:
    """One 8-bit resampling pass along axis 1 of an (R, C, 3) uint8 image: ss = 2^21 + sum pixel * k, clip8(ss >> 22)."""
    out = np.empty((img.shape[0], len(bounds), img.shape[2]), np.uint8)
    src = img.astype(np.int64)
    for xx, (xmin, xmax) in enumerate(bounds):
        ss = (1 << (_PRECISION_BITS - 1)) + (src[:, xmin:xmin + xmax, :] * kk[xx, :xmax, None]).sum(1)
        out[:, xx, :] = np.clip(ss >> _PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def pil_resize(img_hwc: np.ndarray, out_w: int, out_h: int, flt: int) -> np.ndarray:
    """PIL.Image.resize((out_w, out_h), resample=flt) of an (H, W, 3) uint8 RGB image (ImagingResample: horizontal pass first, each pass only when that
    axis changes size)."""
    img = np.ascontiguousarray(img_hwc, np.uint8)
    h, w = img.shape[:2]
    if out_w != w:
        bx, kx = pil_coeffs(w, out_w, flt)
        img = _pil_pass(img, bx, kx)
    if out_h != h:
        by, ky = pil_coeffs(h, out_h, flt)
        img = _pil_pass(img.transpose(1, 0, 2), by, ky).transpose(1, 0, 2)
    return np.ascontiguousarray(img)


def raw_image_geometry(h: int, w: int, pipeline: str, size: int, crop_pct: float = 1.0, crop_mode: str = "center"):
    """(filter, resized (H, W), crop (top, left), crop size) of the three pipelines.  Shortest edge -> s, the long edge int(s * long / short) (torchvision
    ``_compute_resized_output_size`` and transformers ``get_resize_output_image_size`` agree); centre-crop offsets: torchvision int(round((H - c) / 2.0))
    (Python's round-half-even), transformers (H - c) // 2."""
    if pipeline == "clip":
        flt, s, squash = PIL_BICUBIC, size, False
    elif pipeline == "timm":
        flt, s, squash = PIL_BICUBIC, int(np.floor(size / crop_pct)), crop_mode == "squash"
    elif pipeline == "torchvision":
        flt, s, squash = PIL_BILINEAR, size, False
    else:
        raise ValueError(pipeline)
    if squash:
        hr, wr = s, s
    elif w <= h:
        wr, hr = s, int(s * h / w)
    else:
        hr, wr = s, int(s * w / h)
    if pipeline == "clip":
        top, left = (hr - size) // 2, (wr - size) // 2
    else:
        top, left = int(round((hr - size) / 2.0)), int(round((wr - size) / 2.0))
    return flt, (hr, wr), (top, left), size


def raw_image_pixel_values(img_hwc: np.ndarray, pipeline: str, size: int, mean, std, crop_pct: float = 1.0, crop_mode: str = "center"):
    """(uint8 (size, size, 3) crop, float32 (3, size, size) pixel_values) of one RGB image.  Float side: torchvision ToTensor divides the uint8 by 255 in fp32
    and Normalize computes (x - mean) / std in fp32; transformers rescales by 1/255 and normalises in fp32 as well (image * scale, then (image - mean) / std)."""
    h, w = img_hwc.shape[:2]
    flt, (hr, wr), (top, left), c = raw_image_geometry(h, w, pipeline, size, crop_pct, crop_mode)
    if hr < c or wr < c:
        raise ValueError("raw_image_pixel_values: the resized image is smaller than the crop (the upstream transforms would pad): not restated")
    r = pil_resize(img_hwc, wr, hr, flt)
    u8 = r[top:top + c, left:left + c]
    x = u8.astype(np.float32).transpose(2, 0, 1)
    m = np.asarray(mean, np.float32).reshape(3, 1, 1)
    s = np.asarray(std, np.float32).reshape(3, 1, 1)
    x = x * np.float32(1 / 255) if pipeline == "clip" else x / np.float32(255)
    return u8, ((x - m) / s).astype(np.float32)


def to_rgb(arr: np.ndarray, mode: str) -> np.ndarray:
    """PIL ``Image.convert("RGB")`` for the modes the fixture holds: "L" replicates the channel, "RGBA" drops alpha (no premultiplication), "RGB" is itself."""
    a = np.asarray(arr, np.uint8)
    if mode == "L":
        return np.repeat(a[..., None], 3, axis=2)
    if mode == "RGBA":
        return np.ascontiguousarray(a[..., :3])
    if mode == "RGB":
        return a
    raise ValueError(mode)
