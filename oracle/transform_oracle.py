"""TEST INFRASTRUCTURE ONLY (imported by tests/ and nothing else).

CPU restatement of the reference's eval image transform:
  mini-imagenet  (test_phase/datasets/mini_imagenet.py:47-56):   Resize((88, 88)) -> CenterCrop(80) -> ToTensor -> Normalize
  tiered-imagenet (test_phase/datasets/tiered_imagenet.py:21,53-57): BGR->RGB flip, Resize(80) -> ToTensor -> Normalize
torchvision's Resize on a PIL image is `Image.resize(size, BILINEAR)`; the arithmetic restated here is Pillow's
(third-party dependency of the reference, version unpinned by it; Pillow 12.2.0 in this image) src/libImaging/Resample.c:
`precompute_coeffs` (double-precision triangle filter, support scaled by max(1, in/out)), `normalize_coeffs_8bpc`
(22-bit fixed point) and the two 8-bit passes `ImagingResampleHorizontal_8bpc` then `ImagingResampleVertical_8bpc`, each
rounding to uint8.  Pinned: tests/test_transform_cpu.py checks it bit-for-bit against Pillow itself and against the
committed vectors tests/golden/transform_pil.npz (made by tests/golden/make_transform_golden.py).
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)      # mini_imagenet.py:43-44
STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)


def bilinear_coeffs(in_size: int, out_size: int):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the BILINEAR filter (support 1.0), box = (0, in_size).
    Returns xmin [out], count [out], coef [out, ksize] int32."""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    xmin = np.zeros(out_size, np.int32)
    cnt = np.zeros(out_size, np.int32)
    coef = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        lo = int(center - support + 0.5)
        lo = max(lo, 0)
        hi = int(center + support + 0.5)
        hi = min(hi, in_size)
        n = hi - lo
        w = np.zeros(n, np.float64)
        for x in range(n):
            a = abs((x + lo - center + 0.5) * ss)
            w[x] = 1.0 - a if a < 1.0 else 0.0
        ww = 0.0
        for x in range(n):
            ww += w[x]
        if ww != 0.0:
            w = w / ww
        for x in range(n):
            v = w[x] * (1 << PRECISION_BITS)
            coef[xx, x] = int(-0.5 + v) if w[x] < 0 else int(0.5 + v)
        xmin[xx], cnt[xx] = lo, n
    return xmin, cnt, coef


def _pass(img, xmin, cnt, coef, axis):
    """One 8-bit resampling pass along `axis` of img [H, W, C] uint8."""
    img = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((len(xmin),) + img.shape[1:], np.uint8)
    for o in range(len(xmin)):
        acc = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for k in range(cnt[o]):
            acc += img[xmin[o] + k] * int(coef[o, k])
        out[o] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def pil_resize_bilinear(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """img [H, W, 3] uint8 -> [out_h, out_w, 3] uint8, == np.asarray(Image.fromarray(img).resize((out_w, out_h), BILINEAR))."""
    h, w = img.shape[:2]
    if w != out_w:                                    # horizontal pass first (Resample.c ImagingResample)
        img = _pass(img, *bilinear_coeffs(w, out_w), axis=1)
    if h != out_h:
        img = _pass(img, *bilinear_coeffs(h, out_h), axis=0)
    return img


def eval_transform(img: np.ndarray, resize: int, crop: int) -> np.ndarray:
    """uint8 [H, W, 3] -> float32 [3, crop, crop]: Resize((resize, resize)) -> CenterCrop(crop) -> ToTensor -> Normalize."""
    r = pil_resize_bilinear(img, resize, resize)
    o = int(round((resize - crop) / 2.0))              # torchvision center_crop offset
    r = r[o:o + crop, o:o + crop]
    t = r.astype(np.float32) / np.float32(255.0)
    t = (t - MEAN) / STD
    return np.ascontiguousarray(t.transpose(2, 0, 1))
