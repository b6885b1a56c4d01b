"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's input-side conventions.

* `encode` / `batch_encode` restate `Tokenizer.preprocess/tokenize/encode/batch_encode` (utils/tokenizer.py:96-137,196-250,
  312-333): lower-case, every character that is not a word character, whitespace or an apostrophe becomes a space, split on
  whitespace, START + words + END, truncate to max_length with END forced onto the last slot, pad with PAD; mask 1 = real.
  PINNED by tests/golden/input_pipeline.npz, written by the real reference class (tests/golden/make_golden.py gen_input).
* `to_tensor_normalize` restates torchvision's ToTensor + Normalize with the reference's constants (data/preprocess.py:34-35,
  117-121).  torchvision is not installed in the build container, so the image half is pinned only against this torch-only
  restatement of torchvision's documented formulas: PARITY UNPINNED against the reference for the image transform
  (the Resize / ColorJitter steps in front of it are PIL code and are not restated at all).
Only tests/ may import this module.
"""
import re
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

PAD_IDX, UNK_IDX, START_IDX, END_IDX = 0, 1, 2, 3           # utils/tokenizer.py:40-43
IMAGENET_MEAN = (0.485, 0.456, 0.406)                        # data/preprocess.py:34
IMAGENET_STD = (0.229, 0.224, 0.225)                         # data/preprocess.py:35


def tokenize(text: str) -> List[str]:                        # utils/tokenizer.py:96-137
    text = re.sub(r"[^\w\s']", " ", text.lower())
    return re.sub(r"\s+", " ", text).strip().split()


def encode(text: str, word2idx: Dict[str, int], max_length: int, add_special_tokens: bool = True) -> Tuple[List[int], List[int]]:
    """utils/tokenizer.py:196-250 with padding=True, truncation=True (the defaults every caller uses)."""
    toks = tokenize(text)
    ids = [word2idx.get(t, UNK_IDX) for t in toks]
    if add_special_tokens:
        ids = [START_IDX] + ids + [END_IDX]
    if len(ids) > max_length:
        ids = ids[:max_length]
        if add_special_tokens:
            ids[-1] = END_IDX
    mask = [1] * len(ids)
    pad = max_length - len(ids)
    return ids + [PAD_IDX] * pad, mask + [0] * pad


def batch_encode(texts: Sequence[str], word2idx: Dict[str, int], max_length: int, add_special_tokens: bool = True):
    out = [encode(t, word2idx, max_length, add_special_tokens) for t in texts]
    return np.array([o[0] for o in out], dtype=np.int64), np.array([o[1] for o in out], dtype=np.int64)


def to_tensor_normalize(img_u8_hwc: torch.Tensor, flip=None) -> torch.Tensor:
    """uint8 [B,H,W,3] -> float32 [B,3,H,W]: ToTensor (permute, /255) then Normalize ((x - mean) / std); flip[b]: horizontal flip first."""
    x = img_u8_hwc
    if flip is not None:
        x = torch.where(torch.as_tensor(flip, dtype=torch.bool)[:, None, None, None], x.flip(2), x)
    t = x.permute(0, 3, 1, 2).contiguous().to(torch.float32).div(255)
    mean = torch.as_tensor(IMAGENET_MEAN, dtype=torch.float32)[None, :, None, None]
    std = torch.as_tensor(IMAGENET_STD, dtype=torch.float32)[None, :, None, None]
    return t.sub(mean).div(std)


# --------------------------------------------------------------------------------------------------------------------------
# transforms.Resize((224, 224)) on a PIL image (data/preprocess.py:90,118; api/inference.py:140-170) is
# PIL.Image.resize(size, BILINEAR).  The algorithm lives in Pillow (third-party, not in /root/reference): src/libImaging/Resample.c of
# the Pillow pinned by this image (12.2.0): precompute_coeffs (double precision, support scaled by the down-scale factor),
# normalize_coeffs_8bpc (fixed point, PRECISION_BITS = 32 - 8 - 2), ImagingResampleHorizontal_8bpc then ImagingResampleVertical_8bpc
# with a uint8-ROUNDED intermediate image, each pass skipped when that axis keeps its size.  Restated here in numpy integer
# arithmetic and PINNED bit-exactly by tests/golden/resize_pil.npz, written by the real PIL.Image.resize in this container
# (tests/golden/make_golden.py gen_resize).
# --------------------------------------------------------------------------------------------------------------------------
PRECISION_BITS = 32 - 8 - 2


def pil_bilinear_coeffs(in_size: int, out_size: int):
    """precompute_coeffs + normalize_coeffs_8bpc for the BILINEAR filter (support 1.0) over the whole axis (box = 0 .. in_size).
    Returns (xmin int32 [out], xn int32 [out], k int32 [out][ksize])."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    xmin = np.zeros(out_size, np.int32); xn = np.zeros(out_size, np.int32); kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        lo = int(center - support + 0.5)                     # C (int) cast: truncation toward zero
        lo = max(lo, 0)
        hi = int(center + support + 0.5)
        hi = min(hi, in_size)
        n = hi - lo
        w = np.zeros(n, np.float64)
        ww = 0.0
        for x in range(n):
            a = (x + lo - center + 0.5) * ss
            a = -a if a < 0.0 else a
            w[x] = 1.0 - a if a < 1.0 else 0.0
            ww += w[x]
        if ww != 0.0:
            w = w / ww
        for x in range(n):
            v = w[x] * (1 << PRECISION_BITS)
            kk[xx, x] = int(-0.5 + v) if w[x] < 0 else int(0.5 + v)
        xmin[xx], xn[xx] = lo, n
    return xmin, xn, kk


def _resample_axis0(img: np.ndarray, out_size: int) -> np.ndarray:
    """One 8bpc pass along axis 0 of uint8 [n][...]: clip8((1 << (PRECISION_BITS-1)) + sum k*pixel >> PRECISION_BITS)."""
    xmin, xn, kk = pil_bilinear_coeffs(img.shape[0], out_size)
    out = np.empty((out_size,) + img.shape[1:], np.uint8)
    src = img.astype(np.int64)
    for o in range(out_size):
        acc = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(int(xn[o])):
            acc += src[xmin[o] + x] * int(kk[o, x])
        out[o] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def pil_resize_bilinear(img_u8_hwc: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """PIL.Image.fromarray(img).resize((out_w, out_h), Image.BILINEAR) for a uint8 [H][W][C] array: horizontal pass first (uint8
    intermediate), then vertical; a pass whose axis already has the target size is skipped (ImagingResample)."""
    x = np.ascontiguousarray(img_u8_hwc)
    H, W = x.shape[:2]
    if W != out_w:
        x = np.swapaxes(_resample_axis0(np.swapaxes(x, 0, 1), out_w), 0, 1)
    if H != out_h:
        x = _resample_axis0(x, out_h)
    return np.ascontiguousarray(x)


def pattern_image(H: int, W: int, seed: int, noise_bits: int = 5) -> np.ndarray:
    """Deterministic uint8 [H][W][3] test image from INTEGER arithmetic only (identical on every platform / numpy version, so the
    fixture generator and the tests rebuild the same inputs without storing them): a wrapped gradient with per-channel slopes plus
    splitmix64 noise of `noise_bits` bits (8: pure noise)."""
    y, x, c = np.meshgrid(np.arange(H, dtype=np.uint64), np.arange(W, dtype=np.uint64), np.arange(3, dtype=np.uint64), indexing="ij")
    idx = (y * np.uint64(W) + x) * np.uint64(3) + c
    with np.errstate(over="ignore"):
        z = idx + np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    noise = (z >> np.uint64(64 - noise_bits)).astype(np.int64) if noise_bits else np.zeros_like(idx, dtype=np.int64)
    base = (y.astype(np.int64) * (3 + seed % 5) + x.astype(np.int64) * (2 + c.astype(np.int64)) + c.astype(np.int64) * 85) % 256
    if noise_bits >= 8:
        return (noise & 255).astype(np.uint8)
    return ((base + noise) % 256).astype(np.uint8)


# the cases of tests/golden/resize_pil.npz: (tag, H, W, resize target, crop size | None, crop origin (y, x), flip, seed, noise bits)
RESIZE_CASES = [
    ("down_both", 480, 640, 224, None, (0, 0), 0, 1, 5),
    ("up_both", 100, 150, 224, None, (0, 0), 0, 2, 5),
    ("identity", 96, 96, 96, None, (0, 0), 0, 3, 8),
    ("h_only", 224, 300, 224, None, (0, 0), 0, 4, 5),
    ("v_only", 500, 224, 224, None, (0, 0), 0, 5, 5),
    ("extreme_aspect", 37, 1000, 96, None, (0, 0), 0, 6, 8),
    ("aug_crop_flip", 333, 500, 256, 224, (16, 5), 1, 7, 5),
    ("one_pixel", 1, 1, 224, None, (0, 0), 0, 8, 8),
    ("off_by_one", 97, 95, 96, None, (0, 0), 0, 9, 8),
    ("noise_down3x", 300, 290, 96, None, (0, 0), 1, 10, 8),
    ("noise_up", 40, 56, 96, None, (0, 0), 0, 11, 8),
]
