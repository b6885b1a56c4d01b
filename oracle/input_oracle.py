"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's input-side conventions.

* `encode` / `batch_encode` restate `Tokenizer.preprocess/tokenize/encode/batch_encode` (utils/tokenizer.py:96-137,196-250,
  312-333): lower-case, every character that is not a word character, whitespace or an apostrophe becomes a space, split on
  whitespace, START + words + END, truncate to max_length with END forced onto the last slot, pad with PAD; mask 1 = real.
  PINNED by tests/golden/input_pipeline.npz, written by the real reference class (tests/golden/make_golden.py gen_input).
* `to_tensor_normalize` restates torchvision's ToTensor + Normalize with the reference's constants (data/preprocess.py:34-35,
  117-121).  torchvision is not installed in the build container, so the image half is pinned only against this torch-only
  restatement of torchvision's documented formulas: PARITY UNPINNED against torchvision for that step.
* `pil_resize_bilinear` restates transforms.Resize and `color_jitter` the four ColorJitter adjustments (data/preprocess.py:66-84):
  on PIL images torchvision runs PIL code for both, PIL is installed here, and both restatements are PINNED bit-exactly by outputs of
  the real PIL (tests/golden/resize_pil.npz, tests/golden/jitter_pil.npz; the colour-space conversions over all 2^24 inputs).
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


# --------------------------------------------------------------------------------------------------------------------------
# transforms.ColorJitter(brightness=0.2, contrast=0.2, saturation=0.2, hue=0.1) (data/preprocess.py:77-82).  On a PIL image torchvision's
# functional ops are Pillow code (third-party, not in /root/reference; Pillow 12.2.0 is installed in this image):
#   adjust_brightness = ImageEnhance.Brightness(img).enhance(f)   = Image.blend(black, img, f)
#   adjust_contrast   = ImageEnhance.Contrast(img).enhance(f)     = Image.blend(gray(int(mean(L(img)) + 0.5)), img, f)
#   adjust_saturation = ImageEnhance.Color(img).enhance(f)        = Image.blend(L(img) as RGB, img, f)
#   adjust_hue        = img.convert("HSV"), H += uint8(f * 255) with uint8 wrap-around, .convert("RGB")
# ColorJitter.forward applies them in the order of a random permutation `fn_idx` (0 brightness, 1 contrast, 2 saturation, 3 hue) with
# factors drawn uniformly from [1 - x, 1 + x] (hue: [-x, x]).  The C side (libImaging Blend.c, Convert.c rgb2l / rgb2hsv_row / hsv2rgb)
# is restated below and PINNED against the real PIL: `pil_rgb_to_l`, `pil_rgb_to_hsv`, `pil_hsv_to_rgb` over ALL 2^24 inputs and
# `pil_blend` over every (a, b) byte pair for a set of factors (tests/test_input_cpu.py, which imports PIL), whole compositions by
# tests/golden/jitter_pil.npz (written by PIL's own ImageEnhance / convert in tests/golden/make_golden.py gen_jitter).
# torchvision itself is absent: the permutation / factor semantics follow its documented behaviour and are PARITY UNPINNED.
# --------------------------------------------------------------------------------------------------------------------------
JIT_BRIGHTNESS, JIT_CONTRAST, JIT_SATURATION, JIT_HUE = 0, 1, 2, 3


def pil_rgb_to_l(rgb: np.ndarray) -> np.ndarray:
    """Convert.c rgb2l: ITU-R 601-2 luma in 16.16 fixed point, rounded."""
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def pil_blend(degenerate: np.ndarray, image: np.ndarray, alpha: float) -> np.ndarray:
    """Blend.c ImagingBlend(in1 = degenerate, in2 = image, alpha): float arithmetic (alpha is a C float; the product and the sum are
    separate float operations), TRUNCATED to uint8 for 0 <= alpha <= 1, clipped then truncated outside; alpha == 0 / 1 are copies."""
    alpha = np.float32(alpha)
    if alpha == 0:
        return degenerate.copy()
    if alpha == 1:
        return image.copy()
    a, b = degenerate.astype(np.int32), image.astype(np.int32)
    t = (a.astype(np.float32) + (alpha * (b - a).astype(np.float32)).astype(np.float32)).astype(np.float32)
    if 0 <= alpha <= 1:
        return t.astype(np.int32).astype(np.uint8)
    return np.where(t <= 0, 0, np.where(t >= 255, 255, t.astype(np.int32))).astype(np.uint8)


def pil_rgb_to_hsv(rgb: np.ndarray) -> np.ndarray:
    """Convert.c rgb2hsv_row: s and the three (max - c) / (max - min) quotients in float, the hue sum and fmod(h / 6 + 1, 1) in
    double (the C literals are doubles) stored back to float, (int)(x * 255) truncated and clipped."""
    f32, f64 = np.float32, np.float64
    r, g, b = (rgb[..., i].astype(np.int32) for i in range(3))
    maxc = np.maximum(r, np.maximum(g, b)); minc = np.minimum(r, np.minimum(g, b))
    gray = maxc == minc
    cr = np.where(gray, 1, maxc - minc).astype(f32)
    s = cr / np.where(maxc == 0, 1, maxc).astype(f32)
    rc = (maxc - r).astype(f32) / cr; gc = (maxc - g).astype(f32) / cr; bc = (maxc - b).astype(f32) / cr
    h = np.where(r == maxc, bc.astype(f64) - gc.astype(f64),
                 np.where(g == maxc, (2.0 + rc.astype(f64)) - bc.astype(f64), (4.0 + gc.astype(f64)) - rc.astype(f64))).astype(f32)
    h = np.fmod(h.astype(f64) / 6.0 + 1.0, 1.0).astype(f32)
    uh = np.clip((h.astype(f64) * 255.0).astype(np.int32), 0, 255)
    us = np.clip((s.astype(f64) * 255.0).astype(np.int32), 0, 255)
    return np.stack([np.where(gray, 0, uh), np.where(gray, 0, us), maxc], -1).astype(np.uint8)


def pil_hsv_to_rgb(hsv: np.ndarray) -> np.ndarray:
    """Convert.c hsv2rgb: sector i = floor(h * 6 / 255), f its remainder and s / 255 as floats, p / q / t = round(v * (1 - ...)) in
    double, channel order by sector (i % 6); s == 0 is gray."""
    f32, f64 = np.float32, np.float64
    h, s, v = (hsv[..., i].astype(np.int32) for i in range(3))
    hf = h.astype(f64) * 6.0 / 255.0
    i = np.floor(hf)
    f = (hf - i).astype(f32).astype(f64)
    fs = (s.astype(f64) / 255.0).astype(f32).astype(f64)
    vf = v.astype(f64)
    rnd = lambda x: np.clip(np.where(x >= 0, np.floor(x + 0.5), np.ceil(x - 0.5)), 0, 255).astype(np.int32)     # C round(): half away from zero
    p, q, t = rnd(vf * (1.0 - fs)), rnd(vf * (1.0 - fs * f)), rnd(vf * (1.0 - fs * (1.0 - f)))
    ii = i.astype(np.int32) % 6
    R = np.choose(ii, [v, q, p, p, t, v]); G = np.choose(ii, [t, v, v, q, p, p]); B = np.choose(ii, [p, p, t, v, v, q])
    out = np.where((s == 0)[..., None], np.stack([v, v, v], -1), np.stack([R, G, B], -1))
    return out.astype(np.uint8)


def hue_delta(hue_factor: float) -> int:
    """torchvision's adjust_hue on PIL images adds uint8(hue_factor * 255) to the H band with uint8 wrap-around: the product is
    truncated toward zero, a negative value wraps modulo 256."""
    return int(hue_factor * 255) % 256


def color_jitter(img_u8_hwc: np.ndarray, order: Sequence[int], brightness=None, contrast=None, saturation=None, hue=None) -> np.ndarray:
    """ColorJitter.forward on one uint8 [H][W][3] image for given factors (None = that adjustment is off) and permutation `order`."""
    x = np.ascontiguousarray(img_u8_hwc)
    for fn in order:
        if fn == JIT_BRIGHTNESS and brightness is not None:
            x = pil_blend(np.zeros_like(x), x, brightness)
        elif fn == JIT_CONTRAST and contrast is not None:
            L = pil_rgb_to_l(x)
            mean = int(float(int(L.astype(np.int64).sum())) / L.size + 0.5)         # ImageStat.Stat(L).mean[0]: float sum / int count
            x = pil_blend(np.full_like(x, mean), x, contrast)
        elif fn == JIT_SATURATION and saturation is not None:
            x = pil_blend(np.repeat(pil_rgb_to_l(x)[..., None], 3, -1), x, saturation)
        elif fn == JIT_HUE and hue is not None:
            if not -0.5 <= hue <= 0.5:
                raise ValueError(f"hue_factor ({hue}) is not in [-0.5, 0.5].")
            hsv = pil_rgb_to_hsv(x)
            hsv[..., 0] = ((hsv[..., 0].astype(np.int32) + hue_delta(hue)) & 255).astype(np.uint8)
            x = pil_hsv_to_rgb(hsv)
    return x


# the cases of tests/golden/jitter_pil.npz: (tag, H, W, seed, noise bits, order, brightness, contrast, saturation, hue)
JITTER_CASES = [
    ("all_0123", 224, 224, 21, 5, (0, 1, 2, 3), 1.13, 0.87, 1.19, 0.07),
    ("all_3210", 224, 224, 22, 5, (3, 2, 1, 0), 0.81, 1.2, 0.8, -0.1),
    ("all_2031", 96, 160, 23, 8, (2, 0, 3, 1), 1.2, 0.8, 1.07, 0.033),
    ("all_1302", 57, 41, 24, 8, (1, 3, 0, 2), 0.9, 1.05, 0.95, -0.049),
    ("bright_only", 64, 64, 25, 8, (0, 1, 2, 3), 1.37, None, None, None),
    ("contrast_only", 64, 64, 26, 5, (0, 1, 2, 3), None, 0.55, None, None),
    ("sat_only", 64, 64, 27, 8, (0, 1, 2, 3), None, None, 1.9, None),
    ("hue_only_neg", 64, 64, 28, 8, (0, 1, 2, 3), None, None, None, -0.5),
    ("hue_only_pos", 64, 64, 29, 8, (0, 1, 2, 3), None, None, None, 0.5),
    ("identity_factors", 32, 48, 30, 8, (2, 1, 3, 0), 1.0, 1.0, 1.0, 0.0),
    ("zero_factors", 32, 48, 31, 8, (0, 2, 1, 3), 0.0, 0.0, 0.0, 0.002),
]

# one case of tests/golden/jitter_pil.npz runs the WHOLE training transform of data/preprocess.py:66-84 in the real PIL:
# (tag, H, W, seed, noise bits, resize, crop, (cy, cx), flip, order, brightness, contrast, saturation, hue)
PIPELINE_CASES = [
    ("train_pipeline_a", 333, 500, 41, 5, 256, 224, (16, 5), 1, (1, 3, 0, 2), 1.17, 0.83, 1.1, -0.08),
    ("train_pipeline_b", 480, 360, 42, 5, 256, 224, (0, 32), 0, (3, 0, 2, 1), 0.84, 1.19, 0.81, 0.1),
]
