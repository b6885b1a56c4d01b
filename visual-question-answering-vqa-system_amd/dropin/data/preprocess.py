"""Device-side counterpart of `data/preprocess.py`.

The reference builds torchvision pipelines `Resize((224, 224))` -> `ToTensor()` -> `Normalize(IMAGENET_MEAN, IMAGENET_STD)`
(validation / inference: data/preprocess.py:88-92,116-121; api/inference.py:140-170) and `Resize(256) -> RandomCrop(224) ->
RandomHorizontalFlip -> ColorJitter -> ToTensor -> Normalize` (training: :66-87) that run per image on the CPU with
`num_workers=0`, and `vqa_collate_fn` (:285-315) stacks the float tensors.  At > 3x10^4 pairs/s that path is the bottleneck.
Here the host keeps only the JPEG decode: the decoded uint8 HWC images (any sizes) are uploaded once and
  * `DeviceImageResizer` (`vqa_image_resize`): Resize -- PIL.Image.resize(BILINEAR) restated bit-exactly -- [+ crop window + flip]
    fused with ToTensor + Normalize and the NCHW layout, for a ragged batch;
  * `DeviceColorJitter` (`vqa_image_color_jitter`): ColorJitter(brightness, contrast, saturation, hue) -- PIL's ImageEnhance blends and
    HSV hue shift restated bit-exactly -- fused with ToTensor + Normalize; `DeviceImageResizer(..., jitter=...)` chains it behind the
    crop / flip, which makes the whole training pipeline of data/preprocess.py:66-87 device-side;
  * `DeviceImageNormalizer` (`vqa_image_normalize`): ToTensor + Normalize (+ flip) for a batch that is already resized.
All are bit-identical to the PIL / torch code the reference transforms run (tests/golden/resize_pil.npz, jitter_pil.npz,
input_pipeline.npz); the random draws (crop origin, flip, jitter permutation and factors) follow torchvision's documented
distributions with torch's generator, not its exact call sequence (torchvision is absent: unpinned).
`gpu_collate_fn` is the `vqa_collate_fn` counterpart: same dict keys and dtypes.
There is no CPU path: host tensors are uploaded, the transforms themselves only run on the GPU.
"""
from __future__ import annotations

import importlib
import os
from typing import Optional, Sequence

import torch

IMAGENET_MEAN = [0.485, 0.456, 0.406]          # data/preprocess.py:34
IMAGENET_STD = [0.229, 0.224, 0.225]           # data/preprocess.py:35


def _pkg():
    import sys
    here = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    root = os.path.dirname(here)
    if root not in sys.path:
        sys.path.insert(0, root)
    return importlib.import_module(os.path.basename(here))


class DeviceImageNormalizer:
    """uint8 [B, H, W, 3] (cuda) -> float32 [B, 3, H, W] normalised; `flip`: optional bool/uint8 [B] (horizontal flip per sample)."""

    def __init__(self, mean: Sequence[float] = IMAGENET_MEAN, std: Sequence[float] = IMAGENET_STD):
        self.mean, self.std = [float(m) for m in mean], [float(s) for s in std]
        self._L = _pkg()._lib

    def __call__(self, images_u8: torch.Tensor, flip: Optional[torch.Tensor] = None) -> torch.Tensor:
        if images_u8.dtype != torch.uint8 or images_u8.dim() != 4 or images_u8.shape[-1] != 3:
            raise RuntimeError("DeviceImageNormalizer expects a uint8 [B, H, W, 3] batch")
        if not images_u8.is_cuda:
            raise RuntimeError("DeviceImageNormalizer runs on the GPU only: upload the uint8 batch first (no CPU fallback)")
        x = images_u8.contiguous()
        B, H, W, _ = x.shape
        out = torch.empty((B, 3, H, W), device=x.device, dtype=torch.float32)
        f = None
        if flip is not None:
            if flip.numel() != B:
                raise RuntimeError(f"DeviceImageNormalizer: `flip` must hold one flag per sample ({B}), got {flip.numel()}")
            f = flip.to(device=x.device, dtype=torch.uint8).contiguous()
        self._L.call("vqa_image_normalize", x.data_ptr(), out.data_ptr(), None if f is None else f.data_ptr(), B, H, W,
                     *self.mean, *self.std)
        return out


class DeviceColorJitter:
    """transforms.ColorJitter(brightness, contrast, saturation, hue) -> ToTensor -> Normalize on the GPU for a uint8 [B, H, W, 3] batch
    (data/preprocess.py:77-84).  Arguments as torchvision's: a float x means factors drawn from [max(0, 1 - x), 1 + x] (hue: [-x, x],
    x <= 0.5), a (lo, hi) pair is used as is, 0 / None switches the adjustment off.  `__call__(images_u8, order=None, factors=None,
    generator=None, return_u8=False)`: per-image permutations `order` ([B, 4] of 0 brightness, 1 contrast, 2 saturation, 3 hue) and
    `factors` ([B, 4]: brightness, contrast, saturation factor, hue factor; NaN = off) are drawn here unless given."""

    def __init__(self, brightness=0.0, contrast=0.0, saturation=0.0, hue=0.0, mean: Sequence[float] = IMAGENET_MEAN,
                 std: Sequence[float] = IMAGENET_STD):
        def rng(v, name, center=1.0, bound=(0.0, float("inf")), clip_first=True):
            if v is None:
                return None
            if isinstance(v, (int, float)):
                if v < 0:
                    raise ValueError(f"If {name} is a single number, it must be non negative.")
                lo, hi = center - float(v), center + float(v)
                if clip_first:
                    lo = max(lo, 0.0)
            else:
                lo, hi = float(v[0]), float(v[1])
            if not bound[0] <= lo <= hi <= bound[1]:
                raise ValueError(f"{name} values should be between {bound}, but got ({lo}, {hi}).")
            return None if lo == hi == center else (lo, hi)
        self.brightness = rng(brightness, "brightness")
        self.contrast = rng(contrast, "contrast")
        self.saturation = rng(saturation, "saturation")
        self.hue = rng(hue, "hue", center=0.0, bound=(-0.5, 0.5), clip_first=False)
        self.mean, self.std = [float(m) for m in mean], [float(s) for s in std]
        self._L = _pkg()._lib

    def draw(self, B: int, generator=None):
        """Per-image permutation and factors (host tensors): a uniformly random order of the four adjustments and uniform factors, the
        distributions of ColorJitter.get_params, drawn for the whole batch at once (argsort of iid uniforms = a uniform permutation)."""
        order = torch.rand(B, 4, generator=generator).argsort(1).to(torch.uint8)
        cols = []
        for r in (self.brightness, self.contrast, self.saturation, self.hue):
            cols.append(torch.full((B,), float("nan"), dtype=torch.float64) if r is None
                        else torch.empty(B, dtype=torch.float64).uniform_(r[0], r[1], generator=generator))
        return order, torch.stack(cols, 1)

    def __call__(self, images_u8: torch.Tensor, order=None, factors=None, generator=None, return_u8: bool = False):
        x = images_u8
        if x.dtype != torch.uint8 or x.dim() != 4 or x.shape[-1] != 3:
            raise RuntimeError("DeviceColorJitter expects a uint8 [B, H, W, 3] batch")
        if not x.is_cuda:
            raise RuntimeError("DeviceColorJitter runs on the GPU only: upload the uint8 batch first (no CPU fallback)")
        x = x.contiguous()
        B, H, W, _ = x.shape
        if order is None or factors is None:
            order, factors = self.draw(B, generator)
        order = torch.as_tensor(order, dtype=torch.uint8).reshape(-1, 4)
        factors = torch.as_tensor(factors, dtype=torch.float64).reshape(-1, 4).clone()
        if order.shape[0] != B or factors.shape[0] != B:
            raise RuntimeError(f"DeviceColorJitter: `order` / `factors` must hold one row per image ({B})")
        hue = factors[:, 3]
        if bool(((hue < -0.5) | (hue > 0.5)).any()):
            raise ValueError("hue_factor is not in [-0.5, 0.5].")
        # torchvision adds uint8(hue_factor * 255) to the H band: truncation toward zero in double, wrap modulo 256
        factors[:, 3] = torch.where(torch.isnan(hue), hue, torch.remainder(torch.trunc(hue * 255.0), 256.0))
        od = order.to(x.device).contiguous()
        fd = factors.to(torch.float32).to(x.device).contiguous()       # Image.blend takes its factor as a C float
        sums = torch.empty(B, dtype=torch.int64, device=x.device)
        out = torch.empty((B, 3, H, W), dtype=torch.float32, device=x.device)
        u8 = torch.empty_like(x) if return_u8 else None
        self._L.call("vqa_image_color_jitter", x.data_ptr(), od.data_ptr(), fd.data_ptr(), B, H, W, None if u8 is None else u8.data_ptr(),
                     out.data_ptr(), *self.mean, *self.std, sums.data_ptr())
        return (out, u8) if return_u8 else out


class DeviceImageResizer:
    """transforms.Resize((size, size)) [-> RandomCrop(crop) -> RandomHorizontalFlip] -> ToTensor -> Normalize on the GPU for a list of
    decoded uint8 [H_i, W_i, 3] images of ANY sizes (data/preprocess.py:66-92,116-121).  `__call__(images, crop_yx=None, flip=None,
    return_u8=False)` -> float32 [B, 3, out, out] (and the uint8 [B, out, out, 3] PIL would return, for inspection / parity).
    size: the Resize target (224; 256 in the augmented pipeline); crop: RandomCrop size (None: no crop); crop_yx: [B][2] window
    origins inside the resized image (host ints), flip: [B] flags; jitter: a `DeviceColorJitter` applied behind crop and flip (the
    training pipeline's order, data/preprocess.py:70-84), `jitter_params` = (order, factors) to fix its draws."""

    def __init__(self, size: int = 224, crop: Optional[int] = None, mean: Sequence[float] = IMAGENET_MEAN, std: Sequence[float] = IMAGENET_STD,
                 device="cuda", jitter: Optional["DeviceColorJitter"] = None):
        self.jitter = jitter
        self.size, self.out = int(size), int(crop or size)
        if self.out > self.size:
            raise ValueError("crop must not exceed the resize target")
        self.mean, self.std = [float(m) for m in mean], [float(s) for s in std]
        self.device = torch.device(device)
        self._L = _pkg()._lib
        self._stage, self._stage_ev, self._stage_i = [None, None], [None, None], 0

    def __call__(self, images, crop_yx=None, flip: Optional[torch.Tensor] = None, return_u8: bool = False, jitter_params=None, generator=None):
        import ctypes as C
        n = len(images)
        if n == 0:
            raise RuntimeError("DeviceImageResizer: empty batch")
        ts = []
        for im in images:
            t = torch.as_tensor(im)
            if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[-1] != 3 or t.shape[0] < 1 or t.shape[1] < 1:
                raise RuntimeError("DeviceImageResizer expects decoded uint8 [H, W, 3] images")
            ts.append(t.contiguous())
        Hs, Ws = [int(t.shape[0]) for t in ts], [int(t.shape[1]) for t in ts]
        offs, off = [], 0
        for t in ts:
            offs.append(off)
            off += (t.numel() + 15) // 16 * 16
        if all(t.is_cuda for t in ts):
            packed = torch.empty(off, dtype=torch.uint8, device=self.device)
            for t, o in zip(ts, offs):
                packed[o: o + t.numel()] = t.reshape(-1)
        else:                                        # one pinned staging buffer (two, alternating: kept across calls), one upload
            k = self._stage_i = 1 - self._stage_i
            if self._stage[k] is None or self._stage[k].numel() < off:
                self._stage[k] = torch.empty(max(off, 1 << 20) * 5 // 4, dtype=torch.uint8, pin_memory=True)   # pinning costs milliseconds: not per call
            elif self._stage_ev[k] is not None:
                self._stage_ev[k].synchronize()      # the upload that last read this buffer has finished
            host = self._stage[k][:off]
            hnp = host.numpy()                       # plain memcpy per image (torch's copy_ fans a 0.6 MB copy out over every OpenMP thread)
            for t, o in zip(ts, offs):
                hnp[o: o + t.numel()] = t.reshape(-1).cpu().numpy()
            packed = host.to(self.device, non_blocking=True)
            self._stage_ev[k] = torch.cuda.Event(); self._stage_ev[k].record()
        IA, LA = C.c_int * n, C.c_longlong * n
        Ha, Wa, Oa = IA(*Hs), IA(*Ws), LA(*offs)
        crop = None
        if crop_yx is not None:
            flat = [int(v) for yx in crop_yx for v in yx]
            if len(flat) != 2 * n:
                raise RuntimeError("crop_yx must hold one (y, x) origin per image")
            crop = (C.c_int * (2 * n))(*flat)
        elif self.out != self.size:
            raise RuntimeError("a crop smaller than the resize target needs crop_yx (the RandomCrop origins)")
        f = None
        if flip is not None:
            if flip.numel() != n:
                raise RuntimeError(f"`flip` must hold one flag per image ({n}), got {flip.numel()}")
            f = flip.to(device=self.device, dtype=torch.uint8).contiguous()
        S, Oo = self.size, self.out
        wsb = self._L.count("vqa_image_resize_ws", n, Ha, Wa, S, S, Oo)
        ws = torch.empty(max(int(wsb), 16), dtype=torch.uint8, device=self.device)
        if self.jitter is not None:                  # Resize / crop / flip leave the uint8 image, the jitter pass normalises
            mid = torch.empty((n, Oo, Oo, 3), dtype=torch.uint8, device=self.device)
            self._L.call("vqa_image_resize", packed.data_ptr(), Oa, Ha, Wa, crop, n, S, S, Oo, Oo, mid.data_ptr(), None,
                         None if f is None else f.data_ptr(), *self.mean, *self.std, ws.data_ptr(), int(wsb))
            order, factors = jitter_params if jitter_params is not None else (None, None)
            return self.jitter(mid, order=order, factors=factors, generator=generator, return_u8=return_u8)
        out = torch.empty((n, 3, Oo, Oo), dtype=torch.float32, device=self.device)
        u8 = torch.empty((n, Oo, Oo, 3), dtype=torch.uint8, device=self.device) if return_u8 else None
        self._L.call("vqa_image_resize", packed.data_ptr(), Oa, Ha, Wa, crop, n, S, S, Oo, Oo, None if u8 is None else u8.data_ptr(),
                     out.data_ptr(), None if f is None else f.data_ptr(), *self.mean, *self.std, ws.data_ptr(), int(wsb))
        return (out, u8) if return_u8 else out


def gpu_collate_fn(batch, device="cuda", normalizer: Optional[DeviceImageNormalizer] = None, flip_p: float = 0.0, generator=None,
                   resizer: Optional[DeviceImageResizer] = None):
    """`vqa_collate_fn` (data/preprocess.py:285-315) for items whose image is still a uint8 HWC array / tensor (decoded, not yet
    transformed): (image_u8 [H,W,3], token_ids [L], attention_mask [L], answer_idx).  With `resizer` the images may have any sizes
    (Resize on the GPU); without it they must already share one size.  Returns the same dict ('images' float32 [B,3,H,W] normalised
    on the GPU, 'token_ids', 'attention_mask', 'answers' int64), all on `device`."""
    flip = None
    if flip_p > 0.0:
        flip = torch.rand(len(batch), generator=generator) < flip_p
    if resizer is not None:
        crop_yx = None
        if resizer.out != resizer.size:              # RandomCrop origins (data/preprocess.py:71), drawn like torchvision: uniform over the valid range
            hi = resizer.size - resizer.out + 1
            crop_yx = torch.randint(0, hi, (len(batch), 2), generator=generator).tolist()
        images = resizer([item[0] for item in batch], crop_yx=crop_yx, flip=flip, generator=generator)
    else:
        norm = normalizer or DeviceImageNormalizer()
        imgs = torch.stack([torch.as_tensor(item[0]) for item in batch]).to(device, non_blocking=True)
        images = norm(imgs, flip)
    return {
        "images": images,
        "token_ids": torch.stack([torch.as_tensor(item[1], dtype=torch.long) for item in batch]).to(device),
        "attention_mask": torch.stack([torch.as_tensor(item[2], dtype=torch.long) for item in batch]).to(device),
        "answers": torch.tensor([int(item[3]) for item in batch], dtype=torch.long, device=device),
    }
