"""Device-side counterpart of `data/preprocess.py`.

The reference builds torchvision pipelines `Resize((224, 224))` -> `ToTensor()` -> `Normalize(IMAGENET_MEAN, IMAGENET_STD)`
(validation / inference: data/preprocess.py:88-92,116-121; api/inference.py:140-170) and `Resize(256) -> RandomCrop(224) ->
RandomHorizontalFlip -> ColorJitter -> ToTensor -> Normalize` (training: :66-87) that run per image on the CPU with
`num_workers=0`, and `vqa_collate_fn` (:285-315) stacks the float tensors.  At > 3x10^4 pairs/s that path is the bottleneck.
Here the host keeps only the JPEG decode: the decoded uint8 HWC images (any sizes) are uploaded once and
  * `DeviceImageResizer` (`vqa_image_resize`): Resize -- PIL.Image.resize(BILINEAR) restated bit-exactly -- [+ crop window + flip]
    fused with ToTensor + Normalize and the NCHW layout, for a ragged batch;
  * `DeviceImageNormalizer` (`vqa_image_normalize`): ToTensor + Normalize (+ flip) for a batch that is already resized.
Both are bit-identical to the reference transforms (tests/golden/resize_pil.npz, input_pipeline.npz).  ColorJitter is not
restated (PIL enhancer code; augmentation only).  `gpu_collate_fn` is the `vqa_collate_fn` counterpart: same dict keys and dtypes.
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


class DeviceImageResizer:
    """transforms.Resize((size, size)) [-> RandomCrop(crop) -> RandomHorizontalFlip] -> ToTensor -> Normalize on the GPU for a list of
    decoded uint8 [H_i, W_i, 3] images of ANY sizes (data/preprocess.py:66-92,116-121).  `__call__(images, crop_yx=None, flip=None,
    return_u8=False)` -> float32 [B, 3, out, out] (and the uint8 [B, out, out, 3] PIL would return, for inspection / parity).
    size: the Resize target (224; 256 in the augmented pipeline); crop: RandomCrop size (None: no crop); crop_yx: [B][2] window
    origins inside the resized image (host ints), flip: [B] flags."""

    def __init__(self, size: int = 224, crop: Optional[int] = None, mean: Sequence[float] = IMAGENET_MEAN, std: Sequence[float] = IMAGENET_STD,
                 device="cuda"):
        self.size, self.out = int(size), int(crop or size)
        if self.out > self.size:
            raise ValueError("crop must not exceed the resize target")
        self.mean, self.std = [float(m) for m in mean], [float(s) for s in std]
        self.device = torch.device(device)
        self._L = _pkg()._lib

    def __call__(self, images, crop_yx=None, flip: Optional[torch.Tensor] = None, return_u8: bool = False):
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
        else:                                        # one staging buffer, one upload
            host = torch.empty(off, dtype=torch.uint8, pin_memory=True)
            for t, o in zip(ts, offs):
                host[o: o + t.numel()] = t.reshape(-1).cpu()
            packed = host.to(self.device, non_blocking=True)
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
        images = resizer([item[0] for item in batch], crop_yx=crop_yx, flip=flip)
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
