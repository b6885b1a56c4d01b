"""Device-side counterpart of the tensor half of `data/preprocess.py`.

The reference builds torchvision pipelines ending in `ToTensor()` + `Normalize(IMAGENET_MEAN, IMAGENET_STD)`
(data/preprocess.py:34-35,64-121) that run per image on the CPU with `num_workers=0`, and `vqa_collate_fn` (:285-315) stacks
the float tensors.  At > 3x10^4 pairs/s that path is the bottleneck (77 KB of float32 per image cross PCIe instead of 19 KB of
bytes).  Here the host keeps only decode + resize (PIL); the uint8 HWC batch is uploaded once and `vqa_image_normalize`
(include/vqa_hip.h) produces the normalised NCHW float batch on the GPU, bit-identical to ToTensor + Normalize, with the
optional RandomHorizontalFlip folded in.  `gpu_collate_fn` is the `vqa_collate_fn` counterpart: same dict keys and dtypes.
There is no CPU path: host tensors are uploaded, the transform itself only runs on the GPU.
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


def gpu_collate_fn(batch, device="cuda", normalizer: Optional[DeviceImageNormalizer] = None, flip_p: float = 0.0, generator=None):
    """`vqa_collate_fn` (data/preprocess.py:285-315) for items whose image is still a uint8 HWC array / tensor (decoded and resized,
    not yet ToTensor'ed): (image_u8 [H,W,3], token_ids [L], attention_mask [L], answer_idx).  Returns the same dict
    ('images' float32 [B,3,H,W] normalised on the GPU, 'token_ids', 'attention_mask', 'answers' int64), all on `device`."""
    norm = normalizer or DeviceImageNormalizer()
    imgs = torch.stack([torch.as_tensor(item[0]) for item in batch]).to(device, non_blocking=True)
    flip = None
    if flip_p > 0.0:
        flip = torch.rand(len(batch), generator=generator) < flip_p
    return {
        "images": norm(imgs, flip),
        "token_ids": torch.stack([torch.as_tensor(item[1], dtype=torch.long) for item in batch]).to(device),
        "attention_mask": torch.stack([torch.as_tensor(item[2], dtype=torch.long) for item in batch]).to(device),
        "answers": torch.tensor([int(item[3]) for item in batch], dtype=torch.long, device=device),
    }
