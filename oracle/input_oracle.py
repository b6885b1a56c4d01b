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
