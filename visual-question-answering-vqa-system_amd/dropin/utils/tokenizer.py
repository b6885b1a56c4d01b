"""`utils.tokenizer.Tokenizer` drop-in whose batch encoding is packed on the GPU (reference: utils/tokenizer.py:46-333).

Same vocabulary file format (`save` / `load`: {"word2idx", "max_length", "max_vocab_size"}), same special indices
(PAD 0, UNK 1, START 2, END 3), same text normalisation and the same `encode` conventions.  The string work (lower-casing,
splitting, dictionary lookup) is inherently host-side; what moves to the GPU is the batch assembly: `batch_encode_device`
uploads the ragged vocabulary indices once and `vqa_pack_tokens` (include/vqa_hip.h) writes the padded [B, max_length] ids and
mask tensors there -- the tensors `VQAModel.forward` consumes -- instead of building B Python lists and stacking them.
`encode` / `batch_encode` keep the reference's list-returning signatures for callers that want host lists.
"""
from __future__ import annotations

import importlib
import json
import os
import re
from collections import Counter
from typing import Dict, List, Optional, Sequence, Tuple

import torch

PAD_TOKEN, UNK_TOKEN, START_TOKEN, END_TOKEN = "<PAD>", "<UNK>", "<START>", "<END>"
SPECIAL_TOKENS = [PAD_TOKEN, UNK_TOKEN, START_TOKEN, END_TOKEN]
PAD_IDX, UNK_IDX, START_IDX, END_IDX = 0, 1, 2, 3
_NOT_WORD = re.compile(r"[^\w\s']")
_SPACES = re.compile(r"\s+")


def _pkg():
    import sys
    here = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    root = os.path.dirname(here)
    if root not in sys.path:
        sys.path.insert(0, root)
    return importlib.import_module(os.path.basename(here))


class Tokenizer:
    def __init__(self, max_length: int = 20, vocab_size: Optional[int] = None):
        self.max_length = max_length
        self.max_vocab_size = vocab_size
        self.word2idx: Dict[str, int] = {t: i for i, t in enumerate(SPECIAL_TOKENS)}
        self.idx2word: Dict[int, str] = {i: t for t, i in self.word2idx.items()}
        self._is_fitted = False

    @property
    def vocab_size(self) -> int:
        return len(self.word2idx)

    # ---- text side (host): utils/tokenizer.py:96-137
    @staticmethod
    def preprocess(text: str) -> str:
        return _SPACES.sub(" ", _NOT_WORD.sub(" ", text.lower())).strip()

    def tokenize(self, text: str) -> List[str]:
        return self.preprocess(text).split()

    def build_vocab(self, questions: Sequence[str], min_freq: int = 2) -> None:      # utils/tokenizer.py:139-194
        counts = Counter()
        for qn in questions:
            counts.update(self.tokenize(qn))
        words = sorted((w for w, c in counts.items() if c >= min_freq), key=lambda w: counts[w], reverse=True)   # stable: first-seen order on ties
        if self.max_vocab_size is not None:
            words = words[: self.max_vocab_size - len(SPECIAL_TOKENS)]
        for w in words:
            if w not in self.word2idx:
                self.idx2word[len(self.word2idx)] = w
                self.word2idx[w] = len(self.word2idx)
        self._is_fitted = True

    def _lookup(self, text: str) -> List[int]:
        get = self.word2idx.get
        return [get(t, UNK_IDX) for t in self.tokenize(text)]

    # ---- reference-signature encoders (host lists): utils/tokenizer.py:196-250, 312-333
    def encode(self, text: str, add_special_tokens: bool = True, padding: bool = True, truncation: bool = True) -> Tuple[List[int], List[int]]:
        ids = self._lookup(text)
        if add_special_tokens:
            ids = [START_IDX] + ids + [END_IDX]
        if truncation and len(ids) > self.max_length:
            del ids[self.max_length:]
            if add_special_tokens:
                ids[-1] = END_IDX
        mask = [1] * len(ids)
        if padding and len(ids) < self.max_length:
            fill = self.max_length - len(ids)
            ids += [PAD_IDX] * fill
            mask += [0] * fill
        return ids, mask

    def batch_encode(self, texts: Sequence[str], add_special_tokens: bool = True) -> Tuple[List[List[int]], List[List[int]]]:
        pairs = [self.encode(t, add_special_tokens=add_special_tokens) for t in texts]
        return [p[0] for p in pairs], [p[1] for p in pairs]

    # ---- device batch assembly
    def batch_encode_device(self, texts: Sequence[str], device="cuda", add_special_tokens: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
        """(token_ids, attention_mask) int64 [B, max_length] on `device`, equal to torch.tensor(batch_encode(texts))."""
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("batch_encode_device packs on the GPU; use batch_encode for host lists")
        rows = [self._lookup(t) for t in texts]
        offs, flat = [0], []
        for r in rows:
            flat.extend(r)
            offs.append(len(flat))
        L = _pkg()._lib
        words = torch.tensor(flat if flat else [0], dtype=torch.int32).to(dev, non_blocking=True)
        offsets = torch.tensor(offs, dtype=torch.int64).to(dev, non_blocking=True)
        B = len(rows)
        ids = torch.empty((B, self.max_length), device=dev, dtype=torch.int64)
        mask = torch.empty((B, self.max_length), device=dev, dtype=torch.int64)
        if B:
            L.call("vqa_pack_tokens", words.data_ptr(), offsets.data_ptr(), ids.data_ptr(), mask.data_ptr(), B, self.max_length,
                   int(add_special_tokens), START_IDX, END_IDX, PAD_IDX)
        return ids, mask

    def decode(self, token_ids: Sequence[int], skip_special_tokens: bool = True) -> str:      # utils/tokenizer.py:252-272
        words = (self.idx2word.get(int(i), UNK_TOKEN) for i in token_ids)
        return " ".join(w for w in words if not (skip_special_tokens and w in SPECIAL_TOKENS))

    def save(self, filepath: str) -> None:                                                    # utils/tokenizer.py:274-290
        with open(filepath, "w", encoding="utf-8") as f:
            json.dump({"word2idx": self.word2idx, "max_length": self.max_length, "max_vocab_size": self.max_vocab_size}, f,
                      indent=2, ensure_ascii=False)

    def load(self, filepath: str) -> None:                                                    # utils/tokenizer.py:292-310
        with open(filepath, "r", encoding="utf-8") as f:
            data = json.load(f)
        self.word2idx = data["word2idx"]
        self.idx2word = {int(v): k for k, v in self.word2idx.items()}
        self.max_length = data.get("max_length", self.max_length)
        self.max_vocab_size = data.get("max_vocab_size", self.max_vocab_size)
        self._is_fitted = True


def create_tokenizer_from_questions(questions: Sequence[str], max_length: int = 20, vocab_size: Optional[int] = 10000, min_freq: int = 2,
                                    save_path: Optional[str] = None) -> Tokenizer:          # utils/tokenizer.py:340-366
    tok = Tokenizer(max_length=max_length, vocab_size=vocab_size)
    tok.build_vocab(questions, min_freq=min_freq)
    if save_path:
        tok.save(save_path)
    return tok
