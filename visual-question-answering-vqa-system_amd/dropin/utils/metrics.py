"""`utils.metrics.VQAAccuracy` drop-in with device-resident counters (reference: utils/metrics.py:29-135).

The reference's `update` moves argmax / top-5 indices to the host and calls `.item()` on every batch
(utils/metrics.py:80-94): two device syncs per train step.  Here `update` launches one HIP kernel
(`vqa_accuracy_update`, include/vqa_hip.h) that adds {top-1 correct, top-5 correct, samples} into a 3-element
u64 device buffer; nothing is read back until `compute()`.  Same interface: reset / update / compute / __str__.
There is no CPU path: logits on the host raise.
"""
from __future__ import annotations

import importlib
import os
from typing import Dict, List, Optional

import torch


def _pkg():
    import sys
    here = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    root = os.path.dirname(here)
    if root not in sys.path:
        sys.path.insert(0, root)
    return importlib.import_module(os.path.basename(here))


class VQAAccuracy:
    """Running top-1 / top-5 accuracy (utils/metrics.py:29-135), counters kept on the GPU."""

    def __init__(self):
        self._L = _pkg()._lib
        self._counters: Optional[torch.Tensor] = None      # int64 view of {correct, correct_top5, total}
        self.reset()

    def reset(self):                                          # utils/metrics.py:47-53
        if self._counters is not None:
            self._counters.zero_()
        self.per_type_correct: Dict[str, int] = {}
        self.per_type_total: Dict[str, int] = {}

    def _buf(self, device) -> torch.Tensor:
        if self._counters is None or self._counters.device != device:
            self._counters = torch.zeros(3, dtype=torch.int64, device=device)
        return self._counters

    def update(self, predictions: torch.Tensor, targets: torch.Tensor, question_types: Optional[List[str]] = None):
        """predictions: logits [B, C] (fp32, GPU) or indices [B]; targets: i64 [B] (utils/metrics.py:55-106)."""
        if not predictions.is_cuda:
            raise RuntimeError("VQAAccuracy (HIP) keeps its counters on the GPU: predictions must be a GPU tensor; there is no CPU path")
        targets = targets.to(predictions.device, torch.int64).contiguous()
        c = self._buf(predictions.device)
        if predictions.dim() == 2:
            lg = predictions.detach()
            lg = lg.float().contiguous() if lg.dtype != torch.float32 else lg.contiguous()
            self._L.call("vqa_accuracy_update", lg.data_ptr(), targets.data_ptr(), c.data_ptr(), lg.shape[0], lg.shape[1])
            mask = None
        else:                                                  # index predictions: top-1 only, still no host sync
            mask = predictions.to(torch.int64) == targets
            c[0] += mask.sum()
            c[2] += targets.shape[0]
        if question_types is not None:                         # per-type breakdown needs the per-sample mask on the host (one copy)
            if mask is None:
                mask = predictions.argmax(dim=-1) == targets
            mask = mask.cpu()
            for i, qtype in enumerate(question_types):
                self.per_type_total[qtype] = self.per_type_total.get(qtype, 0) + 1
                self.per_type_correct[qtype] = self.per_type_correct.get(qtype, 0) + int(mask[i])

    # the reference exposes these as plain attributes
    def _read(self):
        return [0, 0, 0] if self._counters is None else [int(v) for v in self._counters.cpu()]

    @property
    def correct(self) -> int:
        return self._read()[0]

    @property
    def correct_top5(self) -> int:
        return self._read()[1]

    @property
    def total(self) -> int:
        return self._read()[2]

    def compute(self) -> Dict[str, float]:                    # utils/metrics.py:108-130
        correct, top5, total = self._read()
        results = {"accuracy": correct / max(total, 1), "accuracy_top5": top5 / max(total, 1), "correct": correct, "total": total}
        if self.per_type_total:
            results["per_type"] = {q: self.per_type_correct[q] / max(self.per_type_total[q], 1) for q in self.per_type_total}
        return results

    def __str__(self) -> str:
        m = self.compute()
        return f"Accuracy: {m['accuracy']:.4f} | Top-5: {m['accuracy_top5']:.4f}"
