"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's accuracy bookkeeping.

Restates `VQAAccuracy.update` / `.compute` (utils/metrics.py:55-130): top-1 = argmax == target, top-5 = target among
topk(5).  Pinned by tests/golden/metrics.npz, produced by the real reference class (tests/golden/make_golden.py).
Only tests/ may import this module.
"""
import numpy as np


def accuracy_counts(logits: np.ndarray, targets: np.ndarray):
    """(correct, correct_top5, total) for fp32 logits [B, C], i64 targets [B] (utils/metrics.py:70-94).
    Rank of the target among the logits with ties resolved to the lowest index (argmax / stable descending sort)."""
    B, C = logits.shape
    correct = top5 = 0
    for b in range(B):
        t = int(targets[b])
        if not (0 <= t < C):
            continue
        xt = logits[b, t]
        rank = int((logits[b] > xt).sum()) + int((logits[b, :t] == xt).sum())
        correct += rank == 0
        top5 += rank < 5
    return correct, top5, B
