"""GPU: device-side accuracy counters (vqa_accuracy_update + the utils.metrics.VQAAccuracy drop-in) against the reference's
golden counters and the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from _pkg import pkg, sub
from oracle import metrics_oracle as MO

pytestmark = pytest.mark.gpu
DEV = "cuda"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "metrics.npz")


def test_accuracy_dropin_matches_reference_golden():
    M = pkg().load_dropin_metrics()
    g = np.load(GOLD)
    acc = M.VQAAccuracy()
    for i in range(g["running"].shape[0]):
        acc.update(torch.from_numpy(g[f"logits{i}"]).to(DEV), torch.from_numpy(g[f"targets{i}"]).to(DEV))
        assert [acc.correct, acc.correct_top5, acc.total] == g["running"][i].tolist()
    m = acc.compute()
    assert abs(m["accuracy"] - g["accuracy"][0]) < 1e-12 and abs(m["accuracy_top5"] - g["accuracy"][1]) < 1e-12
    assert str(acc).startswith("Accuracy: ")
    acc.reset()
    assert acc.compute()["total"] == 0


@pytest.mark.parametrize("shape", [(512, 1000), (3, 5), (130, 64), (1, 1)])
def test_accuracy_kernel_matches_oracle_with_ties_and_invalid_targets(shape):
    L = sub("_lib")
    B, C = shape
    gen = torch.Generator().manual_seed(B * 7 + C)
    logits = torch.randint(-3, 4, (B, C), generator=gen).float()        # small integers: plenty of exact ties
    targets = torch.randint(0, C, (B,), generator=gen)
    if B > 2:
        targets[1] = C + 3                                              # out of range: counted in total only
        targets[2] = -1
    counters = torch.zeros(3, dtype=torch.int64, device=DEV)
    lg, tg = logits.to(DEV), targets.to(DEV)
    for _ in range(2):                                                  # accumulates (+=)
        L.call("vqa_accuracy_update", lg.data_ptr(), tg.data_ptr(), counters.data_ptr(), B, C)
    torch.cuda.synchronize()
    c, c5, n = MO.accuracy_counts(logits.numpy(), targets.numpy())
    assert counters.cpu().tolist() == [2 * c, 2 * c5, 2 * n]


def test_accuracy_dropin_has_no_cpu_path_and_index_predictions():
    M = pkg().load_dropin_metrics()
    acc = M.VQAAccuracy()
    with pytest.raises(RuntimeError):
        acc.update(torch.zeros(2, 3), torch.zeros(2, dtype=torch.long))
    acc.update(torch.tensor([1, 2, 3], device=DEV), torch.tensor([1, 0, 3], device=DEV), question_types=["a", "b", "a"])
    m = acc.compute()
    assert m["correct"] == 2 and m["total"] == 3 and m["per_type"] == {"a": 1.0, "b": 0.0}


def test_trainer_step_updates_metrics_without_changing_the_step():
    P = pkg()
    M = P.load_dropin()
    torch.manual_seed(0)
    cfg = dict(vocab_size=100, num_answers=10, embed_dim=32, dropout=0.0, answer_dropout=0.0)
    model = M.VQAModel(**cfg, compute_dtype="fp32", seed=5).to(DEV).train()
    tr = P.trainer.HipTrainer(model)
    acc = P.load_dropin_metrics().VQAAccuracy()
    g = torch.Generator().manual_seed(3)
    images = torch.randn(4, 3, 64, 64, generator=g).to(DEV)
    ids = torch.randint(1, 100, (4, 10), generator=g).to(DEV)
    mask = torch.ones(4, 10, dtype=torch.long, device=DEV)
    ans = torch.randint(0, 10, (4,), generator=g).to(DEV)
    _, logits = tr.step(images, ids, mask, ans, metrics=acc)
    c, c5, n = MO.accuracy_counts(logits.float().cpu().numpy(), ans.cpu().numpy())
    assert [acc.correct, acc.correct_top5, acc.total] == [c, c5, n]
