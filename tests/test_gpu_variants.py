"""GPU: ablation / shape variants of the drop-in against the CPU oracle (fp32 path, train step with dropout 0):
no SE, no spatial attention, no gating, fewer layers, attention_mask=None, odd batch, non-224 images, short questions,
and eval-mode backward (BatchNorm on running statistics)."""
import numpy as np
import pytest
import torch

from _pkg import pkg
from oracle import vqa_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"

VARIANTS = [
    ("no_attention", dict(use_se_attention=False, use_spatial_attention=False), 3, 64, 12, True, True),
    ("se_only", dict(use_spatial_attention=False), 2, 96, 20, True, True),
    ("no_gating_1cross_2text", dict(use_gating=False, num_cross_layers=1, num_transformer_layers=2), 3, 64, 9, True, True),
    ("mask_none", dict(), 2, 64, 20, False, True),
    ("eval_mode_backward", dict(), 2, 64, 14, True, False),
    ("wide_head_dim64", dict(embed_dim=512, num_attention_heads=8, num_answers=24), 2, 64, 20, True, True),
]


@pytest.mark.parametrize("tag,kw,B,isz,L,use_mask,training", VARIANTS)
def test_variant_train_step_matches_oracle(tag, kw, B, isz, L, use_mask, training):
    base = dict(dropout=0.0, answer_dropout=0.0, vocab_size=300, num_answers=40, embed_dim=64)
    base.update(kw)
    cfg = O.full_config(**base)
    sd = O.init_state_dict(cfg, 31, jitter=True)
    images, ids, mask, answers = O.synthetic_batch(B, seed=41, image_size=isz, seq_len=L, vocab=300, num_answers=cfg["num_answers"])
    if not use_mask:
        mask = None
    m = pkg().load_dropin().VQAModel(**cfg, compute_dtype="fp32")
    m.load_state_dict(sd)
    m = m.to(DEV)
    m.train(training)
    logits, _ = m(images.to(DEV), ids.to(DEV), None if mask is None else mask.to(DEV))
    loss = torch.nn.functional.cross_entropy(logits, answers.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    tr = O.OracleTrainer(sd, cfg)
    lref, _ = O.vqa_forward(images, ids, mask, tr.sd, cfg, training, {})
    loss_ref = torch.nn.functional.cross_entropy(lref, answers)
    loss_ref.backward()
    assert (logits.detach().cpu() - lref.detach()).abs().max().item() < 1e-3, tag
    assert abs(loss.item() - loss_ref.item()) < 1e-4
    P = dict(m.named_parameters())
    names = O.parameter_names(cfg)
    got = np.array([float(P[n].grad.double().norm()) for n in names])
    ref = np.array([float(tr.sd[n].grad.double().norm()) for n in names])
    rel = np.abs(got - ref) / np.maximum(ref, 1e-6 * ref.max())
    assert rel.max() < 3e-2, (tag, names[int(rel.argmax())], rel.max())
