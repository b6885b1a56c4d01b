"""CPU: pin oracle/vqa_oracle.py against outputs of the real reference (tests/golden/*.npz,
written by tests/golden/make_golden.py).  Tolerances: the oracle and the reference run the same
ATen CPU kernels in a different composition, so agreement is ~1e-6; bounds below are 2e-5."""
import os

import numpy as np
import pytest
import torch

from oracle import vqa_oracle as O

torch.set_num_threads(8)


def _checksum(sd):
    return np.array([[float(sd[k].double().sum()), float(sd[k].double().abs().sum())] for k in sorted(sd)])


def _mask(lens, L=20):
    return (torch.arange(L)[None, :] < torch.tensor(lens)[:, None]).long()


def test_state_dict_layout_counts():
    cfg = O.full_config()
    shapes = O.param_shapes(cfg)
    assert len(shapes) == 225 and len(O.parameter_names(cfg)) == 164
    sd = O.init_state_dict(cfg, 0)
    total = sum(sd[n].numel() for n in O.parameter_names(cfg))
    assert total == 19_310_316


def test_full_eval_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "full_eval.npz"))
    cfg = O.full_config()
    sd = O.init_state_dict(cfg, 1, jitter=True)
    np.testing.assert_allclose(_checksum(sd), g["weight_checksum"], rtol=0, atol=0)
    images, ids, _, _ = O.synthetic_batch(4, seed=11)
    with torch.no_grad():
        logits, aux = O.vqa_forward(images, ids, _mask([20, 15, 7, 5]), sd, cfg, training=False)
    np.testing.assert_allclose(logits.numpy(), g["logits"], atol=2e-5, rtol=0)
    assert (logits.argmax(-1).numpy() == g["logits"].argmax(-1)).all() and g["margin"].min() > 1e-2
    for k in ("fused", "text_pooled", "attended_pooled", "text_features", "image_projected", "image_features"):
        np.testing.assert_allclose(aux[k].numpy(), g[k], atol=2e-5, rtol=0, err_msg=k)
    np.testing.assert_allclose(aux["cross_attention_weights"][0].numpy(), g["cross_w0"], atol=2e-6)
    np.testing.assert_allclose(aux["cross_attention_weights"][1].numpy(), g["cross_w1"], atol=2e-6)


def test_all_padding_row_is_nan_like_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "full_eval_allpad.npz"))
    cfg = O.full_config()
    sd = O.init_state_dict(cfg, 1, jitter=True)
    images, ids, _, _ = O.synthetic_batch(4, seed=11)
    m = _mask([20, 15, 7, 5])
    m[2] = 0
    with torch.no_grad():
        logits, _ = O.vqa_forward(images, ids, m, sd, cfg, training=False)
    assert (np.isnan(logits.numpy()) == np.isnan(g["logits"])).all()
    ok = ~np.isnan(g["logits"])
    np.testing.assert_allclose(logits.numpy()[ok], g["logits"][ok], atol=2e-5)


@pytest.mark.parametrize("tag,cfgkw,seed,B,isz,L,vocab", [
    ("full_train", dict(dropout=0.0, answer_dropout=0.0), 2, 4, 224, 20, 1000),
    ("small_train", dict(dropout=0.0, answer_dropout=0.0, vocab_size=100, num_answers=10, embed_dim=32), 3, 2, 64, 10, 100),
])
def test_train_step_matches_reference(golden_dir, tag, cfgkw, seed, B, isz, L, vocab):
    g = np.load(os.path.join(golden_dir, tag + ".npz"))
    cfg = O.full_config(**cfgkw)
    sd = O.init_state_dict(cfg, seed, jitter=True)
    np.testing.assert_allclose(_checksum(sd), g["weight_checksum"], rtol=0, atol=0)
    images, ids, mask, answers = O.synthetic_batch(B, seed=seed + 100, image_size=isz, seq_len=L, vocab=vocab,
                                                   num_answers=cfg["num_answers"])
    tr = O.OracleTrainer(sd, cfg)
    before = {k: v.detach().clone() for k, v in tr.sd.items()}
    # grads before clipping: replicate step() but read the raw norms first
    nb = {}
    logits, aux = O.vqa_forward(images, ids, mask, tr.sd, cfg, True, nb)
    loss = torch.nn.functional.cross_entropy(logits, answers)
    loss.backward()
    names = O.parameter_names(cfg)
    norms = np.array([float(tr.sd[n].grad.double().norm()) for n in names])
    np.testing.assert_allclose(logits.detach().numpy(), g["logits"], atol=5e-5)
    assert abs(loss.item() - float(g["loss"])) < 1e-5
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=2e-3, atol=1e-6)
    heads = np.stack([np.pad(tr.sd[n].grad.flatten()[:64].numpy(), (0, max(0, 64 - tr.sd[n].numel()))) for n in names])
    scale = np.maximum(np.abs(g["grad_heads"]).max(axis=1, keepdims=True), 1e-6)
    # B=4 train-mode BN makes early-layer grads ill-conditioned: fp32 summation-order noise reaches ~7e-3 of the
    # tensor max at the stem between two CPU compositions of the same math; bound 2e-2.
    assert np.max(np.abs(heads - g["grad_heads"]) / scale) < 2e-2
    gn = torch.nn.utils.clip_grad_norm_(tr.params, 1.0)
    assert abs(float(gn) - float(g["gnorm"])) / float(g["gnorm"]) < 1e-3
    tr.opt.step()
    delta = np.array([float((tr.sd[n].detach() - before[n]).double().norm()) for n in names])
    np.testing.assert_allclose(delta, g["step_delta_norms"], rtol=5e-3, atol=1e-7)
    bn_keys = [k for k in tr.sd if "running_" in k]
    got = np.concatenate([nb[k].numpy() for k in bn_keys])
    np.testing.assert_allclose(got, g["bn_running"], atol=2e-5)
    assert int(nb["image_encoder.stem.1.num_batches_tracked"]) == int(g["nbt"]) == 1


def test_overfit_behaviour_like_reproduce_issue(golden_dir):
    """reproduce_issue.py:16-76 -- the reference reaches acc 1.0; the oracle must overfit the same batch."""
    g = np.load(os.path.join(golden_dir, "overfit.npz"))
    assert float(g["acc"]) > 0.9
    cfg = O.full_config(vocab_size=100, num_answers=10, embed_dim=32)
    sd = O.init_state_dict(cfg, 42)
    gen = torch.Generator().manual_seed(42)
    images = torch.randn(4, 3, 224, 224, generator=gen)
    ids = torch.randint(0, 100, (4, 10), generator=gen)
    mask = torch.ones(4, 10)
    targets = torch.tensor([1] * 4)
    tr = O.OracleTrainer(sd, cfg, lr=1e-3, max_grad_norm=1e9)
    torch.manual_seed(0)
    for _ in range(30):
        loss, logits, _ = tr.step(images, ids, mask, targets)
    assert (logits.argmax(-1) == targets).float().mean().item() > 0.9
