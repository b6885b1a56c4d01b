"""GPU parity of the whole drop-in VQAModel (HIP kernels through the C ABI) against
(a) the golden outputs of the real reference (tests/golden/*.npz) and (b) the CPU oracle on the same seeded inputs.
fp32 bar (north_star): logits within 1e-3, argmax identical.  bf16 path: looser bound stated per test."""
import os

import numpy as np
import pytest
import torch

from _pkg import pkg
from oracle import vqa_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _mask(lens, L=20):
    return (torch.arange(L)[None, :] < torch.tensor(lens)[:, None]).long()


def _model(cfg, sd, dtype):
    M = pkg().load_dropin()
    m = M.VQAModel(**cfg, compute_dtype=dtype)
    m.load_state_dict(sd)
    return m.to(DEV)


def test_eval_fp32_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "full_eval.npz"))
    cfg = O.full_config()
    sd = O.init_state_dict(cfg, 1, jitter=True)
    m = _model(cfg, sd, "fp32").eval()
    images, ids, _, _ = O.synthetic_batch(4, seed=11)
    mask = _mask([20, 15, 7, 5])
    with torch.no_grad():
        logits, aux = m(images.to(DEV), ids.to(DEV), mask.to(DEV), return_aux=True)
    torch.cuda.synchronize()
    lg = logits.cpu().numpy()
    assert np.abs(lg - g["logits"]).max() < 1e-3
    assert (lg.argmax(-1) == g["logits"].argmax(-1)).all()
    for k in ("fused", "text_pooled", "attended_pooled", "text_features", "image_projected", "image_features"):
        ref = g[k]
        assert np.abs(aux[k].cpu().numpy() - ref).max() < 1e-3 * max(1.0, np.abs(ref).max()), k
    assert np.abs(aux["cross_attention_weights"][1].cpu().numpy() - g["cross_w1"]).max() < 1e-4
    # predict() / get_attention_maps() surface
    idx, pr = m.predict(images.to(DEV), ids.to(DEV), mask.to(DEV), top_k=5)
    assert idx.shape == (4, 5) and (idx[:, 0].cpu().numpy() == g["logits"].argmax(-1)).all()
    maps = m.get_attention_maps(images.to(DEV), ids.to(DEV), mask.to(DEV))
    assert maps["cross_attention_spatial"].shape == (4, 20, 7, 7)


def test_eval_allpad_row_nan_like_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "full_eval_allpad.npz"))
    cfg = O.full_config()
    sd = O.init_state_dict(cfg, 1, jitter=True)
    m = _model(cfg, sd, "fp32").eval()
    images, ids, _, _ = O.synthetic_batch(4, seed=11)
    mask = _mask([20, 15, 7, 5]); mask[2] = 0
    with torch.no_grad():
        logits, _ = m(images.to(DEV), ids.to(DEV), mask.to(DEV))
    lg = logits.cpu().numpy()
    assert (np.isnan(lg) == np.isnan(g["logits"])).all()
    ok = ~np.isnan(g["logits"])
    assert np.abs(lg[ok] - g["logits"][ok]).max() < 1e-3


def test_eval_bf16_close_to_reference(golden_dir):
    """bf16 storage/MFMA path: SURVEY measured ~8e-3 drift for CPU bf16 autocast; bound 5e-2 abs on logits (|logit| ~ 0.5)."""
    g = np.load(os.path.join(golden_dir, "full_eval.npz"))
    cfg = O.full_config()
    sd = O.init_state_dict(cfg, 1, jitter=True)
    m = _model(cfg, sd, "bf16").eval()
    images, ids, _, _ = O.synthetic_batch(4, seed=11)
    with torch.no_grad():
        logits, _ = m(images.to(DEV), ids.to(DEV), _mask([20, 15, 7, 5]).to(DEV))
    assert np.abs(logits.cpu().numpy() - g["logits"]).max() < 5e-2


@pytest.mark.parametrize("tag,cfgkw,seed,B,isz,L,vocab", [
    ("full_train", dict(dropout=0.0, answer_dropout=0.0), 2, 4, 224, 20, 1000),
    ("small_train", dict(dropout=0.0, answer_dropout=0.0, vocab_size=100, num_answers=10, embed_dim=32), 3, 2, 64, 10, 100),
])
def test_train_step_fp32_matches_reference_golden(golden_dir, tag, cfgkw, seed, B, isz, L, vocab):
    g = np.load(os.path.join(golden_dir, tag + ".npz"))
    cfg = O.full_config(**cfgkw)
    sd = O.init_state_dict(cfg, seed, jitter=True)
    m = _model(cfg, sd, "fp32").train()
    images, ids, mask, answers = O.synthetic_batch(B, seed=seed + 100, image_size=isz, seq_len=L, vocab=vocab,
                                                   num_answers=cfg["num_answers"])
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=0.01, betas=(0.9, 0.999))
    opt.zero_grad()
    logits, _ = m(images.to(DEV), ids.to(DEV), mask.to(DEV))
    loss = torch.nn.functional.cross_entropy(logits, answers.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    assert np.abs(logits.detach().cpu().numpy() - g["logits"]).max() < 1e-3
    assert abs(loss.item() - float(g["loss"])) < 1e-4
    names = O.parameter_names(cfg)
    P = dict(m.named_parameters())
    norms = np.array([float(P[n].grad.double().norm()) for n in names])
    rel = np.abs(norms - g["grad_norms"]) / np.maximum(g["grad_norms"], 1e-6)
    # per-tensor gradient norms: fp32 summation-order noise through train-mode BN at B<=4 (see test_oracle_golden) -> 2e-2
    assert rel.max() < 2e-2, (names[int(rel.argmax())], rel.max())
    heads = np.stack([np.pad(P[n].grad.flatten()[:64].cpu().numpy(), (0, max(0, 64 - P[n].numel()))) for n in names])
    # NB: conv weights are channels_last, so flatten() of the logical OIHW tensor still walks OIHW order like the reference
    scale = np.maximum(np.abs(g["grad_heads"]).max(axis=1, keepdims=True), 1e-6)
    assert np.max(np.abs(heads - g["grad_heads"]) / scale) < 5e-2
    gn = torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
    assert abs(float(gn) - float(g["gnorm"])) / float(g["gnorm"]) < 5e-3
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    opt.step()
    delta = np.array([float((P[n].detach() - before[n]).double().norm()) for n in names])
    np.testing.assert_allclose(delta, g["step_delta_norms"], rtol=2e-2, atol=1e-7)
    st = m.state_dict()
    bn_keys = [k for k in st if "running_" in k]
    got = np.concatenate([st[k].cpu().numpy() for k in bn_keys])
    assert np.abs(got - g["bn_running"]).max() < 1e-4
    assert int(st["image_encoder.stem.1.num_batches_tracked"]) == 1


def test_train_bf16_grads_close_to_oracle():
    """bf16 path, B=8, dropout 0: loss and EVERY parameter tensor's gradient against the fp32 CPU oracle.

    What bf16 itself costs is measured, not guessed: the same oracle under PyTorch's CPU bf16 autocast gives per-tensor
    relative errors e = |g - g_fp32| / |g_fp32| of 0.4-0.55 for the CNN weights at this batch size (cosine 0.85-0.93; gradients
    at random init are sums with heavy cancellation, and train-mode BN at B=8 amplifies 8-bit rounding), while the HIP fp32
    path sits at cosine 1.0000 on every tensor (tools/diag_bf16_grads.py).  So the bound is self-calibrating:
        every tensor:  e_hip_bf16 <= 1.25 * e_cpu_autocast_bf16 + 0.10        (not noisier than torch's own bf16, + margin)
        and in any case e_hip_bf16 <= 0.75 for weights with >= 2 dims          (a wrong tile / halo mask / permutation gives e >= 1)
    except the squeeze-excitation fc1 weights of stages 1-3 (|g| ~ 1e-2, a 4..16 x C matrix fed by a global average: both bf16
    implementations decorrelate there, the autocast run even flips sign at B=8): norm within a factor 2.5 only.
    The worst tensor is named in the assertion message."""
    cfg = O.full_config(dropout=0.0, answer_dropout=0.0)
    sd = O.init_state_dict(cfg, 7, jitter=True)
    m = _model(cfg, sd, "bf16").train()
    images, ids, mask, answers = O.synthetic_batch(8, seed=77)
    logits, _ = m(images.to(DEV), ids.to(DEV), mask.to(DEV))
    loss = torch.nn.functional.cross_entropy(logits, answers.to(DEV))
    loss.backward()
    names = O.parameter_names(cfg)

    def oracle(autocast):
        tr = O.OracleTrainer(sd, cfg)
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
            lo, _ = O.vqa_forward(images, ids, mask, tr.sd, cfg, True, {})
            l = torch.nn.functional.cross_entropy(lo.float(), answers)
        l.backward()
        return {n: tr.sd[n].grad.float().reshape(-1) for n in names}, float(l)

    ref, lref = oracle(False)
    acb, _ = oracle(True)
    assert abs(loss.item() - lref) < 2e-2
    P = dict(m.named_parameters())
    noisy = {f"image_encoder.stage{s}.attention.se.fc1.weight" for s in (1, 2, 3)}
    rows = []
    for n in names:
        g, r, a = P[n].grad.detach().float().cpu().reshape(-1), ref[n], acb[n]
        rn = float(r.norm())
        if rn < 1e-10:
            assert float(g.norm()) < 1e-6, n
            continue
        rows.append((n, float((g - r).norm()) / rn, float((a - r).norm()) / rn, float(g.norm()) / rn, P[n].dim()))
    worst = max((t for t in rows if t[0] not in noisy), key=lambda t: t[1] - 1.25 * t[2])
    for n, e_hip, e_acb, ratio, dim in rows:
        if n in noisy:
            assert 0.4 < ratio < 2.5, (n, ratio)
            continue
        assert e_hip <= 1.25 * e_acb + 0.10, (n, e_hip, e_acb, "worst", worst)
        if dim >= 2:
            assert e_hip <= 0.75, (n, e_hip, "worst", worst)
    # whole-model gradient vector
    G, R, A = (torch.cat([d[n] for n in names]) for d in ({n: P[n].grad.detach().float().cpu().reshape(-1) for n in names}, ref, acb))
    assert float((G - R).norm() / R.norm()) <= 1.15 * float((A - R).norm() / R.norm()) + 0.02


def test_train_dropout_runs_and_is_deterministic_per_seed():
    """Model-level smoke of the dropout path (the per-site contract lives in tests/test_gpu_dropout.py): two training forwards
    of the same batch differ (fresh masks per forward), gradients are finite, replaying the step counter replays the masks."""
    cfg = O.full_config()
    sd = O.init_state_dict(cfg, 5)
    m = _model(cfg, sd, "bf16").train()
    images, ids, mask, answers = O.synthetic_batch(4, seed=9)
    a, _ = m(images.to(DEV), ids.to(DEV), mask.to(DEV))
    b, _ = m(images.to(DEV), ids.to(DEV), mask.to(DEV))
    torch.cuda.synchronize()
    assert torch.isfinite(a).all() and torch.isfinite(b).all()
    assert not torch.equal(a, b)
    m._engine.step_id -= 2                               # replay the first forward's seeds
    c, _ = m(images.to(DEV), ids.to(DEV), mask.to(DEV))
    assert torch.equal(a, c)
    torch.nn.functional.cross_entropy(c, answers.to(DEV)).backward()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters())


def test_cpu_inputs_fail_loudly():
    cfg = O.full_config()
    m = pkg().load_dropin().VQAModel(**cfg)
    images, ids, mask, _ = O.synthetic_batch(1, seed=1)
    with pytest.raises(RuntimeError):
        m(images, ids, mask)
