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


@pytest.mark.parametrize("B,isz,L", [(5, 224, 13), (3, 160, 20), (7, 96, 7), (1, 224, 20), (9, 224, 1)])
def test_train_step_fp32_ragged_shapes_match_oracle(B, isz, L):
    """Odd batch sizes (M-tile tails in every GEMM), sequence lengths other than 20 (down to a single token), one fp32 train step
    (dropout 0) against autograd of the CPU oracle: logits, loss, every parameter tensor's gradient.  Image sizes other than 224 give
    5x5 / 3x3 final maps: the reference then uses the first H*W rows of the position embedding (models/fusion.py:108-110), the rest of
    that parameter receives a zero gradient."""
    cfg = O.full_config(dropout=0.0, answer_dropout=0.0)
    sd = O.init_state_dict(cfg, 40 + B, jitter=True)
    m = _model(cfg, sd, "fp32").train()
    images, ids, mask, answers = O.synthetic_batch(B, seed=300 + B, image_size=isz, seq_len=max(L, 5))
    ids, mask = ids[:, :L].contiguous(), mask[:, :L].contiguous()
    mask[:, 0] = 1                                                  # (an all-padding row is the NaN case of its own test)
    logits, _ = m(images.to(DEV), ids.to(DEV), mask.to(DEV))
    loss = torch.nn.functional.cross_entropy(logits, answers.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    names = O.parameter_names(cfg)
    sdr = {k: (v.clone().requires_grad_(True) if k in set(names) else v.clone()) for k, v in sd.items()}
    logits_ref, _ = O.vqa_forward(images, ids, mask, sdr, cfg, True, {})
    lref = torch.nn.functional.cross_entropy(logits_ref, answers)
    lref.backward()
    assert (logits.detach().cpu() - logits_ref.detach()).abs().max().item() < 1e-3
    assert abs(loss.item() - float(lref.detach())) < 1e-4
    P = dict(m.named_parameters())
    worst = (0.0, None)
    for n in names:
        gh, gr = P[n].grad.detach().cpu().double().flatten(), sdr[n].grad.double().flatten()
        if float(gr.norm()) < 1e-12:
            assert float(gh.norm()) < 1e-9, n
            continue
        rel = float((gh - gr).norm() / gr.norm())
        if rel > worst[0]:
            worst = (rel, n)
    # B = 1: train-mode BatchNorm over 49 positions at stage 4 amplifies fp32 summation-order noise (tests/test_oracle_golden.py)
    assert worst[0] < (0.25 if B == 1 else 5e-2), worst


@pytest.mark.parametrize("B,isz,L", [(5, 224, 13), (3, 160, 20), (9, 224, 1)])
def test_train_step_bf16_ragged_shapes_run_the_fast_kernels(B, isz, L):
    """The same ragged shapes through the bf16 path (8-wave stage-1 kernels at 40x40 maps, window-loader tails, grouped launches with odd
    row counts): loss within 5e-2 of the fp32 oracle, every gradient finite, and the run is bit-reproducible."""
    cfg = O.full_config(dropout=0.0, answer_dropout=0.0)
    sd = O.init_state_dict(cfg, 40 + B, jitter=True)
    images, ids, mask, answers = O.synthetic_batch(B, seed=300 + B, image_size=isz, seq_len=max(L, 5))
    ids, mask = ids[:, :L].contiguous(), mask[:, :L].contiguous()
    mask[:, 0] = 1
    lref = torch.nn.functional.cross_entropy(O.vqa_forward(images, ids, mask, sd, cfg, True, {})[0], answers)
    grads = []
    for rep in range(2):
        m = _model(cfg, sd, "bf16").train()
        logits, _ = m(images.to(DEV), ids.to(DEV), mask.to(DEV))
        loss = torch.nn.functional.cross_entropy(logits.float(), answers.to(DEV))
        loss.backward()
        torch.cuda.synchronize()
        assert abs(loss.item() - float(lref)) < 5e-2
        g = torch.cat([p.grad.flatten() for p in m.parameters()])
        assert torch.isfinite(g).all() and float(g.abs().max()) > 0
        grads.append(g)
    assert torch.equal(grads[0], grads[1])


def test_train_bf16_grads_close_to_oracle():
    """bf16 path, B=8, dropout 0: loss within 2e-2 and EVERY parameter tensor's gradient against the fp32 CPU oracle, held to the
    measured noise floor of the same model under PyTorch's CPU bf16 autocast (bounds and rationale: tests/_bf16check.py; the worst
    tensor is named in the assertion message)."""
    from _bf16check import check_bf16_grads
    cfg = O.full_config(dropout=0.0, answer_dropout=0.0)
    sd = O.init_state_dict(cfg, 7, jitter=True)
    m = _model(cfg, sd, "bf16").train()
    images, ids, mask, answers = O.synthetic_batch(8, seed=77)
    logits, _ = m(images.to(DEV), ids.to(DEV), mask.to(DEV))
    loss = torch.nn.functional.cross_entropy(logits, answers.to(DEV))
    loss.backward()
    noisy = {f"image_encoder.stage{s}.attention.se.fc1.weight" for s in (1, 2, 3)}
    worst, lref = check_bf16_grads(m, sd, cfg, images, ids, mask, answers, noisy)
    assert abs(loss.item() - lref) < 2e-2


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


def test_two_forwards_then_one_backward_like_the_reference():
    """Gradient accumulation / loss1 + loss2 (two training forwards, then ONE backward through both) works in the reference
    (plain autograd); the drop-in keeps a tape per forward, by id.  Checked against autograd of the CPU oracle."""
    cfg = O.full_config(dropout=0.0, answer_dropout=0.0, vocab_size=100, num_answers=10, embed_dim=32)
    sd = O.init_state_dict(cfg, 7, jitter=True)
    m = _model(cfg, sd, "fp32").train()
    b1 = O.synthetic_batch(3, seed=61, image_size=64, seq_len=10, vocab=100, num_answers=10)
    b2 = O.synthetic_batch(2, seed=62, image_size=64, seq_len=10, vocab=100, num_answers=10)
    l1, _ = m(b1[0].to(DEV), b1[1].to(DEV), b1[2].to(DEV))
    l2, _ = m(b2[0].to(DEV), b2[1].to(DEV), b2[2].to(DEV))
    assert len(m._tapes) == 2
    loss = torch.nn.functional.cross_entropy(l1, b1[3].to(DEV)) + torch.nn.functional.cross_entropy(l2, b2[3].to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    assert len(m._tapes) == 0
    # oracle: same two forwards (BatchNorm buffers advance between them exactly as in the model), summed loss, autograd
    pnames = set(O.parameter_names(cfg))
    osd = {k: (v.clone().requires_grad_(True) if k in pnames else v.clone()) for k, v in sd.items()}
    bufs = {}
    o1, _ = O.vqa_forward(b1[0], b1[1], b1[2], osd, cfg, True, bufs)
    o2, _ = O.vqa_forward(b2[0], b2[1], b2[2], osd, cfg, True, bufs)
    lref = torch.nn.functional.cross_entropy(o1, b1[3]) + torch.nn.functional.cross_entropy(o2, b2[3])
    lref.backward()
    assert abs(loss.item() - lref.item()) < 1e-4
    worst = 0.0
    for name, p in m.named_parameters():
        gr = osd[name].grad
        if gr is None:
            continue
        den = float(gr.norm()) + 1e-8
        worst = max(worst, float((p.grad.cpu() - gr).norm()) / den if den > 1e-6 else 0.0)
    assert worst < 5e-2, worst
    # a tape evicted before its backward raises a clear error, it does not KeyError
    m.max_live_tapes = 1
    la, _ = m(b1[0].to(DEV), b1[1].to(DEV), b1[2].to(DEV))
    lb, _ = m(b2[0].to(DEV), b2[1].to(DEV), b2[2].to(DEV))
    with pytest.raises(RuntimeError, match="max_live_tapes"):
        (la.sum() + lb.sum()).backward()


def test_attention_maps_follow_num_image_tokens():
    """get_attention_maps at the 144-token extension (BASELINE configs[4] geometry): output_spatial_size is derived (12), not 7."""
    cfg = O.full_config(dropout=0.0, answer_dropout=0.0, vocab_size=100, num_answers=10, embed_dim=64)
    m = pkg().load_dropin().VQAModel(**cfg, compute_dtype="fp32", num_image_tokens=144).to(DEV).eval()
    assert m.image_encoder.output_spatial_size == 12
    images, ids, mask, _ = O.synthetic_batch(2, seed=5, image_size=384, seq_len=10, vocab=100, num_answers=10)
    with torch.no_grad():
        maps = m.get_attention_maps(images.to(DEV), ids.to(DEV), mask.to(DEV))
    assert maps["cross_attention_spatial"].shape == (2, 10, 12, 12)


def test_train_bf16_gradients_at_batch64_against_the_fp32_oracle():
    """Whole-model bf16 gradients at B=64 against autograd of the fp32 CPU oracle, every parameter tensor bounded:
        token side (text encoder, fusion, answer head -- 120 tensors): relative L2 error <= 0.25   (measured <= 0.163)
        CNN tensors: <= 0.75, whole-model vector <= 0.45                                       (measured <= 0.533 / 0.32)
    The CNN figure is the noise floor of bf16 ITSELF on this model at random init (PyTorch's own CPU bf16 autocast sits at 0.4-0.55,
    tests/_bf16check.py) and does not shrink with the batch (B = 8 and B = 64 measure the same: the batch gradient is the small
    residual of per-sample gradients that cancel) -- so the CNN path is held to 4e-3 per LAYER by tests/test_gpu_insitu.py on the
    live tape of this very configuration instead, and this test keeps the end-to-end view."""
    torch.set_num_threads(16)
    B = 64
    cfg = O.full_config(dropout=0.0, answer_dropout=0.0)
    sd = O.init_state_dict(cfg, 7, jitter=True)
    images, ids, mask, answers = O.synthetic_batch(B, seed=77)
    tr = O.OracleTrainer(sd, cfg)
    lo, _ = O.vqa_forward(images, ids, mask, tr.sd, cfg, True, {})
    lref = torch.nn.functional.cross_entropy(lo, answers)
    lref.backward()
    m = _model(cfg, sd, "bf16").train()
    logits, _ = m(images.to(DEV), ids.to(DEV), mask.to(DEV))
    loss = torch.nn.functional.cross_entropy(logits, answers.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - lref.item()) < 2e-2 and (logits.cpu() - lo.detach()).abs().max().item() < 5e-2
    P = dict(m.named_parameters())
    names = O.parameter_names(cfg)
    worst_tok, worst_cnn = ("", 0.0), ("", 0.0)
    gs, rs = [], []
    for n in names:
        r = tr.sd[n].grad.float().reshape(-1)
        g = P[n].grad.detach().float().cpu().reshape(-1)
        gs.append(g); rs.append(r)
        if float(r.norm()) < 1e-10:
            assert float(g.norm()) < 1e-6, n
            continue
        e = float((g - r).norm() / r.norm())
        if n.endswith(".spatial.conv.weight"):
            # 98 weights fed by a channel-max / channel-mean map: PyTorch's own bf16 autocast sits at e ~ 1.2 on this tensor
            # (tests/_bf16check.py), and its value moves between 0.4 and 0.9 with last-bit changes upstream -- bounded loosely here,
            # its kernel is pinned by the fp32 goldens and tests/test_gpu_variants.py
            assert e <= 1.5, (n, e)
        elif n.startswith("image_encoder."):
            worst_cnn = max(worst_cnn, (n, e), key=lambda t: t[1])
            assert e <= 0.75, (n, e)
        else:
            worst_tok = max(worst_tok, (n, e), key=lambda t: t[1])
            assert e <= 0.25, (n, e)
    G, R = torch.cat(gs), torch.cat(rs)
    whole = float((G - R).norm() / R.norm())
    assert whole <= 0.45, whole
    print("worst token-side", worst_tok, "worst CNN", worst_cnn, "whole model", whole)
