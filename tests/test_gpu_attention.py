"""GPU: bf16 MFMA attention forward against the (oracle-verified) LDS/VALU kernel and against a plain fp32 torch formula."""
import math

import pytest
import torch

from _pkg import sub

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("case", [(3, 8, 20, 20, 32, True, 0.0), (2, 8, 20, 49, 32, False, 0.0), (2, 4, 17, 64, 64, True, 0.0),
                                  (5, 8, 20, 49, 32, False, 0.1), (1, 2, 32, 33, 32, True, 0.25),
                                  # five key tiles: the 144 image tokens of the 384x384 stress shape (d=512 -> head dim 64), tails, masks
                                  (3, 8, 20, 144, 64, False, 0.0), (2, 8, 20, 144, 64, False, 0.1), (2, 4, 32, 160, 32, True, 0.0),
                                  (3, 2, 7, 65, 32, True, 0.2), (1, 8, 20, 129, 64, True, 0.0)])
def test_mfma_attention_forward(case):
    L = sub("_lib")
    B, H, Lq, Lk, hd, masked, p = case
    d = H * hd
    g = torch.Generator().manual_seed(B * 100 + Lk)
    q = torch.randn(B * Lq, d, generator=g).to(DEV, torch.bfloat16)
    k = torch.randn(B * Lk, d, generator=g).to(DEV, torch.bfloat16)
    v = torch.randn(B * Lk, d, generator=g).to(DEV, torch.bfloat16)
    kmask = None
    if masked:
        lens = torch.randint(1, Lk + 1, (B,), generator=g)
        kmask = (torch.arange(Lk)[None, :] < lens[:, None]).float().to(DEV)
    outs = []
    for name in ("vqa_attention_fwd_mfma", "vqa_attention_fwd"):
        probs = torch.empty(B, H, Lq, Lk, device=DEV)
        ctx = torch.empty(B * Lq, d, device=DEV, dtype=torch.bfloat16)
        args = (q.data_ptr(), k.data_ptr(), v.data_ptr(), d, d, d, None if kmask is None else kmask.data_ptr(), probs.data_ptr(),
                ctx.data_ptr(), d, B, H, Lq, Lk, hd, p, 1234)
        if name == "vqa_attention_fwd":
            args = (1,) + args
        L.call(name, *args)
        outs.append((probs, ctx))
    torch.cuda.synchronize()
    (p1, c1), (p2, c2) = outs
    assert (p1 - p2).abs().max().item() < 2e-5                      # same bf16 inputs, fp32 accumulate: only summation order differs
    assert (c1.float() - c2.float()).abs().max().item() < 3e-2       # MFMA path rounds P to bf16 before PV
    # independent fp32 formula (no dropout case)
    if p == 0.0:
        qf = q.float().view(B, Lq, H, hd).transpose(1, 2)
        kf = k.float().view(B, Lk, H, hd).transpose(1, 2)
        vf = v.float().view(B, Lk, H, hd).transpose(1, 2)
        s = qf @ kf.transpose(-1, -2) / math.sqrt(hd)
        if kmask is not None:
            s = s.masked_fill(kmask[:, None, None, :] == 0, float("-inf"))
        pr = torch.softmax(s, -1)
        ref = (pr @ vf).transpose(1, 2).reshape(B * Lq, d)
        assert (p1 - pr).abs().max().item() < 2e-5
        assert (c1.float() - ref).abs().max().item() < 3e-2


def test_mfma_attention_all_masked_row_is_nan():
    L = sub("_lib")
    B, H, Lq, Lk, hd = 2, 8, 20, 20, 32
    d = H * hd
    q = torch.randn(B * Lq, d).to(DEV, torch.bfloat16); k = torch.randn(B * Lk, d).to(DEV, torch.bfloat16); v = torch.randn(B * Lk, d).to(DEV, torch.bfloat16)
    kmask = torch.ones(B, Lk, device=DEV); kmask[1] = 0
    probs = torch.empty(B, H, Lq, Lk, device=DEV); ctx = torch.empty(B * Lq, d, device=DEV, dtype=torch.bfloat16)
    L.call("vqa_attention_fwd_mfma", q.data_ptr(), k.data_ptr(), v.data_ptr(), d, d, d, kmask.data_ptr(), probs.data_ptr(), ctx.data_ptr(), d,
           B, H, Lq, Lk, hd, 0.0, 0)
    torch.cuda.synchronize()
    assert torch.isfinite(ctx[:Lq].float()).all() and torch.isnan(ctx[Lq:].float()).all() and torch.isnan(probs[1]).all()


@pytest.mark.parametrize("case", [(3, 8, 20, 20, 32, True, 0.0), (2, 8, 20, 49, 32, False, 0.0), (2, 4, 17, 64, 64, True, 0.0),
                                  (5, 8, 20, 49, 32, False, 0.1), (1, 2, 32, 33, 32, True, 0.25), (2, 2, 20, 20, 64, False, 0.1),
                                  (3, 8, 20, 144, 64, False, 0.0), (2, 8, 20, 144, 64, False, 0.1), (2, 4, 32, 160, 32, True, 0.0),
                                  (3, 2, 7, 65, 32, True, 0.2), (1, 8, 20, 129, 64, True, 0.0)])
def test_mfma_attention_backward(case):
    """bf16 MFMA attention backward vs the LDS/VALU kernel (same inputs, same saved probabilities, same dropout mask) and,
    without dropout, vs torch autograd of the fp32 formula."""
    L = sub("_lib")
    B, H, Lq, Lk, hd, masked, p = case
    d = H * hd
    g = torch.Generator().manual_seed(B * 100 + Lk + 7)
    q = torch.randn(B * Lq, d, generator=g).to(DEV, torch.bfloat16)
    k = torch.randn(B * Lk, d, generator=g).to(DEV, torch.bfloat16)
    v = torch.randn(B * Lk, d, generator=g).to(DEV, torch.bfloat16)
    dctx = torch.randn(B * Lq, d, generator=g).to(DEV, torch.bfloat16)
    kmask = None
    if masked:
        lens = torch.randint(1, Lk + 1, (B,), generator=g)
        kmask = (torch.arange(Lk)[None, :] < lens[:, None]).float().to(DEV)
    probs = torch.empty(B, H, Lq, Lk, device=DEV)
    ctx = torch.empty(B * Lq, d, device=DEV, dtype=torch.bfloat16)
    L.call("vqa_attention_fwd", 1, q.data_ptr(), k.data_ptr(), v.data_ptr(), d, d, d, None if kmask is None else kmask.data_ptr(),
           probs.data_ptr(), ctx.data_ptr(), d, B, H, Lq, Lk, hd, p, 99)
    outs = []
    for name in ("vqa_attention_bwd_mfma", "vqa_attention_bwd"):
        dq = torch.full((B * Lq, d), 7.0, device=DEV, dtype=torch.bfloat16)
        dk = torch.full((B * Lk, d), 7.0, device=DEV, dtype=torch.bfloat16)
        dv = torch.full((B * Lk, d), 7.0, device=DEV, dtype=torch.bfloat16)
        args = (dctx.data_ptr(), d, q.data_ptr(), k.data_ptr(), v.data_ptr(), d, d, d, probs.data_ptr(), dq.data_ptr(), dk.data_ptr(),
                dv.data_ptr(), d, d, d, B, H, Lq, Lk, hd, p, 99)
        if name == "vqa_attention_bwd":
            args = (1,) + args
        L.call(name, *args)
        outs.append((dq.float(), dk.float(), dv.float()))
    torch.cuda.synchronize()
    for a, b_, nm in zip(outs[0], outs[1], ("dq", "dk", "dv")):
        scale = b_.abs().max().item() + 1e-6
        assert (a - b_).abs().max().item() / scale < 2.5e-2, nm       # operands of the second MFMA are rounded to bf16
    if p == 0.0:
        qf = q.float().view(B, Lq, H, hd).transpose(1, 2).requires_grad_(True)
        kf = k.float().view(B, Lk, H, hd).transpose(1, 2).requires_grad_(True)
        vf = v.float().view(B, Lk, H, hd).transpose(1, 2).requires_grad_(True)
        s = qf @ kf.transpose(-1, -2) / math.sqrt(hd)
        if kmask is not None:
            s = s.masked_fill(kmask[:, None, None, :] == 0, float("-inf"))
        o = (torch.softmax(s, -1) @ vf).transpose(1, 2).reshape(B * Lq, d)
        o.backward(dctx.float())
        for a, ref, nm in zip(outs[0], (qf.grad, kf.grad, vf.grad), ("dq", "dk", "dv")):
            L_ = Lq if nm == "dq" else Lk
            refm = ref.transpose(1, 2).reshape(B * L_, d)
            assert (a - refm).abs().max().item() / (refm.abs().max().item() + 1e-6) < 2.5e-2, nm
