"""GPU: bf16 MFMA attention forward against the (oracle-verified) LDS/VALU kernel and against a plain fp32 torch formula."""
import math

import pytest
import torch

from _pkg import sub

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("case", [(3, 8, 20, 20, 32, True, 0.0), (2, 8, 20, 49, 32, False, 0.0), (2, 4, 17, 64, 64, True, 0.0),
                                  (5, 8, 20, 49, 32, False, 0.1), (1, 2, 32, 33, 32, True, 0.25)])
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
