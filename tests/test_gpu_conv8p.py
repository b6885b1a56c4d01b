"""The 8-phase GEMM core (csrc/gemm8p.hip): dense GEMM against fp32 matmul; the implicit-GEMM 3x3 conv (stride 1, 2) and data gradient
against ATen convolutions on the same bf16 operands and against the 128x128 igemm tile.  Bit-equality with igemm is not expected (the
fp32 sums run over K in another order); the bound is the bf16 rounding of the stored value.  Models: models/cnn_backbone.py:182-187."""
import pytest
import torch
import torch.nn.functional as F

from _pkg import sub

pytestmark = pytest.mark.gpu
DEV = "cuda"
bf = torch.bfloat16


@pytest.mark.parametrize("entry", ["vqa_gemm8p", "vqa_gemm4w"])
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (512, 256, 128), (256, 512, 192), (1024, 768, 2304), (768, 256, 4608)])
def test_gemm8p_matches_matmul(M, N, K, entry):
    L = sub("_lib")
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(DEV, bf)
    B = torch.randn(N, K, generator=g).to(DEV, bf)
    C = torch.full((M, N), float("nan"), device=DEV, dtype=bf)
    L.call(entry, A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K)
    torch.cuda.synchronize()
    ref = A.float() @ B.float().t()
    assert torch.isfinite(C.float()).all()
    assert (C.float() - ref).abs().max().item() <= 6e-3 * ref.abs().max().item()
    with pytest.raises(RuntimeError):
        L.call(entry, A.data_ptr(), B.data_ptr(), C.data_ptr(), M + 1, N, K)
    if entry == "vqa_gemm4w":                                        # same K order per output element as the 8-phase kernel: same bits
        C2 = torch.empty_like(C)
        L.call("vqa_gemm8p", A.data_ptr(), B.data_ptr(), C2.data_ptr(), M, N, K)
        assert torch.equal(C, C2)


@pytest.mark.parametrize("B,H,W,C,N", [(4, 28, 28, 128, 256), (8, 14, 14, 256, 512), (3, 9, 11, 64, 256)])
def test_conv8p_stride2_forward_matches_aten(B, H, W, C, N):
    """The stage-entry 3x3 / 2 / pad 1 convs (models/cnn_backbone.py:243-247 main branch) on the same tile: gathered input pixels oh*2 + r - 1."""
    K, L = sub("kernels"), sub("_lib")
    g = torch.Generator().manual_seed(B + H + C + N)
    x = torch.randn(B * H * W, C, generator=g).to(DEV, bf)
    wq = (torch.randn(N, 3, 3, C, generator=g) * 0.05).to(DEV, bf)
    acc = torch.zeros(L.count("vqa_bn_acc_words", 2, N), device=DEV, dtype=torch.int64)
    out = K.conv8p(x, wq.view(N, 9 * C), B, H, W, C, N, stride=2, stats_acc=acc)
    torch.cuda.synchronize()
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    ref = F.conv2d(x.float().view(B, H, W, C).permute(0, 3, 1, 2).cpu(), wq.float().permute(0, 3, 1, 2).cpu(), stride=2, padding=1)
    got = out.float().view(B, Ho, Wo, N).permute(0, 3, 1, 2).cpu()
    assert ref.shape == got.shape and (got - ref).abs().max().item() <= 6e-3 * max(1.0, ref.abs().max().item())
    sums, flag = _acc_decode(acc, max(1, min(8, 512 // N)), 2, N)
    assert flag == 0 and (sums[0] - out.double().sum(0)).abs().max().item() <= 1e-3 * max(1.0, float(out.double().sum(0).abs().max()))
    with pytest.raises(RuntimeError):
        K.conv8p(x, wq.view(N, 9 * C), B, H, W, C, N, stride=2, transposed=1)


def _acc_decode(acc, R, K, C):
    n = R * K * C
    return (acc[:n].view(R, K, C).sum(0).double() / 16.0 + acc[n + 1: 2 * n + 1].view(R, K, C).sum(0).double() / float(1 << 50)), int(acc[n])


@pytest.mark.parametrize("B,H,W,C,N", [(4, 14, 14, 256, 256), (8, 7, 7, 512, 512), (3, 14, 14, 256, 256), (5, 7, 7, 512, 512), (2, 28, 28, 128, 256),
                                       (1, 5, 9, 64, 256), (16, 14, 14, 256, 256),
                                       (2, 28, 28, 128, 128), (3, 28, 28, 128, 128), (1, 6, 10, 64, 128), (5, 7, 7, 128, 384)])
@pytest.mark.parametrize("transposed", [0, 1])
def test_conv8p_matches_aten(B, H, W, C, N, transposed):
    """Forward: conv2d(x, w, padding=1).  transposed: the stride-1 data gradient dx = conv_transpose2d(dy, w) through the packed
    [Cin][(tap, Cout)] operand (vqa_pack_transpose), exactly how engine._block_bwd calls the igemm kernel."""
    K, L = sub("kernels"), sub("_lib")
    assert K.conv8p_ok(B, H, W, C, N)
    g = torch.Generator().manual_seed(B * 100 + H + C + N + transposed)
    x = torch.randn(B * H * W, C, generator=g).to(DEV, bf)
    if not transposed:
        w = (torch.randn(N, 3, 3, C, generator=g) * 0.05)                       # [Cout][R][S][Cin]
        wq = w.to(DEV, bf)
        acc = torch.zeros(L.count("vqa_bn_acc_words", 2, N), device=DEV, dtype=torch.int64)
        out = K.conv8p(x, wq.view(N, 9 * C), B, H, W, C, N, stats_acc=acc)
        ref = F.conv2d(x.float().view(B, H, W, C).permute(0, 3, 1, 2).cpu(), wq.float().permute(0, 3, 1, 2).cpu(), padding=1)
    else:
        # data gradient of a conv with weight w [Cout = C][3][3][Cin = N]: dx[b,h,w,n] = sum dy[b,h+1-r,w+1-s,c] w[c][r][s][n]
        w = torch.randn(C, 3, 3, N, generator=g) * 0.05
        wq = w.to(bf).float()
        wt = K.pack_transpose(wq.view(C, 9, N).to(DEV), bf)                      # [N][(tap, C)]
        out = K.conv8p(x, wt.view(N, 9 * C), B, H, W, C, N, transposed=1)
        ref = F.conv_transpose2d(x.float().view(B, H, W, C).permute(0, 3, 1, 2).cpu(), wq.permute(0, 3, 1, 2), padding=1)
    torch.cuda.synchronize()
    got = out.float().view(B, H, W, N).permute(0, 3, 1, 2).cpu()
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() <= 6e-3 * max(1.0, ref.abs().max().item())
    # the igemm tile on the same operands: same bf16 result up to one rounding of sums taken in a different order
    geom = (B, H, W, C, H, W, 3, 3, 1, 1)
    o2, _, _ = K.igemm(x, (wq.to(DEV).view(N, 9 * C) if not transposed else wt.view(N, 9 * C)), B * H * W, N, 9 * C, geom, dtype=bf, transposed=transposed)
    assert (out.float() - o2.float()).abs().max().item() <= 1.6e-2 * max(1.0, float(o2.float().abs().max()))
    if transposed:
        # the epilogue of the identity-path data gradient: (conv + addend * (addmask > 0)) * (outmask > 0), bit-equal to igemm's on the same conv value
        add = torch.randn(B * H * W, N, generator=g).to(DEV, bf)
        am = torch.randn(B * H * W, N, generator=g).to(DEV, bf)
        om = torch.randn(B * H * W, N, generator=g).to(DEV, bf)
        for kw in (dict(addend=add), dict(addend=add, addmask=am), dict(outmask=om), dict(addend=add, addmask=am, outmask=om)):
            o3 = K.conv8p(x, wt.view(N, 9 * C), B, H, W, C, N, transposed=1, **kw)
            e = out.float()
            if "addend" in kw:
                e = (e + (add.float() * (am.float() > 0) if "addmask" in kw else add.float())).to(bf).float()
            if "outmask" in kw:
                e = e * (om.float() > 0)
            assert torch.equal(o3.float(), e), kw.keys()
        # the BatchNorm-backward column sums of the stored tile (bn1 behind conv2's data gradient): same output bits, and the sums of
        # vqa_bn_bwd_reduce(self_mask) over (that output, y) -- equal as exact fixed-point totals up to the fp32 rounding of the partials
        y = torch.randn(B * H * W, N, generator=g).to(DEV, bf)
        coef = torch.stack([torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g) * 0.3, torch.randn(N, generator=g) * 0.1,
                            torch.rand(N, generator=g) + 0.5]).to(DEV).contiguous()
        words = L.count("vqa_bn_acc_words", 3, N)
        for kw in (dict(), dict(outmask=om)):
            facc, fref = torch.zeros(words, device=DEV, dtype=torch.int64), torch.zeros(words, device=DEV, dtype=torch.int64)
            o4 = K.conv8p(x, wt.view(N, 9 * C), B, H, W, C, N, transposed=1, bnred=(y, coef, facc), **kw)
            o5 = K.conv8p(x, wt.view(N, 9 * C), B, H, W, C, N, transposed=1, **kw)
            assert torch.equal(o4, o5)
            R = max(1, min(8, 512 // N))
            s4, f4 = _acc_decode(facc, R, 3, N)
            s5, f5 = s4, 0
            if 256 % (N // 8) == 0:                                              # (the row-block kernel's own shape rule; the model's widths all pass)
                L.call("vqa_bn_bwd_reduce", L.dt(bf), o5.data_ptr(), None, y.data_ptr(), coef.data_ptr(), None, None, fref.data_ptr(), B * H * W, N, 1, 1)
                s5, f5 = _acc_decode(fref, R, 3, N)
            gm = o5.double() * ((y.double() * coef[0].double() + coef[1].double()) > 0)          # (the mask itself is taken in fp32 on both sides)
            e0, e1 = gm.sum(0), (gm * (y.double() - coef[2].double()) * coef[3].double()).sum(0)
            assert f4 == 0 and f5 == 0
            for k, e in ((0, e0), (1, e1)):
                scale = max(1.0, float(gm.abs().sum(0).max()))
                assert (s4[k] - s5[k]).abs().max().item() <= 2e-5 * scale and (s4[k] - e).abs().max().item() <= 1e-3 * scale, (k, kw.keys())
            assert s4[2].abs().max().item() == 0
        # ... and for an already masked tile (the gradient handed to the previous block: outmask = its output) with that block's shortcut
        # BatchNorm sharing the gradient: g = the stored value, three rows (sum g | sum g xhat(y) | sum g xhat(y2)) = vqa_bn_bwd_reduce(y2 =, coef2 =)
        y2 = torch.randn(B * H * W, N, generator=g).to(DEV, bf)
        coef2 = torch.stack([torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g) * 0.3, torch.randn(N, generator=g) * 0.1,
                             torch.rand(N, generator=g) + 0.5]).to(DEV).contiguous()
        for dual in (False, True):
            facc = torch.zeros(words, device=DEV, dtype=torch.int64)
            o6 = K.conv8p(x, wt.view(N, 9 * C), B, H, W, C, N, transposed=1, addend=add, outmask=om,
                          bnred=(y, coef, facc, False, y2 if dual else None, coef2 if dual else None))
            o7 = K.conv8p(x, wt.view(N, 9 * C), B, H, W, C, N, transposed=1, addend=add, outmask=om)
            assert torch.equal(o6, o7)
            R = max(1, min(8, 512 // N))
            s6, f6 = _acc_decode(facc, R, 3, N)
            gd = o7.double()
            e = [gd.sum(0), (gd * (y.double() - coef[2].double()) * coef[3].double()).sum(0),
                 (gd * (y2.double() - coef2[2].double()) * coef2[3].double()).sum(0) if dual else torch.zeros(N, dtype=torch.float64, device=DEV)]
            scale = max(1.0, float(gd.abs().sum(0).max()))
            assert f6 == 0
            for k in range(3):
                assert (s6[k] - e[k]).abs().max().item() <= 1e-3 * scale, (dual, k)
            if not dual:
                assert s6[2].abs().max().item() == 0
        with pytest.raises(RuntimeError):                                        # a second BatchNorm only for an already masked tile
            K.conv8p(x, wt.view(N, 9 * C), B, H, W, C, N, transposed=1, bnred=(y, coef, torch.zeros(words, device=DEV, dtype=torch.int64), True, y2, coef2))
        with pytest.raises(RuntimeError):                                        # all three or none
            L.call("vqa_conv8p", x.data_ptr(), wt.data_ptr(), out.data_ptr(), None, None, None, None, y.data_ptr(), None, None, 1, None, None, B, H, W, C, N, 1, 1)
    if not transposed:
        R = max(1, min(8, 512 // N))
        sums, flag = _acc_decode(acc, R, 2, N)
        of = out.double()
        assert flag == 0
        assert (sums[0] - of.sum(0)).abs().max().item() <= 1e-3 * max(1.0, float(of.sum(0).abs().max()))
        assert (sums[1] - (of * of).sum(0)).abs().max().item() <= 1e-3 * float((of * of).sum(0).max())
        # bit-reproducible
        acc2 = torch.zeros_like(acc)
        out2 = K.conv8p(x, wq.view(N, 9 * C), B, H, W, C, N, stats_acc=acc2)
        assert torch.equal(out, out2) and torch.equal(acc, acc2)
