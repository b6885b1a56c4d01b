"""GPU parity of the implicit-GEMM kernels (vqa_igemm / vqa_wgrad through the C ABI) against the oracle's
ATen CPU convolutions on identical inputs.  fp32: exact-fp32 MFMA, tolerance 2e-4 of the output scale.
bf16: operands are rounded to bf16 first so only accumulation order and the final bf16 rounding differ."""
import pytest
import torch
import torch.nn.functional as F

from _pkg import sub

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _tol(dtype):
    return 2e-4 if dtype == torch.float32 else 1.2e-2


def _round(t, dtype):
    return t.to(dtype).float()


def _relerr(got, ref):
    return float((got - ref).abs().max() / ref.abs().max().clamp(min=1e-6))


CONV_CASES = [  # B, Cin, Cout, H, R, stride, pad
    (2, 64, 64, 12, 3, 1, 1),
    (3, 64, 128, 14, 3, 2, 1),
    (2, 128, 256, 9, 3, 1, 1),
    (2, 64, 128, 14, 1, 2, 0),
    (1, 256, 512, 7, 3, 2, 1),
    (5, 128, 128, 28, 3, 1, 1),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(case, dtype):
    K = sub("kernels")
    B, Cin, Cout, H, R, stride, pad = case
    g = torch.Generator().manual_seed(hash(case) & 0xffff)
    x = _round(torch.randn(B, Cin, H, H, generator=g), dtype)
    w = _round(torch.randn(Cout, Cin, R, R, generator=g) * (2.0 / (Cin * R * R)) ** 0.5, dtype)
    Ho = (H + 2 * pad - R) // stride + 1
    dy = _round(torch.randn(B, Cout, Ho, Ho, generator=g), dtype)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, stride=stride, padding=pad)
    yr.backward(dy)

    x_d = x.permute(0, 2, 3, 1).contiguous().to(DEV, dtype)          # NHWC
    w_krsc = w.permute(0, 2, 3, 1).contiguous().to(DEV)              # [Cout][R][S][Cin] fp32 master layout
    dy_d = dy.permute(0, 2, 3, 1).contiguous().to(DEV, dtype)
    M, Kw = B * Ho * Ho, R * R * Cin
    geom = (B, H, H, Cin, Ho, Ho, R, R, stride, pad)
    wp = K.pack_rows(w_krsc.view(Cout, Kw), dtype)
    y, stats, mt = K.igemm(x_d, wp, M, Cout, Kw, geom, dtype=dtype, want_stats=True)
    torch.cuda.synchronize()
    y_ref = yr.detach().permute(0, 2, 3, 1).reshape(M, Cout)
    assert _relerr(y.float().cpu(), y_ref) < _tol(dtype)
    # BN partial statistics: column sums / sums of squares of the fp32 accumulators
    s = stats.sum(dim=0).cpu()
    assert _relerr(s[0], y_ref.sum(0)) < 5e-3 and _relerr(s[1], (y_ref ** 2).sum(0)) < 5e-3

    # data gradient: transposed gather over dy with [Cin][R][S][Cout] weights
    wt = K.pack_transpose(w_krsc.view(Cout, R * R, Cin), dtype)
    Md = B * H * H
    geom_d = (B, Ho, Ho, Cout, H, H, R, R, stride, pad)
    dx, _, _ = K.igemm(dy_d, wt, Md, Cin, R * R * Cout, geom_d, dtype=dtype, transposed=1)
    torch.cuda.synchronize()
    dx_ref = xr.grad.permute(0, 2, 3, 1).reshape(Md, Cin)
    assert _relerr(dx.float().cpu(), dx_ref) < _tol(dtype)

    # weight gradient (fp32, accumulated with atomics into a zeroed buffer)
    dw = torch.zeros(Cout, Kw, device=DEV, dtype=torch.float32)
    K.wgrad(dy_d, x_d, dw, M, Cout, Kw, geom, dtype=dtype)
    torch.cuda.synchronize()
    dw_ref = wr.grad.permute(0, 2, 3, 1).reshape(Cout, Kw)
    assert _relerr(dw.cpu(), dw_ref) < (2e-4 if dtype == torch.float32 else 3e-3)


LIN_CASES = [(40, 256, 256), (80, 256, 1024), (17, 1024, 256), (4, 256, 1000), (33, 32, 10), (196, 512, 256), (2000, 256, 256)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", LIN_CASES)
def test_linear_fwd_bwd_epilogues(case, dtype):
    K = sub("kernels")
    M, Kin, N = case
    g = torch.Generator().manual_seed(M * 7 + N)
    x = _round(torch.randn(M, Kin, generator=g), dtype)
    w = _round(torch.randn(N, Kin, generator=g) / Kin ** 0.5, dtype)
    b = torch.randn(N, generator=g)
    res = _round(torch.randn(M, N, generator=g), dtype)
    ref = torch.relu(x @ w.t() + b) + res
    geom = K.linear_geom(M, Kin)
    out, _, _ = K.igemm(x.to(DEV, dtype), K.pack_rows(w.to(DEV), dtype), M, N, Kin, geom, dtype=dtype, bias=b.to(DEV),
                        relu=1, addend=res.to(DEV, dtype))
    torch.cuda.synchronize()
    assert _relerr(out.float().cpu(), ref) < _tol(dtype)
    if N % 8 == 0:
        dy = _round(torch.randn(M, N, generator=g), dtype)
        wt = K.pack_transpose(w.to(DEV).view(N, 1, Kin), dtype)
        dx, _, _ = K.igemm(dy.to(DEV, dtype), wt, M, Kin, N, K.linear_geom(M, N), dtype=dtype)
        dw = torch.zeros(N, Kin, device=DEV)
        K.wgrad(dy.to(DEV, dtype), x.to(DEV, dtype), dw, M, N, Kin, geom, dtype=dtype)
        torch.cuda.synchronize()
        assert _relerr(dx.float().cpu(), dy @ w) < _tol(dtype)
        assert _relerr(dw.cpu(), dy.t() @ x) < (2e-4 if dtype == torch.float32 else 3e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("drop_p", [0.0, 0.1, 0.3])
@pytest.mark.parametrize("case", [(40, 256, 1024), (80, 1024, 256), (17, 512, 256), (512, 256, 1000), (2000, 256, 256), (10240, 1024, 256)])
def test_linear_dgrad_with_the_activation_mask_in_its_epilogue(case, drop_p, dtype):
    """vqa_linear_dgrad_act (round 4): the data gradient of a Linear that leaves with the ReLU(+dropout) backward of the layer in front
    applied == vqa_igemm followed by vqa_bias_act_bwd, BIT-equal (the keep scale multiplies the value already rounded to the compute
    dtype), and == the torch formula.  Shapes: FFN fc2 -> fc1 (models/text_encoder.py:309-317), answer head (models/vqa_model.py:74-82)."""
    K, L = sub("kernels"), sub("_lib")
    M, Kin, N = case                                    # dz [M][N], W [N][Kin], h = the masked layer's output [M][Kin]
    g = torch.Generator().manual_seed(M + 3 * N)
    dz = _round(torch.randn(M, N, generator=g), dtype).to(DEV, dtype)
    w = _round(torch.randn(N, Kin, generator=g) / N ** 0.5, dtype)
    h = torch.relu(torch.randn(M, Kin, generator=g))
    h = _round(h * (torch.rand(M, Kin, generator=g) > drop_p), dtype).to(DEV, dtype)      # out > 0 encodes ReLU and dropout
    wt = K.pack_transpose(w.to(DEV).view(N, 1, Kin), dtype)
    fused = K.linear_dgrad_act(dz, wt, M, Kin, N, dtype=dtype, outact=h, drop_p=drop_p)
    dh, _, _ = K.igemm(dz, wt, M, Kin, N, K.linear_geom(M, N), dtype=dtype)
    two = torch.empty_like(dh)
    L.call("vqa_bias_act_bwd", L.dt(dtype), dh.data_ptr(), h.data_ptr(), two.data_ptr(), None, M, Kin, float(drop_p), 0, None, 0)
    torch.cuda.synchronize()
    assert torch.equal(fused, two)
    ref = (dz.float().cpu() @ w) * (h.float().cpu() > 0) / (1.0 - drop_p)
    assert _relerr(fused.float().cpu(), ref) < _tol(dtype)
    assert (fused[h <= 0] == 0).all()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_stem_conv_fwd_wgrad(dtype):
    K = sub("kernels")
    B, H = 2, 40
    g = torch.Generator().manual_seed(5)
    img = torch.randn(B, 3, H, H, generator=g)
    w = _round(torch.randn(64, 3, 7, 7, generator=g) * 0.1, dtype)
    Ho = (H + 6 - 7) // 2 + 1
    dy = _round(torch.randn(B, 64, Ho, Ho, generator=g), dtype)
    img_r = _round(img, dtype)
    wr = w.clone().requires_grad_(True)
    yr = F.conv2d(img_r, wr, None, stride=2, padding=3)
    yr.backward(dy)
    M = B * Ho * Ho
    BK = 64 if dtype == torch.bfloat16 else 32
    Kp = (147 + BK - 1) // BK * BK
    w_krsc = w.permute(0, 2, 3, 1).contiguous().to(DEV)
    wp = K.pack_rows(w_krsc.view(64, 147), dtype, Kp)
    geom = (B, H, H, 3, Ho, Ho, 7, 7, 2, 3)
    y, stats, mt = K.igemm(img.to(DEV), wp, M, 64, Kp, geom, dtype=dtype, loader=K.LOADER_STEM, want_stats=True)
    torch.cuda.synchronize()
    y_ref = yr.detach().permute(0, 2, 3, 1).reshape(M, 64)
    assert _relerr(y.float().cpu(), y_ref) < _tol(dtype)
    dw = torch.zeros(64, 147, device=DEV)
    K.wgrad(dy.permute(0, 2, 3, 1).contiguous().to(DEV, dtype), img.to(DEV), dw, M, 64, 147, geom, dtype=dtype, loader=K.LOADER_STEM)
    torch.cuda.synchronize()
    assert _relerr(dw.cpu(), wr.grad.permute(0, 2, 3, 1).reshape(64, 147)) < (2e-4 if dtype == torch.float32 else 3e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [(2, 64, 128, 16, True), (3, 128, 256, 12, True), (2, 256, 512, 14, False)])
def test_stride2_dgrad_by_parity_classes(case, dtype):
    """conv3x3/2 (+1x1/2 shortcut) data gradient in one launch == autograd of the two ATen convs."""
    K = sub("kernels")
    B, Cin, Cout, H, shortcut = case
    g = torch.Generator().manual_seed(Cin + H)
    w1 = _round(torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05, dtype)
    wd = _round(torch.randn(Cout, Cin, 1, 1, generator=g) * 0.05, dtype)
    Ho = H // 2
    dy = _round(torch.randn(B, Cout, Ho, Ho, generator=g), dtype)
    dyd = _round(torch.randn(B, Cout, Ho, Ho, generator=g), dtype)
    x = torch.zeros(B, Cin, H, H, requires_grad=True)
    out = (F.conv2d(x, w1, None, stride=2, padding=1) * dy).sum()
    if shortcut:
        out = out + (F.conv2d(x, wd, None, stride=2, padding=0) * dyd).sum()
    out.backward()
    ktot = (10 if shortcut else 9) * Cout
    wt = torch.empty(Cin, ktot, device=DEV, dtype=dtype)
    K.pack_transpose(w1.permute(0, 2, 3, 1).contiguous().to(DEV).view(Cout, 9, Cin), dtype, out=wt, ldo=ktot, col0=0)
    if shortcut:
        K.pack_transpose(wd.permute(0, 2, 3, 1).contiguous().to(DEV).view(Cout, 1, Cin), dtype, out=wt, ldo=ktot, col0=9 * Cout)
    dx = K.dgrad_s2(dy.permute(0, 2, 3, 1).contiguous().to(DEV, dtype), dyd.permute(0, 2, 3, 1).contiguous().to(DEV, dtype) if shortcut else None,
                    wt, B, Ho, Ho, Cout, H, H, Cin, 3, 1, dtype=dtype)
    torch.cuda.synchronize()
    ref = x.grad.permute(0, 2, 3, 1).reshape(B * H * H, Cin)
    assert _relerr(dx.float().cpu(), ref) < _tol(dtype)


@pytest.mark.parametrize("hw", [(224, 224), (64, 64), (96, 160)])
def test_stem_conv_bf16_dedicated_kernel(hw):
    K = sub("kernels")
    H, W = hw
    B = 2
    g = torch.Generator().manual_seed(H)
    img = torch.randn(B, 3, H, W, generator=g)
    w = _round(torch.randn(64, 3, 7, 7, generator=g) * 0.1, torch.bfloat16)
    ref = F.conv2d(_round(img, torch.bfloat16), w, None, stride=2, padding=3)
    Ho, Wo = ref.shape[2], ref.shape[3]
    assert K.stem_conv_blocks(B, H, W) == B * Ho // 4
    wst = torch.empty(64, 192, device=DEV, dtype=torch.bfloat16)
    sub("_lib").call("vqa_stem_pack", w.permute(0, 2, 3, 1).contiguous().to(DEV).data_ptr(), wst.data_ptr())
    y, stats, nb = K.stem_conv(img.to(DEV), wst, B, H, W, True)
    torch.cuda.synchronize()
    y_ref = ref.permute(0, 2, 3, 1).reshape(-1, 64)
    assert _relerr(y.float().cpu(), y_ref) < 1.2e-2
    s = stats.sum(0).cpu()
    assert _relerr(s[0], y_ref.sum(0)) < 5e-3 and _relerr(s[1], (y_ref ** 2).sum(0)) < 5e-3


@pytest.mark.parametrize("hw", [(224, 224), (64, 64), (96, 160), (384, 384), (32, 72)])
def test_inference_stem_in_one_launch(hw):
    """vqa_stem_conv_pool == conv7x7/2 -> BatchNorm(running statistics) -> ReLU -> MaxPool3x3/2 (models/cnn_backbone.py:349-354, eval):
    against ATen on the bf16-rounded image / weights (fp32 math: the fused kernel normalises the fp32 accumulators, so only the final
    bf16 rounding differs), and against the two-kernel path (which rounds the conv output to bf16 first).  Shapes: the benchmark's,
    the stress config's, ragged last column tiles (pooled width 40 / 18: not a multiple of 7), two and four row blocks per image."""
    K, L = sub("kernels"), sub("_lib")
    H, W = hw
    B = 3
    g = torch.Generator().manual_seed(H * 7 + W)
    img = torch.randn(B, 3, H, W, generator=g)
    w = _round(torch.randn(64, 3, 7, 7, generator=g) * 0.1, torch.bfloat16)
    scale, shift = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.5
    y = F.conv2d(_round(img, torch.bfloat16), w, None, stride=2, padding=3)
    ref = F.max_pool2d(torch.relu(y * scale[None, :, None, None] + shift[None, :, None, None]), 3, 2, 1)
    Hp, Wp = ref.shape[2], ref.shape[3]
    assert L.count("vqa_stem_conv_pool_ok", B, H, W) == 1
    wst = torch.empty(64, 192, device=DEV, dtype=torch.bfloat16)
    L.call("vqa_stem_pack", w.permute(0, 2, 3, 1).contiguous().to(DEV).data_ptr(), wst.data_ptr())
    coef = torch.cat([scale, shift, torch.zeros(128)]).to(DEV)
    x = K.stem_conv_pool(img.to(DEV), wst, coef, B, H, W)
    torch.cuda.synchronize()
    assert x.shape == (B * Hp * Wp, 64)
    got = x.float().cpu().view(B, Hp, Wp, 64).permute(0, 3, 1, 2)
    assert (got - ref).abs().max().item() <= 4e-3 * max(1.0, ref.abs().max().item())          # one bf16 rounding of the result
    # the two-kernel path: conv -> bf16 y -> BN + ReLU + pool
    if K.stem_conv_blocks(B, H, W) > 0:
        y2, _, _ = K.stem_conv(img.to(DEV), wst, B, H, W, False)
        x2 = torch.empty_like(x); idx = torch.empty(x.shape, device=DEV, dtype=torch.uint8)
        L.call("vqa_stem_pool_fwd", 1, y2.data_ptr(), coef.data_ptr(), x2.data_ptr(), idx.data_ptr(), B, y.shape[2], y.shape[3], 64)
        torch.cuda.synchronize()
        assert (x.float() - x2.float()).abs().max().item() <= 2e-2 * max(1.0, float(x2.float().abs().max()))
    # unsupported shapes are refused, not mis-computed: odd conv output, pooled rows not a multiple of 4
    assert L.count("vqa_stem_conv_pool_ok", B, 226, 224) == 0 and L.count("vqa_stem_conv_pool_ok", B, 40, 40) == 0
    assert K.stem_conv_pool(img.to(DEV)[:, :, :40, :40].contiguous(), wst, coef, B, 40, 40) is None


@pytest.mark.parametrize("hw", [(224, 224), (64, 64), (96, 160)])
def test_stem_wgrad_bf16_dedicated_kernel(hw):
    K = sub("kernels")
    H, W = hw
    B = 3
    g = torch.Generator().manual_seed(W)
    img = torch.randn(B, 3, H, W, generator=g)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    dy = _round(torch.randn(B, 64, Ho, Wo, generator=g), torch.bfloat16)
    w = torch.zeros(64, 3, 7, 7, requires_grad=True)
    F.conv2d(_round(img, torch.bfloat16), w, None, stride=2, padding=3).backward(dy)
    dw = torch.zeros(64, 147, device=DEV)
    K.stem_wgrad(img.to(DEV), dy.permute(0, 2, 3, 1).contiguous().to(DEV, torch.bfloat16), dw, B, H, W)
    torch.cuda.synchronize()
    assert _relerr(dw.cpu(), w.grad.permute(0, 2, 3, 1).reshape(64, 147)) < 3e-3


@pytest.mark.parametrize("case", [(2, 56, 56), (3, 16, 24), (1, 8, 8), (20, 56, 56)])     # (20,56,56): 280 row blocks > 256 persistent workgroups
def test_conv3x3_c64_patch_kernels(case):
    """stage-1 LDS-patch kernels (bf16): forward + stats, data gradient with the masked identity addend, weight gradient."""
    K = sub("kernels")
    B, H, W = case
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(H * 7 + W)
    x = _round(torch.randn(B, 64, H, W, generator=g), dtype)
    w = _round(torch.randn(64, 64, 3, 3, generator=g) * (2.0 / 576) ** 0.5, dtype)
    dy = _round(torch.randn(B, 64, H, W, generator=g), dtype)
    add = _round(torch.randn(B, 64, H, W, generator=g), dtype)
    msk = _round(torch.randn(B, 64, H, W, generator=g), dtype)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, stride=1, padding=1)
    yr.backward(dy)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(DEV, dtype)
    w_krsc = w.permute(0, 2, 3, 1).contiguous().to(DEV)
    assert K.c64_blocks(B, H, W) > 0
    y, stats, nb = K.conv3x3_c64(nhwc(x), K.pack_rows(w_krsc.view(64, 576), dtype), B, H, W, want_stats=True)
    torch.cuda.synchronize()
    y_ref = yr.detach().permute(0, 2, 3, 1).reshape(-1, 64)
    assert _relerr(y.float().cpu(), y_ref) < _tol(dtype)
    s = stats.sum(0).cpu()
    assert _relerr(s[0], y_ref.sum(0)) < 5e-3 and _relerr(s[1], (y_ref ** 2).sum(0)) < 5e-3
    wflip = K.pack_transpose(w_krsc.view(64, 9, 64), dtype, flip=True)
    dx, _, _ = K.conv3x3_c64(nhwc(dy), wflip, B, H, W, addend=nhwc(add), addmask=nhwc(msk))
    torch.cuda.synchronize()
    dx_ref = (xr.grad + add * (msk > 0)).permute(0, 2, 3, 1).reshape(-1, 64)
    assert _relerr(dx.float().cpu(), dx_ref) < _tol(dtype)
    dw = torch.zeros(64, 576, device=DEV)
    K.wgrad3x3_c64(nhwc(x), nhwc(dy), dw, B, H, W)
    torch.cuda.synchronize()
    assert _relerr(dw.cpu(), wr.grad.permute(0, 2, 3, 1).reshape(64, 576)) < 3e-3


@pytest.mark.parametrize("B", [1, 3, 40])                                  # 40 images: 280 row blocks > 128 persistent workgroups per half
def test_wgrad3x3_c128_stage2_kernel(B):
    """Stage-2 weight gradient (128 -> 128, 28 x 28, bf16): 8-wave LDS-DMA kernel against torch's conv2d weight gradient on the
    bf16-rounded operands; += semantics; bit-reproducible."""
    K = sub("kernels")
    H = W = 28
    assert K.c128_wgrad_blocks(B, H, W) > 0 and K.c128_wgrad_blocks(B, 14, 14) == 0
    g = torch.Generator().manual_seed(500 + B)
    x = _round(torch.randn(B, 128, H, W, generator=g), torch.bfloat16)
    dy = _round(torch.randn(B, 128, H, W, generator=g) * 0.1, torch.bfloat16)
    ref = torch.nn.grad.conv2d_weight(x, (128, 128, 3, 3), dy, stride=1, padding=1).permute(0, 2, 3, 1).reshape(128, 1152)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(DEV, torch.bfloat16)
    xd, dyd = nhwc(x), nhwc(dy)
    outs = []
    for rep in range(2):
        dw = torch.full((128, 1152), 0.25, device=DEV)
        K.wgrad3x3_c128(xd, dyd, dw, B, H, W)
        outs.append(dw)
    torch.cuda.synchronize()
    assert _relerr(outs[0].cpu() - 0.25, ref) < 3e-3
    assert torch.equal(outs[0], outs[1])


def test_pack_transpose_batch_matches_single_launches():
    """The tiled one-launch weight transpose (begin_step) against the per-weight kernel, including taps, flip, a column
    offset inside a wider row, and dimensions that are not multiples of the 32x32 tile."""
    K, L = sub("kernels"), sub("_lib")
    g = torch.Generator().manual_seed(9)
    pieces = [(10, 1, 40, False, 0, 0), (64, 9, 64, False, 0, 0), (128, 9, 64, True, 0, 0), (96, 1, 24, False, 9 * 96, 1), (8, 9, 16, False, 0, 0)]
    for dtype in (torch.bfloat16, torch.float32):
        flat_parts, rows, refs, src, dst, blk = [], [], [], 0, 0, 0
        for n, tt, c, flip, col0, wide in pieces:
            w = torch.randn(n, tt, c, generator=g)
            flat_parts.append(w.reshape(-1))
            ld = tt * n + (col0 if wide else 0)
            ref = torch.full((c, ld), 0.0, device=DEV, dtype=dtype)
            K.pack_transpose(w.to(DEV), dtype, out=ref, ldo=ld, col0=col0, flip=flip)
            rows.append([src, dst, n, tt, c, ld, col0, int(flip), blk, 0])
            refs.append((dst, c, ld, col0, tt * n, ref))
            src += n * tt * c; dst += (c * ld + 7) // 8 * 8; blk += tt * ((n + 31) // 32) * ((c + 31) // 32)
        flat = torch.cat(flat_parts).to(DEV)
        out = torch.zeros(dst, device=DEV, dtype=dtype)
        table = torch.tensor(rows, dtype=torch.int64).to(DEV)
        L.call("vqa_pack_transpose_batch", int(dtype == torch.bfloat16), flat.data_ptr(), out.data_ptr(), table.data_ptr(), len(rows), blk)
        torch.cuda.synchronize()
        for d0, c, ld, col0, width, ref in refs:
            got = out[d0: d0 + c * ld].view(c, ld)
            assert torch.equal(got[:, col0: col0 + width], ref[:, col0: col0 + width])


@pytest.mark.parametrize("case", [(2, 56, 56), (3, 16, 24), (1, 8, 8), (40, 56, 56), (5, 24, 16), (3, 96, 96), (2, 12, 16), (70, 20, 40)])
def test_conv3x3_c64_dma_patch_kernel(case):
    """8-wave persistent patch kernel with LDS-DMA patches (stage-1 forward / addend-free data gradient, bf16): output and BN partial
    statistics against ATen, forward weights and the flipped-transposed pack; (40, 56, 56) gives 280 blocks on the 256-workgroup
    persistent grid, i.e. workgroups that walk two blocks through both patch buffers."""
    K = sub("kernels")
    B, H, W = case
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(H * 11 + W + B)
    x = _round(torch.randn(B, 64, H, W, generator=g), dtype)
    w = _round(torch.randn(64, 64, 3, 3, generator=g) * (2.0 / 576) ** 0.5, dtype)
    dy = _round(torch.randn(B, 64, H, W, generator=g), dtype)
    xr = x.clone().requires_grad_(True)
    yr = F.conv2d(xr, w, None, stride=1, padding=1)
    yr.backward(dy)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(DEV, dtype)
    w_krsc = w.permute(0, 2, 3, 1).contiguous().to(DEV)
    # 8 output rows per block where two 10 x (W+2) patches fit the LDS and H allows it, else 4 (the 96 x 96 maps of the stress shape)
    rbp = 8 if (H % 8 == 0 and 2 * 10 * (W + 2) * 128 + 2048 <= 160 * 1024) else 4
    assert K.c64p_blocks(B, H, W) == min(B * H // rbp, 256)
    y, stats, nb = K.conv3x3_c64p(nhwc(x), K.pack_rows(w_krsc.view(64, 576), dtype), B, H, W, want_stats=True)
    torch.cuda.synchronize()
    y_ref = yr.detach().permute(0, 2, 3, 1).reshape(-1, 64)
    assert _relerr(y.float().cpu(), y_ref) < _tol(dtype)
    s = stats.sum(0).cpu()
    assert _relerr(s[0], y_ref.sum(0)) < 5e-3 and _relerr(s[1], (y_ref ** 2).sum(0)) < 5e-3
    wflip = K.pack_transpose(w_krsc.view(64, 9, 64), dtype, flip=True)
    dx, _, _ = K.conv3x3_c64p(nhwc(dy), wflip, B, H, W)
    torch.cuda.synchronize()
    assert _relerr(dx.float().cpu(), xr.grad.permute(0, 2, 3, 1).reshape(-1, 64)) < _tol(dtype)
    # the conv1 data gradient of a residual block: identity-path gradient and ReLU masks in the per-tile epilogue -- the bf16 conv value
    # (bit-identical to the epilogue-free launch) + addend * (addmask > 0), re-rounded, then masked: vqa_igemm's epilogue bit for bit
    add, am, om = (nhwc(torch.randn(B, 64, H, W, generator=g)).view(-1, 64) for _ in range(3))
    for kw in (dict(), dict(addmask=am), dict(outmask=om), dict(addmask=am, outmask=om)):
        d2 = K.conv3x3_c64p_epi(nhwc(dy), wflip, B, H, W, addend=add, **kw)
        e = (dx.float() + (add.float() * (am.float() > 0) if "addmask" in kw else add.float())).to(dtype).float()
        if "outmask" in kw:
            e = e * (om.float() > 0)
        assert torch.equal(d2.float(), e), kw.keys()
    with pytest.raises(RuntimeError):
        sub("_lib").call("vqa_conv3x3_c64p_epi", dy.data_ptr(), wflip.data_ptr(), dx.data_ptr(), None, None, None, B, H, W)
    # conv2's data gradient that also leaves bn1's backward column sums: same output bits; sums = vqa_bn_bwd_reduce(self_mask) over (dx, y)
    L = sub("_lib")
    yv = add                                                                      # any bf16 tensor serves as the BatchNorm input
    coef = torch.stack([torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.3, torch.randn(64, generator=g) * 0.1,
                        torch.rand(64, generator=g) + 0.5]).to(DEV).contiguous()
    words = L.count("vqa_bn_acc_words", 3, 64)
    facc, fref = torch.zeros(words, device=DEV, dtype=torch.int64), torch.zeros(words, device=DEV, dtype=torch.int64)
    d3 = K.conv3x3_c64p_bnred(nhwc(dy), wflip, B, H, W, yv, coef, facc)
    assert torch.equal(d3, dx)
    L.call("vqa_bn_bwd_reduce", L.dt(dtype), dx.data_ptr(), None, yv.data_ptr(), coef.data_ptr(), None, None, fref.data_ptr(), B * H * W, 64, 1, 1)
    def dec(acc, R=8, Kk=3, C=64):
        n = R * Kk * C
        return acc[:n].view(R, Kk, C).sum(0).double() / 16.0 + acc[n + 1: 2 * n + 1].view(R, Kk, C).sum(0).double() / float(1 << 50), int(acc[n])
    (s3, f3), (s4, f4) = dec(facc), dec(fref)
    gm = dx.double() * ((yv.double() * coef[0].double() + coef[1].double()) > 0)
    scale = max(1.0, float(gm.abs().sum(0).max()))
    assert f3 == 0 and f4 == 0 and s3[2].abs().max().item() == 0
    assert (s3[0] - s4[0]).abs().max().item() <= 2e-5 * scale and (s3[1] - s4[1]).abs().max().item() <= 2e-5 * scale
    assert (s3[0] - gm.sum(0)).abs().max().item() <= 1e-3 * scale
    assert K.c64p_blocks(2, 10, 10) == 0 and K.c64p_blocks(2, 6, 16) == 0 and K.c64p_blocks(2, 8, 128) == 0   # refused, not mangled


@pytest.mark.parametrize("case", [(3, 96, 96), (2, 10, 16), (70, 20, 40), (2, 56, 56), (3, 12, 88)])
def test_wgrad3x3_c64_row_block_variants(case):
    """Stage-1 weight gradient on shapes the 4-rows-per-block 8-wave kernel does not take: 96 x 96 maps (the 384 x 384 stress
    configuration: two rows per block so that both operands of two blocks fit the LDS), heights that are a multiple of 2 but not of
    4, more row blocks than persistent workgroups.  Against torch's conv2d weight gradient on the bf16-rounded operands; += semantics;
    bit-reproducible (per-workgroup slabs + fixed-order reduce)."""
    K = sub("kernels")
    B, H, W = case
    assert K.c64w_blocks(B, H, W) > 0 and K.c64w_blocks(B, 7, 16) == 0 and K.c64w_blocks(B, 8, 12) == 0 and K.c64w_blocks(B, 12, 104) == 0
    g = torch.Generator().manual_seed(B + H * 3 + W)
    x = _round(torch.randn(B, 64, H, W, generator=g), torch.bfloat16)
    dy = _round(torch.randn(B, 64, H, W, generator=g) * 0.1, torch.bfloat16)
    ref = torch.nn.grad.conv2d_weight(x, (64, 64, 3, 3), dy, stride=1, padding=1).permute(0, 2, 3, 1).reshape(64, 576)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(DEV, torch.bfloat16)
    xd, dyd = nhwc(x), nhwc(dy)
    outs = []
    for rep in range(2):
        dw = torch.full((64, 576), 0.25, device=DEV)
        K.wgrad3x3_c64(xd, dyd, dw, B, H, W)
        outs.append(dw)
    torch.cuda.synchronize()
    assert _relerr(outs[0].cpu() - 0.25, ref) < 3e-3
    assert torch.equal(outs[0], outs[1])
