"""GPU parity of the kernel instantiations the BENCHMARK times (bf16, B=512/GPU), at shapes large enough that the C dispatch
picks them -- the small-shape cases of test_gpu_gemm_conv.py all fall to the 64x64 tile (fewer than 384 tiles).

Every case asserts through `vqa_igemm_variant` (the same host function the launch uses) which template runs, then compares
with ATen's CPU convolution / autograd on bf16-rounded operands (fp32 math), so only accumulation order and the final bf16
rounding differ: forward and data gradient 1.2e-2 of the output scale, weight gradient (fp32 output) 3e-3.

  stage 1  B=16   64->64   56x56   M=50176  392 tiles of 128x64   window loader <128,64,...,1>   (reference conv: cnn_backbone.py:182-187)
  stage 2  B=64  128->128  28x28   M=50176  392 tiles of 128x128  window loader <128,128,...,1>
  stage 3  B=128 256->256  14x14   M=25088  392 tiles
  stage 4  B=256 512->512   7x7    M=12544  392 tiles
"""
import pytest
import torch
import torch.nn.functional as F

from _pkg import sub

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


def _round(t):
    return t.to(BF).float()


def _relerr(got, ref):
    return float((got - ref).abs().max() / ref.abs().max().clamp(min=1e-6))


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


STAGES = [  # B, C, H, expected variant (BM*10000 + BN*10 + 1 = window loader)
    (16, 64, 56, 128 * 10000 + 64 * 10 + 1),
    (64, 128, 28, 128 * 10000 + 128 * 10 + 1),
    (128, 256, 14, 128 * 10000 + 128 * 10 + 1),
    (256, 512, 7, 128 * 10000 + 128 * 10 + 1),
]


@pytest.mark.parametrize("case", STAGES)
def test_window_loader_conv_fwd_and_dgrad_at_benchmark_tiles(case):
    """conv3x3/1 forward (+ BN partial statistics) and its data gradient (+ masked identity addend, the block's residual path)."""
    K = sub("kernels")
    B, C, H, want = case
    g = torch.Generator().manual_seed(C + H)
    x = _round(torch.randn(B, C, H, H, generator=g))
    w = _round(torch.randn(C, C, 3, 3, generator=g) * (2.0 / (C * 9)) ** 0.5)
    dy = _round(torch.randn(B, C, H, H, generator=g))
    add = _round(torch.randn(B, C, H, H, generator=g))
    msk = _round(torch.randn(B, C, H, H, generator=g))
    xr = x.clone().requires_grad_(True)
    yr = F.conv2d(xr, w, None, stride=1, padding=1)
    yr.backward(dy)
    M, Kw = B * H * H, 9 * C
    geom = (B, H, H, C, H, H, 3, 3, 1, 1)
    assert K.igemm_variant(BF, K.LOADER_NHWC, M, C, Kw, geom) == want
    w_krsc = w.permute(0, 2, 3, 1).contiguous().to(DEV)
    K.PROFILE = []
    try:
        y, stats, mt = K.igemm(_nhwc(x).to(DEV, BF), K.pack_rows(w_krsc.view(C, Kw), BF), M, C, Kw, geom, dtype=BF, want_stats=True)
        wt = K.pack_transpose(w_krsc.view(C, 9, C), BF)
        dx, _, _ = K.igemm(_nhwc(dy).to(DEV, BF), wt, M, C, Kw, geom, dtype=BF, transposed=1,
                           addend=_nhwc(add).to(DEV, BF), addmask=_nhwc(msk).to(DEV, BF))
        torch.cuda.synchronize()
        names = [r[0] for r in K.PROFILE]
    finally:
        K.PROFILE = None
    assert all(n.endswith(", 4, 64, 2, 2, 1>") for n in names), names           # window-loader instantiation, both launches
    y_ref = _nhwc(yr.detach()).reshape(M, C)
    assert _relerr(y.float().cpu(), y_ref) < 1.2e-2
    s = stats.sum(dim=0).cpu()
    assert mt == (M + 127) // 128
    assert _relerr(s[0], y_ref.sum(0)) < 5e-3 and _relerr(s[1], (y_ref ** 2).sum(0)) < 5e-3
    dx_ref = _nhwc(xr.grad + add * (msk > 0)).reshape(M, C)
    assert _relerr(dx.float().cpu(), dx_ref) < 1.2e-2


def test_plain_128_tiles_linear_at_benchmark_rows():
    """The token-side Linears of a B=512 step run the plain (non-window) 128x128 / 128x64 bf16 tiles: M = 512*49 rows."""
    K = sub("kernels")
    g = torch.Generator().manual_seed(3)
    for M, Kin, N, want in ((25088, 512, 256, 128 * 10000 + 128 * 10), (50176, 256, 64, 128 * 10000 + 64 * 10)):
        x = _round(torch.randn(M, Kin, generator=g))
        w = _round(torch.randn(N, Kin, generator=g) / Kin ** 0.5)
        b = torch.randn(N, generator=g)
        geom = K.linear_geom(M, Kin)
        assert K.igemm_variant(BF, K.LOADER_NHWC, M, N, Kin, geom) == want
        out, _, _ = K.igemm(x.to(DEV, BF), K.pack_rows(w.to(DEV), BF), M, N, Kin, geom, dtype=BF, bias=b.to(DEV), relu=1)
        torch.cuda.synchronize()
        assert _relerr(out.float().cpu(), torch.relu(x @ w.t() + b)) < 1.2e-2


WGRAD = [  # B, Cin, Cout, H, R, stride, pad, (kind, tile_n, tile_k) the B=512 benchmark launch of this layer gets, note
    (128, 128, 128, 28, 3, 1, 1, (0, 128, 128), "stage 2 (vqa_wgrad route; the engine uses wgrad3x3_c128p): 4-wave 128x128, XCD-aware order"),
    (272, 64, 128, 56, 3, 2, 1, (1, 128, 256), "stage-2 entry conv 3x3/2: 8-wave LDS-DMA kernel, 128x256 tile"),
    (64, 64, 128, 56, 1, 2, 0, (0, 128, 64), "stage-2 shortcut 1x1/2: 4-wave 128x64 tile"),
    (160, 256, 256, 14, 3, 1, 1, (1, 256, 256), "stage 3: 8-wave LDS-DMA kernel, 256x256 tile"),
    (256, 512, 512, 7, 3, 1, 1, (1, 256, 256), "stage 4: 8-wave LDS-DMA kernel, 256x256 tile, 144 output tiles"),
    (272, 128, 256, 28, 3, 2, 1, (1, 256, 256), "stage-3 entry conv 3x3/2: 8-wave LDS-DMA kernel"),
    (128, 128, 256, 28, 1, 2, 0, (0, 128, 128), "stage-3 shortcut 1x1/2: 4-wave 128x128 tile"),
]


@pytest.mark.parametrize("case", WGRAD, ids=[c[-1].split(":")[0] for c in WGRAD])
def test_wgrad_at_benchmark_tiles(case):
    """Every case first asserts -- through vqa_wgrad_plan, the host function the launch itself uses -- that the test shape gets the
    SAME kernel kind and tile as the B=512 benchmark launch of that layer (the planner switches kernels on the problem's FLOPs, so
    a small batch would silently test a different kernel), then compares with ATen on bf16-rounded operands."""
    K = sub("kernels")
    B, Cin, Cout, H, R, stride, pad, expect, _ = case
    g = torch.Generator().manual_seed(Cin * 3 + H + R)
    Ho = (H + 2 * pad - R) // stride + 1
    M, Kw = B * Ho * Ho, R * R * Cin
    plan = K.wgrad_plan(BF, 0, M, Cout, Kw, B, H, H, Cin, R, R)
    bench = K.wgrad_plan(BF, 0, 512 * Ho * Ho, Cout, Kw, 512, H, H, Cin, R, R)
    assert plan[:3] == expect and bench[:3] == expect, (plan, bench, expect)
    assert plan[3] > 1 and plan[4] == plan[3] * Cout * Kw            # split over M with one slab per split: the fixed-order two-pass path
    x = _round(torch.randn(B, Cin, H, H, generator=g))
    dy = _round(torch.randn(B, Cout, Ho, Ho, generator=g))
    w = torch.zeros(Cout, Cin, R, R, requires_grad=True)
    F.conv2d(x, w, None, stride=stride, padding=pad).backward(dy)
    geom = (B, H, H, Cin, Ho, Ho, R, R, stride, pad)
    dw = torch.zeros(Cout, Kw, device=DEV)
    K.wgrad(_nhwc(dy).to(DEV, BF), _nhwc(x).to(DEV, BF), dw, M, Cout, Kw, geom, dtype=BF)
    torch.cuda.synchronize()
    ref = w.grad.permute(0, 2, 3, 1).reshape(Cout, Kw)
    assert _relerr(dw.cpu(), ref) < 3e-3
    # += semantics: a second launch into the same buffer doubles it
    K.wgrad(_nhwc(dy).to(DEV, BF), _nhwc(x).to(DEV, BF), dw, M, Cout, Kw, geom, dtype=BF)
    torch.cuda.synchronize()
    assert _relerr(dw.cpu(), 2 * ref) < 3e-3


def test_token_side_wgrad_at_benchmark_rows():
    """Weight gradients of the token-side Linears at B=512: M = 10240 (text) / 25088 (image tokens) rows, small N x K."""
    K = sub("kernels")
    g = torch.Generator().manual_seed(11)
    for M, Kin, N in ((10240, 256, 768), (10240, 1024, 256), (10240, 256, 1024), (25088, 512, 256), (25088, 256, 512), (512, 256, 1000)):
        x = _round(torch.randn(M, Kin, generator=g))
        dy = _round(torch.randn(M, N, generator=g))
        dw = torch.zeros(N, Kin, device=DEV)
        K.wgrad(dy.to(DEV, BF), x.to(DEV, BF), dw, M, N, Kin, K.linear_geom(M, Kin), dtype=BF)
        torch.cuda.synchronize()
        assert _relerr(dw.cpu(), dy.t() @ x) < 3e-3, (M, Kin, N)


def test_weight_gradients_are_bit_reproducible():
    """Two-pass split (per-split slabs + fixed-order reduce) instead of float atomics: the same launch twice gives the SAME bits,
    for the 8-wave LDS-DMA kernel (stage-3 conv), the 4-wave kernel (stage-2 conv, token-side Linear) and fp32."""
    K = sub("kernels")
    g = torch.Generator().manual_seed(21)
    cases = [(256, 256, 256, 14, 3, BF, 1), (64, 128, 128, 28, 3, BF, 0), (64, 128, 128, 28, 3, torch.float32, 0)]
    for B, Cin, Cout, H, R, dtype, want_kind in cases:
        x = torch.randn(B * H * H, Cin, generator=g).to(DEV, dtype)
        dy = torch.randn(B * H * H, Cout, generator=g).to(DEV, dtype)
        M, Kw = B * H * H, R * R * Cin
        geom = (B, H, H, Cin, H, H, R, R, 1, 1)
        kind, tn, tk, nsplit, wsf = K.wgrad_plan(dtype, K.LOADER_NHWC, M, Cout, Kw, B, H, H, Cin, R, R)
        assert kind == want_kind and nsplit > 1 and wsf == nsplit * Cout * Kw
        outs = []
        for _ in range(3):
            dw = torch.zeros(Cout, Kw, device=DEV)
            K.wgrad(dy, x, dw, M, Cout, Kw, geom, dtype=dtype)
            torch.cuda.synchronize()
            outs.append(dw)
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    M, Kin, N = 10240, 256, 768
    x = torch.randn(M, Kin, generator=g).to(DEV, BF); dy = torch.randn(M, N, generator=g).to(DEV, BF)
    a, b = torch.zeros(N, Kin, device=DEV), torch.zeros(N, Kin, device=DEV)
    K.wgrad(dy, x, a, M, N, Kin, K.linear_geom(M, Kin), dtype=BF)
    K.wgrad(dy, x, b, M, N, Kin, K.linear_geom(M, Kin), dtype=BF)
    torch.cuda.synchronize()
    assert torch.equal(a, b)


def test_grouped_linear_weight_gradients_equal_single_launches():
    """vqa_wgrad_group (up to 8 token-side Linear weight gradients in one launch + one fixed-order reduce) against vqa_wgrad per
    job: bit-identical (same tiles, splits and slab order), and against torch on the bf16-rounded operands."""
    K = sub("kernels")
    T = torch.bfloat16
    g = torch.Generator().manual_seed(77)
    shapes = [(10240, 768, 256), (10240, 256, 256), (10240, 1024, 256), (10240, 256, 1024), (25088, 512, 256), (25088, 256, 512),
              (512, 1000, 256), (512, 256, 512)]                       # the last two: single split (no slab, no reduce blocks), head shapes
    jobs, singles, refs = [], [], []
    for M, N, Kw in shapes:
        assert K.wgrad_group_ok(T, M, N, Kw), (M, N, Kw)
        dy = (torch.randn(M, N, generator=g) * 0.1).to(T)
        x = torch.randn(M, Kw, generator=g).to(T)
        dyd, xd = dy.to(DEV), x.to(DEV)
        dw_g = torch.full((N, Kw), 0.5, device=DEV)                       # += semantics: start from a non-zero buffer
        dw_s = torch.full((N, Kw), 0.5, device=DEV)
        jobs.append((dyd, xd, dw_g, M, N, Kw))
        singles.append((dyd, xd, dw_s, M, N, Kw))
        refs.append((dy, x))
    K.wgrad_group(jobs, dtype=T)
    for dyd, xd, dw_s, M, N, Kw in singles:
        K.wgrad(dyd, xd, dw_s, M, N, Kw, K.linear_geom(M, Kw), dtype=T)
    torch.cuda.synchronize()
    for (dyd, xd, dw_g, M, N, Kw), (_, _, dw_s, *_), (dy, x) in zip(jobs, singles, refs):
        assert torch.equal(dw_g, dw_s), (M, N, Kw)
        ref = dy.float().t() @ x.float() + 0.5
        assert float((dw_g.cpu() - ref).abs().max() / ref.abs().max()) < 2e-3, (M, N, Kw)
    assert not K.wgrad_group_ok(T, 512, 1001, 256)                         # N % 8 != 0: goes through the padded single-launch path
    # a group of single-split jobs only (no workspace at all)
    dw2 = [torch.zeros_like(j[2]) for j in jobs[6:]]
    K.wgrad_group([(j[0], j[1], d, j[3], j[4], j[5]) for j, d in zip(jobs[6:], dw2)], dtype=T)
    torch.cuda.synchronize()
    for j, d in zip(jobs[6:], dw2):
        assert float((d - (j[2] - 0.5)).abs().max()) < 1e-5 * float(j[2].abs().max())
