"""GPU: the two pass-fusions around squeeze-excitation (SEAttention.forward, models/attention_modules.py:109-136, behind the stage's
last ResidualBlock, models/cnn_backbone.py:186-197), each against the unfused kernels (bit-equal tensors) and against torch math:

  vqa_bn_apply_pool   bn2 + residual + ReLU of the last block, ALSO leaving the SE global-average-pool sums per (sample, row chunk)
                      -> vqa_se_fwd(pool_part=...) skips its pooling pass;
  vqa_se_bwd(bn_*)    the SE backward apply pass ALSO leaving the BatchNorm-backward column sums of the gradient it stores
                      -> the last block's vqa_bn_bwd_reduce is skipped.
Shapes: the four stage outputs of the benchmark (56x56x64 ... 7x7x512) at a small batch, plus ragged sizes (rows not a multiple of
the chunk, one chunk only), both dtypes.  The whole-model goldens (tests/test_gpu_model.py) run with both fusions on (default)."""
import pytest
import torch

from _pkg import sub

pytestmark = pytest.mark.gpu
DEV = "cuda"
SHAPES = [(6, 56 * 56, 64), (5, 28 * 28, 128), (4, 14 * 14, 256), (3, 7 * 7, 512), (2, 15 * 15, 64), (3, 3 * 3, 128)]


def _coef(C, g):            # scale | shift | mean | invstd
    return torch.stack([torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3, torch.randn(C, generator=g) * 0.2,
                        torch.rand(C, generator=g) + 0.5])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("shape", SHAPES)
def test_bn_apply_pool_equals_bn_apply_plus_pooling(shape, dtype):
    L, K = sub("_lib"), sub("kernels")
    B, HW, C = shape
    Cr = max(C // 16, 1)
    g = torch.Generator().manual_seed(HW + C)
    y = torch.randn(B * HW, C, generator=g).to(DEV, dtype)
    res = torch.randn(B * HW, C, generator=g).to(DEV, dtype)
    coef = _coef(C, g).to(DEV)
    w1 = (torch.randn(Cr, C, generator=g) * 0.2).to(DEV)
    w2 = (torch.randn(C, Cr, generator=g) * 0.2).to(DEV)
    ref_out = K.bn_apply(y, coef, C, relu=True, res=res)
    out, part, chunks = K.bn_apply_pool(y, coef, C, True, B, HW, res=res)
    torch.cuda.synchronize()
    assert torch.equal(out, ref_out)                                        # the stored activation is unchanged, bit for bit
    assert part.shape == (B, chunks, C) and chunks == L.count("vqa_bn_apply_pool_chunks", L.dt(dtype), HW, C)
    sums = out.float().view(B, HW, C).sum(1)
    assert (part.sum(1) - sums).abs().max().item() < 1e-4 * max(1.0, sums.abs().max().item())
    # SE forward fed with the partial sums == SE forward that pools by itself
    def se(pool):
        pooled = torch.empty(B, C, device=DEV); hidden = torch.empty(B, Cr, device=DEV); scale = torch.empty(B, C, device=DEV)
        o = torch.empty_like(out)
        L.call("vqa_se_fwd", L.dt(dtype), out.data_ptr(), w1.data_ptr(), w2.data_ptr(), pooled.data_ptr(), hidden.data_ptr(), scale.data_ptr(),
               o.data_ptr(), B, HW, C, Cr, part.data_ptr() if pool else None, chunks if pool else 0)
        return o, pooled, scale
    a, pa, sa = se(True)
    b, pb, sb = se(False)
    torch.cuda.synchronize()
    assert (pa - pb).abs().max().item() < 1e-5 * max(1.0, pb.abs().max().item())
    assert (sa - sb).abs().max().item() < 1e-5
    tol = 1e-2 if dtype == torch.bfloat16 else 1e-5
    assert (a.float() - b.float()).abs().max().item() <= tol * max(1.0, b.float().abs().max().item())
    # and the torch formula of SEAttention.forward
    xf = out.float().view(B, HW, C)
    ref = xf * torch.sigmoid(torch.relu(xf.mean(1) @ w1.t()) @ w2.t())[:, None, :]
    assert (a.float().view(B, HW, C) - ref).abs().max().item() < (2e-2 if dtype == torch.bfloat16 else 1e-4) * max(1.0, ref.abs().max().item())
    # variants without / with the shortcut BatchNorm as residual keep working
    out0, part0, _ = K.bn_apply_pool(y, coef, C, True, B, HW)
    assert torch.equal(out0, K.bn_apply(y, coef, C, relu=True))
    rc = _coef(C, g).to(DEV)
    out2, _, _ = K.bn_apply_pool(y, coef, C, True, B, HW, res=res, rcoef=rc)
    assert torch.equal(out2, K.bn_apply(y, coef, C, relu=True, res=res, rcoef=rc))
    with pytest.raises(RuntimeError):                                       # a chunk count that does not match the shape is refused
        pooled = torch.empty(B, C, device=DEV)
        L.call("vqa_se_fwd", L.dt(dtype), out.data_ptr(), w1.data_ptr(), w2.data_ptr(), pooled.data_ptr(), pooled.data_ptr(), pooled.data_ptr(),
               out0.data_ptr(), B, HW, C, Cr, part.data_ptr(), chunks + 1)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("shape", SHAPES)
def test_se_backward_leaves_the_batchnorm_backward_sums(shape, dtype):
    L = sub("_lib")
    B, HW, C = shape
    Cr = max(C // 16, 1)
    g = torch.Generator().manual_seed(HW * 3 + C)
    x = torch.relu(torch.randn(B * HW, C, generator=g)).to(DEV, dtype)       # SE input = a post-ReLU activation
    dout = torch.randn(B * HW, C, generator=g).to(DEV, dtype)
    y2 = torch.randn(B * HW, C, generator=g).to(DEV, dtype)
    coef = _coef(C, g).to(DEV)
    w1 = (torch.randn(Cr, C, generator=g) * 0.2).to(DEV)
    w2 = (torch.randn(C, Cr, generator=g) * 0.2).to(DEV)
    pooled = x.float().view(B, HW, C).mean(1).contiguous()
    hidden = torch.relu(pooled @ w1.t()).contiguous()
    scale = torch.sigmoid(hidden @ w2.t()).contiguous()

    def run(fused):
        scratch = torch.empty(B * (2 * C + Cr), device=DEV)
        dx = torch.empty_like(x)
        dw1, dw2 = torch.zeros_like(w1), torch.zeros_like(w2)
        nblk = L.count("vqa_se_bwd_blocks", L.dt(dtype), B, HW, C)
        slab = torch.full((nblk, 3, C), float("nan"), device=DEV) if fused else None
        L.call("vqa_se_bwd", L.dt(dtype), dout.data_ptr(), x.data_ptr(), w1.data_ptr(), w2.data_ptr(), pooled.data_ptr(), hidden.data_ptr(),
               scale.data_ptr(), scratch.data_ptr(), dx.data_ptr(), dw1.data_ptr(), dw2.data_ptr(), B, HW, C, Cr, 1,
               y2.data_ptr() if fused else None, coef.data_ptr() if fused else None, slab.data_ptr() if fused else None, 0)
        return dx, dw1, dw2, slab
    dx, dw1, dw2, slab = run(True)
    dx0, dw10, dw20, _ = run(False)
    torch.cuda.synchronize()
    assert torch.equal(dx, dx0) and torch.equal(dw1, dw10) and torch.equal(dw2, dw20)      # the fused reduction changes nothing else
    assert torch.isfinite(slab).all()
    gq = dx.float()
    ref0 = gq.sum(0)
    ref1 = (gq * (y2.float() - coef[2]) * coef[3]).sum(0)
    s = slab.sum(0)
    tol = lambda r: 2e-4 * float(r.abs().max()) + 1e-4
    assert (s[0] - ref0).abs().max().item() < tol(ref0) and (s[1] - ref1).abs().max().item() < tol(ref1)
    assert float(s[2].abs().max()) == 0.0
    # the same sums from the standalone reduce kernel (what the unfused schedule runs on the stored gradient)
    nb = L.count("vqa_bn_bwd_blocks", B * HW)
    slab2 = torch.empty(nb, 3, C, device=DEV)
    L.call("vqa_bn_bwd_reduce", L.dt(dtype), dx.data_ptr(), None, y2.data_ptr(), coef.data_ptr(), None, None, slab2.data_ptr(), B * HW, C, 0, 0)
    torch.cuda.synchronize()
    s2 = slab2.sum(0)
    assert (s[0] - s2[0]).abs().max().item() < tol(ref0) and (s[1] - s2[1]).abs().max().item() < tol(ref1)
    # accumulator mode (what the bf16 training schedule runs: the one-pass-structure kernel + fixed-point sums, common.h)
    facc = torch.zeros(L.count("vqa_bn_acc_words", 3, C), device=DEV, dtype=torch.int64)
    scratch = torch.empty(B * (2 * C + Cr), device=DEV)
    dx3 = torch.empty_like(x); dw13, dw23 = torch.zeros_like(w1), torch.zeros_like(w2)
    L.call("vqa_se_bwd", L.dt(dtype), dout.data_ptr(), x.data_ptr(), w1.data_ptr(), w2.data_ptr(), pooled.data_ptr(), hidden.data_ptr(),
           scale.data_ptr(), scratch.data_ptr(), dx3.data_ptr(), dw13.data_ptr(), dw23.data_ptr(), B, HW, C, Cr, 1, y2.data_ptr(), coef.data_ptr(),
           facc.data_ptr(), 1)
    torch.cuda.synchronize()
    assert torch.equal(dx3, dx) and torch.equal(dw13, dw1) and torch.equal(dw23, dw2)
    R = max(1, min(8, 512 // C))
    sums = facc[: R * 3 * C].view(R, 3, C).sum(0).double() / float(1 << 40)
    assert int(facc[R * 3 * C]) == 0
    assert (sums[0].float() - ref0).abs().max().item() < tol(ref0) and (sums[1].float() - ref1).abs().max().item() < tol(ref1)
    with pytest.raises(RuntimeError):                                       # all three BatchNorm arguments or none
        scratch = torch.empty(B * (2 * C + Cr), device=DEV)
        L.call("vqa_se_bwd", L.dt(dtype), dout.data_ptr(), x.data_ptr(), w1.data_ptr(), w2.data_ptr(), pooled.data_ptr(), hidden.data_ptr(),
               scale.data_ptr(), scratch.data_ptr(), dx.data_ptr(), dw1.data_ptr(), dw2.data_ptr(), B, HW, C, Cr, 1, y2.data_ptr(), None, None, 0)


def test_engine_runs_the_fused_schedule_and_matches_the_unfused_one():
    """One fp32 train step of the full model with both fusions on (default) and off: same loss, gradients equal up to the fp32
    summation order of the pooled / reduced sums (nothing else differs; fp32 so that a last-bit difference is not amplified by
    bf16 re-rounding downstream), and the fused schedule really skips the two passes."""
    from _pkg import pkg
    from oracle import vqa_oracle as O
    P = pkg()
    cfg = O.full_config(dropout=0.0, answer_dropout=0.0)
    sd = O.init_state_dict(cfg, 3, jitter=True)
    batch = [t.to(DEV) for t in O.synthetic_batch(6, seed=9)]
    K = sub("kernels")
    res = {}
    for fused in (True, False):
        m = P.load_dropin().VQAModel(**cfg, compute_dtype="fp32")
        m.load_state_dict(sd)
        m = m.to(DEV).train()
        tr = P.trainer.HipTrainer(m)
        tr.engine.fuse_se_pool = tr.engine.fuse_se_bnred = fused
        K.PROFILE = []
        loss, _ = tr.step(*batch)
        torch.cuda.synchronize()
        names = [n for n, *_ in K.PROFILE]
        K.PROFILE = None
        res[fused] = (float(loss), tr.G.clone(), names)
    (l1, g1, n1), (l0, g0, n0) = res[True], res[False]
    assert sum(n.endswith(":bn_apply_pool") for n in n1) == 4 and sum(n.endswith(":bn_apply_pool") for n in n0) == 0
    assert sum(n.endswith(":bn_bwd_reduce") for n in n0) - sum(n.endswith(":bn_bwd_reduce") for n in n1) == 4
    assert abs(l1 - l0) < 1e-5
    # (train-mode BatchNorm at B = 6 amplifies a last-bit difference of the pooled sums to ~5e-3 of the gradient even in fp32;
    #  the kernel-level cases above are the tight ones)
    assert float((g1 - g0).norm() / g0.norm()) < 2e-2


# ---- BatchNorm finalize folded into the consumers: statistics and backward sums as fixed-point integer accumulators -------------
def _bn_ref_coef(y, gamma, beta, eps=1e-5):
    yf = y.double()
    mean, var = yf.mean(0), yf.var(0, unbiased=False)
    inv = 1.0 / torch.sqrt(var + eps)
    sc = gamma.double() * inv
    return torch.stack([sc, beta.double() - mean * sc, mean, inv]).float(), var


@pytest.mark.parametrize("case", [(4, 56 * 56, 64, "c64p"), (3, 28 * 28, 128, "igemm"), (2, 14 * 14, 256, "igemm"), (2, 7 * 7, 512, "igemm")])
def test_conv_statistics_as_fixed_point_accumulators_and_fused_apply(case):
    """conv (stats_mode = 1) -> vqa_bn_apply_acc  ==  conv (slab) -> vqa_bn_stats_finalize -> vqa_bn_apply: same stored conv output,
    coefficients within fp32 rounding of the fp64 reference, running statistics updated like nn.BatchNorm2d, outputs equal up to
    one bf16 ulp of the coefficient difference; residual / BatchNorm(residual) / pooling variants; bit-identical between two runs."""
    L, K = sub("_lib"), sub("kernels")
    B, HW, C, kind = case
    H = int(round(HW ** 0.5))
    g = torch.Generator().manual_seed(C + HW)
    bf = torch.bfloat16
    x = torch.randn(B * HW, C, generator=g).to(DEV, bf)
    w = (torch.randn(C, 9 * C, generator=g) * (2.0 / (9 * C)) ** 0.5).to(DEV, bf)
    geom = (B, H, H, C, H, H, 3, 3, 1, 1)

    def conv(acc):
        if kind == "c64p":
            return K.conv3x3_c64p(x, w, B, H, H, want_stats=True, stats_acc=acc)
        return K.igemm(x, w, B * HW, C, 9 * C, geom, dtype=bf, want_stats=True, stats_acc=acc)
    mk = lambda: ((torch.rand(C, generator=g) + 0.5).to(DEV), (torch.randn(C, generator=g) * 0.3).to(DEV))
    gamma, beta = mk()
    rgamma, rbeta = mk()
    y_ref, slab, mt = conv(None)
    rm0, rv0 = torch.randn(C, generator=g).to(DEV), (torch.rand(C, generator=g) + 0.5).to(DEV)
    rm, rv, nbt = rm0.clone(), rv0.clone(), torch.zeros((), device=DEV, dtype=torch.int64)
    coef_ref = K.bn_train_coef(slab, mt, C, B * HW, gamma, beta, rm, rv, nbt)
    res = torch.randn(B * HW, C, generator=g).to(DEV, bf)
    out_ref = K.bn_apply(y_ref, coef_ref, C, relu=True, res=res)
    runs = []
    for _ in range(2):
        acc = torch.zeros(L.count("vqa_bn_acc_words", 2, C), device=DEV, dtype=torch.int64)
        y, st, _ = conv(acc)
        assert st.data_ptr() == acc.data_ptr() and torch.equal(y, y_ref)
        rm2, rv2, nbt2 = rm0.clone(), rv0.clone(), torch.zeros((), device=DEV, dtype=torch.int64)
        out, coef, _, part, chunks = K.bn_apply_acc(y, acc, (gamma, beta, rm2, rv2, nbt2), C, True, B, HW, B * HW, res=res, pool=True)
        torch.cuda.synchronize()
        runs.append((acc.clone(), out.clone(), coef.clone(), rm2.clone(), rv2.clone(), part.clone()))
    assert all(torch.equal(a, b) for a, b in zip(runs[0], runs[1]))                  # integer sums: order-free, bit-reproducible
    acc, out, coef, rm2, rv2, part = runs[0]
    R = max(1, min(8, 512 // C))
    assert int(acc[R * 2 * C]) == 0 and int(nbt2) == 1
    ref, var = _bn_ref_coef(y_ref.float().cpu(), gamma.cpu(), beta.cpu())
    # (the 8-wave stage-1 kernel sums its fp32 accumulators, the implicit GEMM the bf16 values it stores: only the latter match
    #  statistics recomputed from the stored tensor to fp32 rounding; against the slab path both do)
    assert (coef.cpu() - ref).abs().max().item() < (2e-4 if kind == "c64p" else 2e-5) * max(1.0, ref.abs().max().item())
    assert (coef - coef_ref).abs().max().item() < 2e-5 * max(1.0, float(coef_ref.abs().max()))
    n = B * HW
    rtol = 1e-4 if kind == "c64p" else 2e-5
    assert (rm2.cpu() - (0.9 * rm0.cpu() + 0.1 * ref[2])).abs().max().item() < rtol
    assert (rv2.cpu() - (0.9 * rv0.cpu() + 0.1 * var.float() * n / (n - 1))).abs().max().item() < rtol
    assert (rm2 - rm).abs().max().item() < 1e-6 and (rv2 - rv).abs().max().item() < 1e-6          # == the slab + finalize path
    assert (out.float() - out_ref.float()).abs().max().item() <= 2e-2 * max(1.0, float(out_ref.float().abs().max()))
    assert (part.sum(1) - out.float().view(B, HW, C).sum(1)).abs().max().item() < 1e-3 * max(1.0, float(out.float().abs().max()) * HW ** 0.5)
    # BatchNorm(residual) variant: two accumulators finalized in the one pass, both coefficient sets published
    accd = torch.zeros(L.count("vqa_bn_acc_words", 2, C), device=DEV, dtype=torch.int64)
    yd, _, _ = conv(accd)
    acc2 = torch.zeros(L.count("vqa_bn_acc_words", 2, C), device=DEV, dtype=torch.int64)
    conv(acc2)
    bnp = lambda ga, be: (ga, be, rm0.clone(), rv0.clone(), torch.zeros((), device=DEV, dtype=torch.int64))
    o2, c2, cd, _, _ = K.bn_apply_acc(y_ref, acc2, bnp(gamma, beta), C, True, B, HW, B * HW, res=yd, racc=accd, rbn=bnp(rgamma, rbeta))
    refd, _ = _bn_ref_coef(y_ref.float().cpu(), rgamma.cpu(), rbeta.cpu())
    assert (cd.cpu() - refd).abs().max().item() < (2e-4 if kind == "c64p" else 2e-5) * max(1.0, refd.abs().max().item()) and torch.equal(c2, coef)
    expect = torch.relu(y_ref.float() * coef[0] + coef[1] + yd.float() * cd[0] + cd[1])
    assert (o2.float() - expect).abs().max().item() <= 1e-2 * max(1.0, float(expect.abs().max()))
    # a non-finite partial sum poisons the statistics instead of wrapping silently
    bad = acc.clone(); bad[R * 2 * C] = 1
    o3, c3, _, _, _ = K.bn_apply_acc(y_ref, bad, bnp(gamma, beta), C, True, B, HW, B * HW)
    assert torch.isnan(c3[2]).all() and torch.isnan(o3.float()).any()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("mode", ["plain", "self", "dual"])
@pytest.mark.parametrize("shape", [(6 * 56 * 56, 64), (5 * 28 * 28, 128), (3 * 49, 512), (450, 64)])
def test_batchnorm_backward_with_fixed_point_sums_equals_the_three_launch_form(shape, mode, dtype):
    """vqa_bn_bwd_reduce(acc_mode = 1) -> vqa_bn_bwd_apply_acc  ==  reduce (slab) -> vqa_bn_bwd_finalize -> vqa_bn_bwd_apply:
    dy (and the shortcut's dy2) equal up to the fp32 rounding of the coefficients, d gamma / d beta (+=) within 1e-5, against the
    closed-form BatchNorm backward in fp64, bit-identical between two runs."""
    K, L = sub("kernels"), sub("_lib")
    rows, C = shape
    g = torch.Generator().manual_seed(rows + C)
    rnd = lambda: torch.randn(rows, C, generator=g).to(DEV, dtype)
    dout, y, y2, act = rnd(), rnd(), rnd(), rnd()
    coef, coef2 = _coef(C, g).to(DEV), _coef(C, g).to(DEV)
    gamma, gamma2 = (torch.rand(C, generator=g) + 0.5).to(DEV), (torch.rand(C, generator=g) + 0.5).to(DEV)
    kw = dict(self_mask=(mode == "self"))
    if mode == "dual":
        kw.update(y2=y2, coef2=coef2, gamma2=gamma2)
    outact = None if mode == "self" else act

    def run(fused):
        dg, db, dg2, db2 = (torch.full((C,), 0.25, device=DEV) for _ in range(4))
        k2 = dict(kw)
        if mode == "dual":
            k2.update(dgamma2=dg2, dbeta2=db2)
        if fused:
            k2.update(facc=torch.zeros(L.count("vqa_bn_acc_words", 3, C), device=DEV, dtype=torch.int64))
        dy, dy2 = K.bn_bwd(dout, outact, y, coef, gamma, C, True, dg, db, **k2)
        torch.cuda.synchronize()
        return dy, dy2, dg, db, dg2, db2
    a, a2, ag, ab, ag2, ab2 = run(True)
    b, b2, bg, bb, bg2, bb2 = run(False)
    c, c2, cg, cb, _, _ = run(True)
    assert torch.equal(a, c) and torch.equal(ag, cg) and torch.equal(ab, cb)         # order-free integer sums
    tol = 1e-2 if dtype == torch.bfloat16 else 1e-5
    assert (a.float() - b.float()).abs().max().item() <= tol * max(1.0, float(b.float().abs().max()))
    assert (ag - bg).abs().max().item() < 1e-4 * max(1.0, float(bg.abs().max())) and (ab - bb).abs().max().item() < 1e-4 * max(1.0, float(bb.abs().max()))
    if mode == "dual":
        assert (a2.float() - b2.float()).abs().max().item() <= tol * max(1.0, float(b2.float().abs().max()))
        assert (ag2 - bg2).abs().max().item() < 1e-4 * max(1.0, float(bg2.abs().max()))
    # closed form in fp64
    yf, df = y.double().cpu(), dout.double().cpu()
    cf = coef.double().cpu()
    if mode == "self":
        gq = df * ((yf * cf[0] + cf[1]) > 0)
    else:
        gq = df * (act.double().cpu() > 0)
    xhat = (yf - cf[2]) * cf[3]
    dbeta, dgamma = gq.sum(0), (gq * xhat).sum(0)
    ref = gamma.double().cpu() * cf[3] * (gq - dbeta / rows - xhat * dgamma / rows)
    assert (a.double().cpu() - ref).abs().max().item() <= (2e-2 if dtype == torch.bfloat16 else 1e-4) * max(1.0, float(ref.abs().max()))
    assert (ag.double().cpu() - 0.25 - dgamma).abs().max().item() < 1e-3 * max(1.0, float(dgamma.abs().max()))
    assert (ab.double().cpu() - 0.25 - dbeta).abs().max().item() < 1e-3 * max(1.0, float(dbeta.abs().max()))
