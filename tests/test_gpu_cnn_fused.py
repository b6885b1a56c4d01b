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

    def run(bn, mode=0):
        """bn: None | "slab" | "acc"."""
        scratch = torch.full((L.count("vqa_se_bwd_scratch", L.dt(dtype), B, HW, C, Cr),), float("nan"), device=DEV)
        dx = torch.empty_like(x)
        dw1, dw2 = torch.zeros_like(w1), torch.zeros_like(w2)
        nblk = L.count("vqa_se_bwd_blocks", L.dt(dtype), B, HW, C)
        slab = None
        if bn == "slab":
            slab = torch.full((nblk, 3, C), float("nan"), device=DEV)
        elif bn == "acc":
            slab = torch.zeros(L.count("vqa_bn_acc_words", 3, C), device=DEV, dtype=torch.int64)
        L.call("vqa_se_bwd", L.dt(dtype), dout.data_ptr(), x.data_ptr(), w1.data_ptr(), w2.data_ptr(), pooled.data_ptr(), hidden.data_ptr(),
               scale.data_ptr(), scratch.data_ptr(), dx.data_ptr(), dw1.data_ptr(), dw2.data_ptr(), B, HW, C, Cr, 1,
               y2.data_ptr() if bn else None, coef.data_ptr() if bn else None, slab.data_ptr() if bn else None, mode | int(bn == "acc"))
        torch.cuda.synchronize()
        return dx, dw1, dw2, slab
    # closed form in fp64 (models/attention_modules.py:109-136 differentiated by hand)
    xd, dd = x.double().view(B, HW, C), dout.double().view(B, HW, C)
    sd = scale.double()
    dz2r = (dd * xd).sum(1) * sd * (1 - sd)
    dhr = (dz2r @ w2.double()) * (hidden.double() > 0)
    dpoolr = dhr @ w1.double()
    dxr = ((dd * sd[:, None, :] + dpoolr[:, None, :] / HW) * (xd > 0)).view(B * HW, C)
    dw2r, dw1r = dz2r.t() @ hidden.double(), dhr.t() @ pooled.double()
    rel = lambda a_, r_: (a_.double() - r_).abs().max().item() / max(1e-6, r_.abs().max().item())
    etol = 1.2e-2 if dtype == torch.bfloat16 else 2e-5
    persample = run(None)                    # one workgroup per sample (a split form was measured slower in round 4: cnn_ops.hip)
    assert rel(persample[0], dxr) < etol and rel(persample[1], dw1r) < 1e-4 and rel(persample[2], dw2r) < 1e-4
    again = run(None)
    assert all(torch.equal(u, v) for u, v in zip(persample[:3], again[:3]))                  # fixed summation order: bit-reproducible
    # the fused BatchNorm-backward reduction changes nothing else, in either family
    dx, dw1, dw2, slab = run("slab")         # (slab mode: per-sample reduce + grid-wide apply, the fp32 schedule)
    assert torch.equal(dx, persample[0]) and torch.equal(dw1, persample[1]) and torch.equal(dw2, persample[2])
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
    # accumulator mode (what the bf16 training schedule runs: fixed-point sums, common.h), both families
    R = max(1, min(8, 512 // C))
    for fam in (persample,):
        dx3, dw13, dw23, facc = run("acc")
        assert torch.equal(dx3, fam[0]) and torch.equal(dw13, fam[1]) and torch.equal(dw23, fam[2])
        sums, flag = _acc_decode(facc, R, 3, C)
        g3 = dx3.float()
        r0, r1 = g3.sum(0), (g3 * (y2.float() - coef[2]) * coef[3]).sum(0)
        assert flag == 0
        assert (sums[0].float() - r0).abs().max().item() < tol(r0) and (sums[1].float() - r1).abs().max().item() < tol(r1)
    with pytest.raises(RuntimeError):                                       # all three BatchNorm arguments or none
        scratch = torch.empty(L.count("vqa_se_bwd_scratch", L.dt(dtype), B, HW, C, Cr), device=DEV)
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


def _bf16_step(cfg, sd, batch, dlogits_scale=1.0, **switches):
    """One bf16 forward + backward of the full model through the engine (no optimizer): (logits, flat gradient)."""
    from _pkg import pkg
    P, L = pkg(), sub("_lib")
    m = P.load_dropin().VQAModel(**cfg, compute_dtype="bf16")
    m.load_state_dict(sd)
    m = m.to(DEV).train()
    eng = m._ensure_engine()
    for k, v in switches.items():
        assert hasattr(eng, k), k
        setattr(eng, k, v)
    images, ids, mask, answers = batch
    logits, _, tape = eng.forward(images, ids, mask.float(), True, False, need_tape=True)
    B, N = logits.shape
    dl = torch.empty_like(logits); loss = torch.zeros(1, device=DEV); ws = torch.empty(B, device=DEV)
    L.call("vqa_cross_entropy", 0, logits.data_ptr(), answers.data_ptr(), loss.data_ptr(), dl.data_ptr(), None, B, N, float(dlogits_scale), None, ws.data_ptr())
    G = torch.zeros_like(m._flat)
    eng.backward(tape, dl, G)
    torch.cuda.synchronize()
    return logits, G


def test_bf16_step_with_and_without_the_round4_kernels():
    """hold the bf16 whole-model step against THE SAME ENGINE with a fusion / kernel switched off, where the bound
    can be tight instead of the bf16-vs-fp32 noise floor.
      * fuse_bn_conv (stage-1 conv2 and its weight gradient on the un-materialised relu(bn1(y1))) is bit-identical at kernel level, so the
        whole gradient vector must be torch.equal with it on and off;
      * use_conv8p swaps the GEMM tile of the stage-3/4 convs: same operands, fp32 sums in another order -> bf16 re-rounding of a few
        elements, which train-mode BatchNorm at B = 16 amplifies: measured 7.9 % of the gradient norm between the two valid bf16
        schedules; bound 15 % (the bf16-vs-fp32 bounds of test_gpu_model.py are 45-75 %; a wiring error -- wrong tap, mirrored weights,
        missing addend -- is O(1)), logits within 2e-2."""
    from oracle import vqa_oracle as O
    cfg = O.full_config(dropout=0.0, answer_dropout=0.0)
    sd = O.init_state_dict(cfg, 5, jitter=True)
    batch = [t.to(DEV) for t in O.synthetic_batch(16, seed=77)]
    l_ref, g_ref = _bf16_step(cfg, sd, batch)
    l_a, g_a = _bf16_step(cfg, sd, batch, fuse_bn_conv=False)
    assert torch.equal(l_a, l_ref) and torch.equal(g_a, g_ref)
    l_b, g_b = _bf16_step(cfg, sd, batch, use_conv8p=False)
    assert (l_b - l_ref).abs().max().item() < 2e-2
    assert float((g_b - g_ref).norm() / g_ref.norm()) < 0.15
    # fuse_bn1_reduce: bn1's backward column sums taken in conv8p's data-gradient epilogue instead of a pass over (da1, y1): same
    # elements, same mask, fp32 partials grouped per tile instead of per row block -> the forward is untouched, the sums agree to
    # ~1e-6 relative and only re-rounded bf16 elements downstream can differ
    l_d, g_d = _bf16_step(cfg, sd, batch, fuse_bn1_reduce=False)
    assert torch.equal(l_d, l_ref)
    assert float((g_d - g_ref).norm() / g_ref.norm()) < 0.03
    # fuse_hand_reduce: the first block's bn2 (+ shortcut BatchNorm) backward sums taken in the epilogue of the second block's conv1 data
    # gradient (stages 2-4) instead of a bn_bwd_reduce pass: forward untouched, sums equal up to the grouping of fp32 partials
    l_f, g_f = _bf16_step(cfg, sd, batch, fuse_hand_reduce=True)      # (off by default: measured neutral, engine.py)
    assert torch.equal(l_f, l_ref)
    assert float((g_f - g_ref).norm() / g_ref.norm()) < 0.03
    # use_c64p_epi: stage 1's conv1 data gradients on the patch kernel instead of the 128 x 64 igemm tile (other summation order, like
    # use_conv8p but backward only)
    l_e, g_e = _bf16_step(cfg, sd, batch, use_c64p_epi=False)
    assert torch.equal(l_e, l_ref)
    assert float((g_e - g_ref).norm() / g_ref.norm()) < 0.15
    # hoist_cross only moves launches between streams (layer 0's query projection beside the CNN, layer 1's K / V path forward and
    # backward on the text stream): the same kernels on the same operands -> bit-identical, or an ordering event is missing
    l_h, g_h = _bf16_step(cfg, sd, batch, hoist_cross=False)
    assert torch.equal(l_h, l_ref) and torch.equal(g_h, g_ref)
    l_c, g_c = _bf16_step(cfg, sd, batch)                            # and the step itself is bit-reproducible
    assert torch.equal(l_c, l_ref) and torch.equal(g_c, g_ref)


@pytest.mark.parametrize("dropout", [0.0, 0.1])
def test_bf16_step_with_and_without_the_activation_mask_in_the_dgrad_epilogue(dropout):
    """fuse_act_dgrad (round 4): the ReLU(+dropout) backward between two Linears (4 text FFNs, 2 cross-attention FFNs, 2 answer-head
    pairs) applied by the data-gradient GEMM's epilogue, the bias column sums taken off the chain.  Same logits; every gradient is
    bit-equal except the 8 bias gradients in front of those activations, which are now the column sums of the STORED (bf16) gradient
    instead of the fp32 value before its rounding: identical without dropout (keep scale 1), within 2e-3 of their norm with it."""
    from oracle import vqa_oracle as O
    from _pkg import sub as _sub
    cfg = O.full_config(dropout=dropout, answer_dropout=0.3 if dropout else 0.0)
    sd = O.init_state_dict(cfg, 6, jitter=True)
    batch = [t.to(DEV) for t in O.synthetic_batch(16, seed=78)]
    l_ref, g_ref = _bf16_step(cfg, sd, batch)
    l_a, g_a = _bf16_step(cfg, sd, batch, fuse_act_dgrad=False)
    assert torch.equal(l_a, l_ref)
    if dropout == 0.0:
        assert torch.equal(g_a, g_ref)
        return
    LY = _sub("layout")
    entries = LY.build_entries(cfg)
    moved = 0
    for e in entries:
        a, r = g_a[e.offset: e.offset + e.numel], g_ref[e.offset: e.offset + e.numel]
        if torch.equal(a, r):
            continue
        moved += 1
        assert e.name.endswith(".bias") and e.name not in ("answer_head.classifier.6.bias",), e.name
        assert float((a - r).norm() / r.norm()) < 2e-3, e.name
    assert 1 <= moved <= 8, moved


@pytest.mark.parametrize("log2_scale", [16, 24])
def test_bf16_backward_is_linear_in_the_loss_scale(log2_scale):
    """The whole bf16 backward under a power-of-two loss scale (GradScaler's 2^16 ... ): every bf16 rounding and every fixed-point sum is
    scale-invariant for powers of two (no overflow, no subnormals at these magnitudes), so G(scale * dlogits) must equal scale * G(dlogits)
    -- to a few ulp where a 2^-50 fixed-point plane rounds differently.  A saturating / wrapping accumulator or a mis-scaled epilogue
    shows up as a relative error of O(1) in some tensor."""
    from oracle import vqa_oracle as O
    cfg = O.full_config(dropout=0.0, answer_dropout=0.0)
    sd = O.init_state_dict(cfg, 6, jitter=True)
    batch = [t.to(DEV) for t in O.synthetic_batch(8, seed=78)]
    _, g1 = _bf16_step(cfg, sd, batch)
    s = float(2 ** log2_scale)
    _, gs = _bf16_step(cfg, sd, batch, dlogits_scale=s)
    assert torch.isfinite(gs).all()
    rel = float((gs / s - g1).norm() / g1.norm())
    assert rel < 1e-3, rel
    frac_exact = float((gs / s == g1).float().mean())
    assert frac_exact > 0.9, frac_exact


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


@pytest.mark.parametrize("B,H,W", [(6, 56, 56), (3, 24, 24), (2, 96, 96), (1, 8, 8), (5, 16, 40)])
def test_conv_on_unmaterialised_bn_relu_equals_bn_apply_then_conv(B, H, W):
    """VERDICT r3 #2 (models/cnn_backbone.py:182-187, training mode): vqa_conv3x3_c64p_bn(y1, statistics) == vqa_conv3x3_c64p(
    vqa_bn_apply_acc(y1, statistics)) BIT FOR BIT -- output, its BatchNorm statistics, the published coefficients, the running
    statistics -- and vqa_wgrad3x3_c64_bn(y1, coef, dy) == vqa_wgrad3x3_c64(a1, dy).  y1 has a large negative mean in a few channels
    so that relu(shift) != 0 there: a padding pixel that went through the transform instead of staying zero would show."""
    K, L = sub("kernels"), sub("_lib")
    if K.c64p_blocks(B, H, W) <= 0 or not K.c64w_bn_ok(B, H, W):
        pytest.skip("shape not taken by the 8-wave patch kernels")
    g = torch.Generator().manual_seed(B * 1000 + H + W)
    bf = torch.bfloat16
    x = torch.randn(B * H * W, 64, generator=g).to(DEV, bf)
    w1 = (torch.randn(64, 576, generator=g) * 0.05).to(DEV, bf)
    w2 = (torch.randn(64, 576, generator=g) * 0.05).to(DEV, bf)
    gamma = (torch.rand(64, generator=g) + 0.5).to(DEV)
    beta = (torch.randn(64, generator=g) * 0.5 + 0.3).to(DEV)          # mostly positive shifts: relu(shift) > 0
    words = L.count("vqa_bn_acc_words", 2, 64)

    def producer():
        acc = torch.zeros(words, device=DEV, dtype=torch.int64)
        y1, st, _ = K.conv3x3_c64p(x, w1, B, H, W, want_stats=True, stats_acc=acc)
        return y1, acc
    bnp = lambda: (gamma, beta, torch.zeros(64, device=DEV), torch.ones(64, device=DEV), torch.zeros((), device=DEV, dtype=torch.int64))
    # reference: bn_apply_acc -> conv
    y1, acc1 = producer()
    bn_a = bnp()
    a1, c_ref, _, _, _ = K.bn_apply_acc(y1, acc1, bn_a, 64, True, B, H * W, B * H * W)
    accr = torch.zeros(words, device=DEV, dtype=torch.int64)
    y2_ref, _, _ = K.conv3x3_c64p(a1, w2, B, H, W, want_stats=True, stats_acc=accr)
    # fused
    y1b, acc1b = producer()
    assert torch.equal(y1, y1b) and torch.equal(acc1, acc1b)
    bn_b = bnp()
    accf = torch.zeros(words, device=DEV, dtype=torch.int64)
    y2, st, _, coef = K.conv3x3_c64p_bn(y1b, acc1b, bn_b, w2, B, H, W, B * H * W, want_stats=True, stats_acc=accf)
    torch.cuda.synchronize()
    assert float(a1.float().min()) == 0.0 and float((a1 == 0).float().mean()) > 0.05       # the ReLU bites
    assert torch.equal(coef, c_ref)
    assert torch.equal(bn_a[2], bn_b[2]) and torch.equal(bn_a[3], bn_b[3]) and int(bn_b[4]) == 1          # running statistics, counter
    assert torch.equal(y2, y2_ref), float((y2.float() - y2_ref.float()).abs().max())
    assert torch.equal(accf, accr)
    # against ATen on the bf16 operands (the conv itself)
    ref = torch.nn.functional.conv2d(a1.float().view(B, H, W, 64).permute(0, 3, 1, 2).cpu(), w2.float().view(64, 3, 3, 64).permute(0, 3, 1, 2).cpu(), padding=1)
    got = y2.float().view(B, H, W, 64).permute(0, 3, 1, 2).cpu()
    assert (got - ref).abs().max().item() < 1e-2 * max(1.0, ref.abs().max().item())
    # weight gradient
    dy = torch.randn(B * H * W, 64, generator=g).to(DEV, bf)
    dw_ref, dw = torch.zeros(64, 576, device=DEV), torch.zeros(64, 576, device=DEV)
    K.wgrad3x3_c64(a1, dy, dw_ref, B, H, W)
    K.wgrad3x3_c64_bn(y1, coef, dy, dw, B, H, W)
    torch.cuda.synchronize()
    assert torch.equal(dw, dw_ref), float((dw - dw_ref).abs().max())
    # slab statistics mode + a second run (bit-reproducible)
    y2s, slab, nb, _ = K.conv3x3_c64p_bn(y1, acc1, bnp(), w2, B, H, W, B * H * W, want_stats=True)
    y2r, slabr, _ = K.conv3x3_c64p(a1, w2, B, H, W, want_stats=True)
    assert torch.equal(y2s, y2_ref) and torch.equal(slab, slabr)


def _acc_decode(acc, R, K, C):
    """common.h layout: hi plane [R][K][C] (units of 2^-4) | flag | lo plane [R][K][C] (units of 2^-50)."""
    n = R * K * C
    hi = acc[:n].view(R, K, C).sum(0).double() / 16.0
    lo = acc[n + 1: 2 * n + 1].view(R, K, C).sum(0).double() / float(1 << 50)
    return hi + lo, int(acc[n])


@pytest.mark.parametrize("C,rows", [(64, 6 * 56 * 56), (256, 8 * 196)])
@pytest.mark.parametrize("log2_scale", [0, 16, 24, 30, 36, 44])
def test_batchnorm_backward_sums_are_scale_times_the_unscaled_ones_or_non_finite(C, rows, log2_scale):
    """under GradScaler the gradients entering the BatchNorm backward are multiplied by a loss scale that starts
    at 2^16 and doubles every 2000 clean steps (training/train.py:179-195).  The fixed-point sums must either be exactly
    scale x the unscaled sums (power-of-two scale: every partial scales exactly) or raise the flag so that dy / d gamma / d beta
    turn NaN -- never finite and wrong (round 3's single 2^-40 plane wrapped its int64 total from ~96 partials near 2^22)."""
    K, L = sub("kernels"), sub("_lib")
    g = torch.Generator().manual_seed(C + log2_scale)
    # same-sign gradients with a large mean: the worst case for a wrapping total (every partial has the same sign)
    dout = (torch.rand(rows, C, generator=g) * 4 + 1).to(DEV)
    y = torch.randn(rows, C, generator=g).to(DEV)
    coef = _coef(C, g).to(DEV)
    gamma = (torch.rand(C, generator=g) + 0.5).to(DEV)

    def run(scale):
        dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        facc = torch.zeros(L.count("vqa_bn_acc_words", 3, C), device=DEV, dtype=torch.int64)
        dy, _ = K.bn_bwd(dout * scale, None, y, coef, gamma, C, True, dg, db, facc=facc)
        torch.cuda.synchronize()
        return dy, dg, db, facc
    dy1, dg1, db1, _ = run(1.0)
    s = float(2 ** log2_scale)
    dy, dg, db, facc = run(s)
    R = max(1, min(8, 512 // C))
    sums, flag = _acc_decode(facc, R, 3, C)
    if flag:
        assert torch.isnan(dy).all() and torch.isnan(dg).all() and torch.isnan(db).all()      # loud: GradScaler skips and backs off
        assert log2_scale >= 30                                                        # and only far beyond what fp16 AMP ever reaches
    else:
        assert log2_scale <= 36
        ref0 = (dout.double() * s).sum(0)
        assert (sums[0] - ref0).abs().max().item() <= 1e-6 * float(ref0.abs().max())  # the TOTAL did not wrap
        assert torch.equal(db, db1 * s) and torch.equal(dg, dg1 * s)                   # exact: power-of-two scaling of every partial
        assert (dy - dy1 * s).abs().max().item() <= 1e-5 * float((dy1 * s).abs().max())


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
