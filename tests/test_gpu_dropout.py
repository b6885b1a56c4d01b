"""GPU: dropout, which is ON in the benchmarked configuration (28 nn.Dropout modules in the reference: text_encoder.py:510,392,397,
fusion.py:108, cross_attention.py, vqa_model.py:76,80).  Bitwise parity with torch's Philox stream is not a goal;
what must hold, per site with its own code path, is torch's dropout CONTRACT:
  keep-rate 1-p (within 3 sigma on >= 1e6 elements), kept values scaled by 1/(1-p), dropped values exactly 0,
  same seed => same mask, different seed => different mask, and the backward regenerates the forward's mask
  (the gradient is zero exactly where the forward output was dropped).
The attention site exposes no elementwise output, so the counter-based generator (csrc/common.h: mix32 / drop_key /
drop_keep32) is restated here in numpy and both attention kernels (MFMA bf16, VALU fp32) are checked against the torch formula
with that mask, forward and backward."""
import math

import numpy as np
import pytest
import torch

from _pkg import pkg, sub
from oracle import vqa_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
M32 = np.uint64(0xFFFFFFFF)


def _mix32(x):
    x = x & M32
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7feb352d)) & M32
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x846ca68b)) & M32
    x ^= x >> np.uint64(16)
    return x


def keep_mask(seed: int, n: int, p: float) -> np.ndarray:
    """numpy restatement of drop_keep32(drop_key(seed), idx, p) for idx in [0, n)."""
    key = (_mix32(np.uint64(seed & 0xFFFFFFFF)) ^ ((np.uint64(seed >> 32) * np.uint64(0x9E3779B9)) & M32)) & M32
    h = _mix32(np.arange(n, dtype=np.uint64) ^ key)
    u = (h >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return u >= np.float32(p)


def _rate_ok(kept: int, n: int, p: float):
    sigma = math.sqrt(n * p * (1 - p))
    assert abs(kept - n * (1 - p)) < 3 * sigma + 1, (kept / n, 1 - p)


@pytest.mark.parametrize("p", [0.1, 0.3])
def test_linear_epilogue_dropout_and_its_backward(p):
    """igemm epilogue (FFN / attention-output / answer-head sites) + vqa_bias_act_bwd."""
    K, L = sub("kernels"), sub("_lib")
    M, Kin, N = 4096, 64, 256                                        # 1 048 576 outputs
    g = torch.Generator().manual_seed(1)
    x = torch.randn(M, Kin, generator=g).to(DEV)
    w = (torch.randn(N, Kin, generator=g) / 8).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    geom = K.linear_geom(M, Kin)
    base, _, _ = K.igemm(x, w, M, N, Kin, geom, dtype=torch.float32, bias=b)
    d1, _, _ = K.igemm(x, w, M, N, Kin, geom, dtype=torch.float32, bias=b, drop_p=p, drop_seed=1234)
    d1b, _, _ = K.igemm(x, w, M, N, Kin, geom, dtype=torch.float32, bias=b, drop_p=p, drop_seed=1234)
    d2, _, _ = K.igemm(x, w, M, N, Kin, geom, dtype=torch.float32, bias=b, drop_p=p, drop_seed=1235)
    torch.cuda.synchronize()
    assert (base != 0).all()
    kept = d1 != 0
    _rate_ok(int(kept.sum()), M * N, p)
    assert torch.allclose(d1[kept], base[kept] / (1 - p), rtol=1e-6, atol=0)
    assert torch.equal(d1, d1b)
    assert not torch.equal(d1 != 0, d2 != 0)
    assert np.array_equal(kept.cpu().numpy().reshape(-1), keep_mask(1234, M * N, p))      # the documented generator, idx = m*N + n
    # backward: dz = dout * keep / (1-p), zero exactly where the forward dropped; bias gradient = column sums of dz
    dout = torch.randn(M, N, generator=g).to(DEV)
    dz = torch.empty_like(dout)
    dbias = torch.zeros(N, device=DEV)
    L.call("vqa_bias_act_bwd", 0, dout.data_ptr(), None, dz.data_ptr(), dbias.data_ptr(), M, N, float(p), 1234, None, 0)
    torch.cuda.synchronize()
    assert torch.equal(dz != 0, kept & (dout != 0))
    assert torch.allclose(dz[kept], dout[kept] / (1 - p), rtol=1e-6)
    assert torch.allclose(dbias, dz.sum(0), rtol=1e-4, atol=1e-3)
    # ReLU + dropout (FFN inner / answer head): out > 0 encodes both masks for the backward
    r1, _, _ = K.igemm(x, w, M, N, Kin, geom, dtype=torch.float32, bias=b, relu=1, drop_p=p, drop_seed=77)
    rbase, _, _ = K.igemm(x, w, M, N, Kin, geom, dtype=torch.float32, bias=b, relu=1)
    dz2 = torch.empty_like(dout)
    L.call("vqa_bias_act_bwd", 0, dout.data_ptr(), r1.data_ptr(), dz2.data_ptr(), None, M, N, float(p), 77, None, 0)
    torch.cuda.synchronize()
    pos = rbase > 0
    _rate_ok(int((r1 > 0).sum()), int(pos.sum()), p)
    assert not (r1[~pos] != 0).any()
    assert torch.allclose(r1[r1 > 0], rbase[r1 > 0] / (1 - p), rtol=1e-6)
    assert torch.equal(dz2 != 0, (r1 > 0) & (dout != 0))
    assert torch.allclose(dz2[r1 > 0], dout[r1 > 0] / (1 - p), rtol=1e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_layernorm_fused_dropout_and_its_backward(dtype):
    """Projector site (models/fusion.py:105-110): LayerNorm -> Dropout -> + position embedding (added AFTER the dropout)."""
    K = sub("kernels")
    rows, D, period, p = 49 * 96, 256, 49, 0.1                      # 1 204 224 elements
    g = torch.Generator().manual_seed(2)
    x = torch.randn(rows, D, generator=g).to(dtype)
    gam, bet = torch.rand(D, generator=g) + 0.5, torch.randn(D, generator=g) + 3.0     # beta != 0: LN output never exactly 0
    pos = torch.randn(period, D, generator=g)
    xd, gd, bd, posd = x.to(DEV), gam.to(DEV), bet.to(DEV), pos.to(DEV)
    plain, _ = K.layernorm_fwd(xd, gd, bd)
    out, st = K.layernorm_fwd(xd, gd, bd, drop_p=p, seed=99, addrow=posd, period=period)
    out2, _ = K.layernorm_fwd(xd, gd, bd, drop_p=p, seed=99, addrow=posd, period=period)
    out3, _ = K.layernorm_fwd(xd, gd, bd, drop_p=p, seed=100, addrow=posd, period=period)
    torch.cuda.synchronize()
    keep = torch.from_numpy(keep_mask(99, rows * D, p)).view(rows, D)
    _rate_ok(int(keep.sum()), rows * D, p)
    ref = plain.float().cpu() * keep / (1 - p) + pos.repeat(rows // period, 1)
    tol = 1e-5 if dtype == torch.float32 else 1.5e-2 * float(ref.abs().max())      # bf16: `plain` is itself rounded to 8 bits
    assert (out.float().cpu() - ref).abs().max().item() < tol
    if dtype == torch.float32:
        assert torch.equal((out.cpu() - pos.repeat(rows // period, 1)) != 0, keep)       # dropped: exactly the addend
    assert torch.equal(out, out2) and not torch.equal(out, out3)
    # backward with the same seed == autograd through  LN(x) * mask/(1-p) + pos
    dout = torch.randn(rows, D, generator=g).to(dtype)
    xr = x.float().requires_grad_(True); gr = gam.clone().requires_grad_(True); br = bet.clone().requires_grad_(True)
    pr = pos.clone().requires_grad_(True)
    y = torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-5) * keep / (1 - p) + pr.repeat(rows // period, 1)
    y.backward(dout.float())
    dgam, dbet, dpos = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV), torch.zeros(period, D, device=DEV)
    dx = K.layernorm_bwd(dout.to(DEV), xd, gd, st, dgam, dbet, drop_p=p, seed=99, dadd=dpos, period=period)
    torch.cuda.synchronize()
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    bound = 2e-4 if dtype == torch.float32 else 1.5e-2
    assert rel(dx.float().cpu(), xr.grad) < bound
    assert rel(dgam.cpu(), gr.grad) < bound and rel(dbet.cpu(), br.grad) < bound and rel(dpos.cpu(), pr.grad) < bound


def test_embedding_dropout_and_its_backward():
    """PositionalEncoding dropout (models/text_encoder.py:112-114 after :504-510): (emb[id]*sqrt(d) + pe) -> Dropout."""
    L = sub("_lib")
    B, Lq, D, V, p = 2048, 20, 32, 500, 0.1                          # 1 310 720 elements
    rows = B * Lq
    g = torch.Generator().manual_seed(4)
    ids = torch.randint(0, V, (B, Lq), generator=g)
    emb = torch.randn(V, D, generator=g); emb[0] = 0
    pe = O.sinusoid_pe(Lq, D)[0] + 5.0                              # never exactly zero
    out = torch.empty(rows, D, device=DEV)
    idd, embd, ped = ids.to(DEV), emb.to(DEV), pe.to(DEV).contiguous()
    L.call("vqa_embed_fwd", 0, idd.data_ptr(), embd.data_ptr(), ped.data_ptr(), out.data_ptr(), rows, Lq, D, V, math.sqrt(D), p, 4242)
    torch.cuda.synchronize()
    keep = torch.from_numpy(keep_mask(4242, rows * D, p)).view(rows, D)
    _rate_ok(int(keep.sum()), rows * D, p)
    ref = (emb[ids.view(-1)] * math.sqrt(D) + pe.repeat(B, 1)) * keep / (1 - p)
    assert torch.equal(out.cpu() != 0, keep)
    assert torch.allclose(out.cpu(), ref, rtol=1e-5, atol=1e-6)
    dout = torch.randn(rows, D, generator=g)
    er = emb.clone().requires_grad_(True)
    ((torch.nn.functional.embedding(ids.view(-1), er, padding_idx=0) * math.sqrt(D) + pe.repeat(B, 1)) * keep / (1 - p)).backward(dout)
    demb = torch.zeros(V, D, device=DEV)
    dd = dout.to(DEV)
    L.call("vqa_embed_bwd", 0, idd.data_ptr(), dd.data_ptr(), demb.data_ptr(), rows, D, V, math.sqrt(D), p, 4242)
    torch.cuda.synchronize()
    assert float((demb.cpu() - er.grad).abs().max() / er.grad.abs().max()) < 1e-4
    assert (demb[0] == 0).all()                                      # padding_idx row receives no gradient


def _attn_ref(q, k, v, kmask, keep, p, heads):
    """softmax(QK^T/sqrt(hd) masked) * keep/(1-p) @ V, per head; q [B,Lq,d], k/v [B,Lk,d]."""
    B, Lq, d = q.shape
    Lk, hd = k.shape[1], d // heads
    Q = q.view(B, Lq, heads, hd).transpose(1, 2); Kt = k.view(B, Lk, heads, hd).transpose(1, 2); V = v.view(B, Lk, heads, hd).transpose(1, 2)
    s = Q @ Kt.transpose(-1, -2) / math.sqrt(hd)
    if kmask is not None:
        s = s.masked_fill(kmask[:, None, None, :] == 0, float("-inf"))
    pr = torch.softmax(s, -1)
    ctx = (pr * keep / (1 - p)) @ V
    return ctx.transpose(1, 2).reshape(B, Lq, d), pr


@pytest.mark.parametrize("kind,Lk", [("mfma", 20), ("mfma", 49), ("mfma", 144), ("valu", 20), ("valu", 49)])
def test_attention_dropout_forward_and_backward_share_the_mask(kind, Lk):
    """Attention-probability dropout (text_encoder.py:247-248, cross_attention.py:184-185) in the MFMA (bf16) and VALU (fp32)
    kernels: forward context and dQ/dK/dV equal the torch formula evaluated with the generator's mask (idx = ((b*H+h)*Lq+r)*Lk+c)."""
    L = sub("_lib")
    B, H, Lq, hd, p, seed = 64, 8, 20, 32, 0.1, 31337
    d = H * hd
    dtype = torch.bfloat16 if kind == "mfma" else torch.float32
    g = torch.Generator().manual_seed(Lk)
    q = torch.randn(B, Lq, d, generator=g).to(dtype); k = torch.randn(B, Lk, d, generator=g).to(dtype); v = torch.randn(B, Lk, d, generator=g).to(dtype)
    dctx = torch.randn(B, Lq, d, generator=g).to(dtype)
    kmask = None
    if Lk == 20:
        lens = torch.randint(5, 21, (B,), generator=g)
        kmask = (torch.arange(Lk)[None] < lens[:, None]).float()
    keep = torch.from_numpy(keep_mask(seed, B * H * Lq * Lk, p)).view(B, H, Lq, Lk)
    _rate_ok(int(keep_mask(seed, 1 << 20, p).sum()), 1 << 20, p)
    qr, kr, vr = (t.float().requires_grad_(True) for t in (q, k, v))
    ctx_ref, pr_ref = _attn_ref(qr, kr, vr, kmask, keep, p, H)
    ctx_ref.backward(dctx.float())
    qd, kd, vd, dcd = (t.reshape(-1, d).to(DEV) for t in (q, k, v, dctx))
    md = None if kmask is None else kmask.to(DEV)
    probs = torch.empty(B, H, Lq, Lk, device=DEV)
    ctx = torch.empty(B * Lq, d, device=DEV, dtype=dtype)
    dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
    mp = None if md is None else md.data_ptr()
    if kind == "mfma":
        L.call("vqa_attention_fwd_mfma", qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), d, d, d, mp, probs.data_ptr(), ctx.data_ptr(), d,
               B, H, Lq, Lk, hd, p, seed)
        L.call("vqa_attention_bwd_mfma", dcd.data_ptr(), d, qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), d, d, d, probs.data_ptr(),
               dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), d, d, d, B, H, Lq, Lk, hd, p, seed)
    else:
        L.call("vqa_attention_fwd", 0, qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), d, d, d, mp, probs.data_ptr(), ctx.data_ptr(), d,
               B, H, Lq, Lk, hd, p, seed)
        L.call("vqa_attention_bwd", 0, dcd.data_ptr(), d, qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), d, d, d, probs.data_ptr(),
               dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), d, d, d, B, H, Lq, Lk, hd, p, seed)
    torch.cuda.synchronize()
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    tol = 2e-2 if kind == "mfma" else 1e-4
    assert (probs.cpu() - pr_ref.detach()).abs().max().item() < (4e-3 if kind == "mfma" else 1e-5)      # probs are PRE-dropout
    assert rel(ctx.float().cpu().view(B, Lq, d), ctx_ref.detach()) < tol
    assert rel(dq.float().cpu().view(B, Lq, d), qr.grad) < tol
    assert rel(dk.float().cpu().view(B, Lk, d), kr.grad) < tol
    assert rel(dv.float().cpu().view(B, Lk, d), vr.grad) < tol
    # a mask at the wrong index would be far outside these bounds: with an independent mask the context error is O(1)
    wrong, _ = _attn_ref(q.float(), k.float(), v.float(), kmask, torch.from_numpy(keep_mask(seed + 1, B * H * Lq * Lk, p)).view(B, H, Lq, Lk), p, H)
    assert rel(wrong, ctx_ref.detach()) > 10 * tol


def test_model_draws_fresh_masks_every_training_forward():
    """The reference's unchanged loop (model(...) -> loss.backward() -> optimizer.step(), training/train.py:176-208) calls only
    the module: every training forward must advance the dropout stream by itself, and a backward must reuse ITS forward's masks."""
    cfg = O.full_config()
    sd = O.init_state_dict(cfg, 5)
    m = pkg().load_dropin().VQAModel(**cfg, compute_dtype="bf16")
    m.load_state_dict(sd)
    m = m.to(DEV).train()
    images, ids, mask, answers = (t.to(DEV) for t in O.synthetic_batch(4, seed=9))
    a, _ = m(images, ids, mask)
    sid = m._engine.step_id
    b, _ = m(images, ids, mask)
    torch.cuda.synchronize()
    assert torch.isfinite(a).all() and torch.isfinite(b).all()
    assert m._engine.step_id == sid + 1
    assert not torch.equal(a, b)                                     # same weights, same batch: only the dropout masks differ
    # the tape of the LAST forward carries its own seeds: backward after more forwards of OTHER models/steps is unaffected
    seeds_b = [m._tapes[m._tape_seq]["head"]["s1"], m._tapes[m._tape_seq]["proj"]["seed"]]
    assert all((s >> 12) & 0xFFFFFFFF == m._engine.seed_base + m._engine.step_id and s >> 44 == 0 for s in seeds_b)
    torch.nn.functional.cross_entropy(b, answers).backward()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters())
    # eval forwards do not consume the stream and are deterministic
    m.eval()
    with torch.no_grad():
        e1, _ = m(images, ids, mask); e2, _ = m(images, ids, mask)
    assert torch.equal(e1, e2) and m._engine.step_id == sid + 1
    # two models with the same seed_base and step draw the same masks (replay); a different rank offset changes them
    m2 = pkg().load_dropin().VQAModel(**cfg, compute_dtype="bf16")
    m2.load_state_dict(sd)
    m2 = m2.to(DEV).train()
    a2, _ = m2(images, ids, mask)
    assert torch.equal(a, a2)
    m3 = pkg().load_dropin().VQAModel(**cfg, compute_dtype="bf16")
    m3.load_state_dict(sd)
    m3 = m3.to(DEV).train()
    m3._ensure_engine().seed_rank = 1                                # what the second data-parallel rank gets (engine.py: rank << 44)
    a3, _ = m3(images, ids, mask)
    assert not torch.equal(a, a3)
