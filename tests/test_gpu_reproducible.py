"""GPU: the train step is bit-reproducible run to run.

The reference runs on the CPU, where every gradient is a fixed-order sum; the HIP path reaches the same property without
float atomics: GEMM-shaped weight gradients go through per-split slabs + a fixed-order reduce, the small reductions (LayerNorm
gamma/beta, position-embedding, bias, loss) through per-workgroup partial rows + fold_rows_kernel (csrc/token_ops.hip: rows
summed in index order by a second launch of the same C entry), the embedding table through a per-row gather, SE / spatial-
attention weights through single-writer kernels.  Checked here:
  * the fixed-order kernels against torch (values) and against themselves (bits) on shapes with many workgroups,
  * two independent runs of the whole step (dropout on, both side streams on): gradient buffer, loss, parameters after three
    steps and BatchNorm buffers are torch.equal."""
import math

import pytest
import torch

from _pkg import pkg, sub
from oracle import vqa_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,D,period", [(10240, 256, 20), (25088, 256, 49), (333, 96, 0)])
def test_layernorm_backward_fixed_order(dtype, rows, D, period):
    K = sub("kernels")
    g = torch.Generator().manual_seed(rows + D)
    x = torch.randn(rows, D, generator=g).to(dtype)
    gam, bet = torch.rand(D, generator=g) + 0.5, torch.randn(D, generator=g)
    dout = torch.randn(rows, D, generator=g).to(dtype)
    xd, gd, bd, dd = x.to(DEV), gam.to(DEV), bet.to(DEV), dout.to(DEV)
    _, st = K.layernorm_fwd(xd, gd, bd)
    xr = x.float().requires_grad_(True); gr = gam.clone().requires_grad_(True); br = bet.clone().requires_grad_(True)
    y = torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-5)
    pr = None
    if period:
        pr = torch.zeros(period, D, requires_grad=True)
        y = y + pr.repeat(rows // period, 1)
    y.backward(dout.float())
    outs = []
    for rep in range(3):
        dgam, dbet = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
        dpos = torch.zeros(period, D, device=DEV) if period else None
        dx = K.layernorm_bwd(dd, xd, gd, st, dgam, dbet, dadd=dpos, period=max(period, 1))
        outs.append((dx, dgam, dbet, dpos))
    torch.cuda.synchronize()
    dx, dgam, dbet, dpos = outs[0]
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    bound = 2e-4 if dtype == torch.float32 else 1.5e-2
    assert rel(dx.float().cpu(), xr.grad) < bound
    assert rel(dgam.cpu(), gr.grad) < bound and rel(dbet.cpu(), br.grad) < bound
    if period:
        assert rel(dpos.cpu(), pr.grad) < bound
    for o in outs[1:]:
        assert torch.equal(o[1], dgam) and torch.equal(o[2], dbet) and torch.equal(o[0], dx)
        if period:
            assert torch.equal(o[3], dpos)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N", [(10240, 1024), (25088, 256), (512, 1000), (77, 12)])
def test_bias_gradient_fixed_order(dtype, M, N):
    L, K = sub("_lib"), sub("kernels")
    g = torch.Generator().manual_seed(M + N)
    dout = torch.randn(M, N, generator=g).to(dtype)
    act = torch.relu(torch.randn(M, N, generator=g)).to(dtype)
    ref = (dout.float() * (act.float() > 0)).sum(0)
    dd, ad = dout.to(DEV), act.to(DEV)
    wsf = K.reduce_ws("vqa_bias_act_bwd_ws", L.dt(dtype), M, N)
    res = []
    for rep in range(3):
        ws = torch.empty(wsf, device=DEV)
        db = torch.zeros(N, device=DEV)
        dz = torch.empty_like(dd)
        L.call("vqa_bias_act_bwd", L.dt(dtype), dd.data_ptr(), ad.data_ptr(), dz.data_ptr(), db.data_ptr(), M, N, 0.0, 0,
               ws.data_ptr(), 0)
        res.append((db, dz))
    torch.cuda.synchronize()
    assert float((res[0][0].cpu() - ref).abs().max()) < 1e-3 * max(1.0, float(ref.abs().max()))
    assert torch.equal(res[0][1].float().cpu(), dout.float() * (act.float() > 0))
    assert torch.equal(res[1][0], res[0][0]) and torch.equal(res[2][0], res[0][0])
    # without scratch: float atomics, same value up to rounding order
    db2 = torch.zeros(N, device=DEV)
    L.call("vqa_bias_act_bwd", L.dt(dtype), dd.data_ptr(), ad.data_ptr(), dz.data_ptr(), db2.data_ptr(), M, N, 0.0, 0, None, 0)
    torch.cuda.synchronize()
    assert float((db2 - res[0][0]).abs().max()) < 1e-3 * max(1.0, float(ref.abs().max()))


def test_deferred_folds_equal_immediate_folds():
    """defer_fold = 1 + one vqa_fold_group launch for several LayerNorm / bias backward calls gives the same bits as each entry's own fold."""
    L, K = sub("_lib"), sub("kernels")
    T = torch.bfloat16
    g = torch.Generator().manual_seed(5)
    q, want = [], []
    for rows, D, period in ((10240, 256, 0), (25088, 256, 49), (512, 256, 0)):
        x = torch.randn(rows, D, generator=g).to(T).to(DEV)
        gam, bet = (torch.rand(D, generator=g) + 0.5).to(DEV), torch.randn(D, generator=g).to(DEV)
        dout = torch.randn(rows, D, generator=g).to(T).to(DEV)
        _, st = K.layernorm_fwd(x, gam, bet)
        res = []
        for fq in (None, q):
            dgam, dbet = torch.full((D,), 0.25, device=DEV), torch.zeros(D, device=DEV)
            dpos = torch.zeros(period, D, device=DEV) if period else None
            K.layernorm_bwd(dout, x, gam, st, dgam, dbet, dadd=dpos, period=max(period, 1), foldq=fq)
            res.append((dgam, dbet, dpos))
        want.append(res)
    for M, N in ((10240, 1024), (512, 1000)):
        dout = torch.randn(M, N, generator=g).to(T).to(DEV)
        wsf = K.reduce_ws("vqa_bias_act_bwd_ws", L.dt(T), M, N)
        res = []
        for defer in (0, 1):
            ws, db = torch.empty(wsf, device=DEV), torch.full((N,), -1.0, device=DEV)
            L.call("vqa_bias_act_bwd", L.dt(T), dout.data_ptr(), None, None, db.data_ptr(), M, N, 0.0, 0, ws.data_ptr(), defer)
            if defer:
                q.append((ws, 0, L.count("vqa_bias_act_bwd_fold_rows", L.dt(T), M, N), N, N, db, N, None))
            res.append((db,))
        want.append(res)
    assert len(q) == 3 + 1 + 2
    K.fold_group(q)
    torch.cuda.synchronize()
    for res in want:
        for a, b in zip(res[0], res[1]):
            if a is not None:
                assert torch.equal(a, b) and float(a.abs().max()) > 0


def test_cross_entropy_loss_fixed_order():
    L = sub("_lib")
    g = torch.Generator().manual_seed(9)
    for B, N in ((512, 1000), (4099, 10), (3, 7)):
        logits = (torch.randn(B, N, generator=g) * 3).to(DEV)
        tgt = torch.randint(0, N, (B,), generator=g).to(DEV)
        ref = torch.nn.functional.cross_entropy(logits.cpu(), tgt.cpu())
        vals = []
        for rep in range(3):
            loss = torch.zeros(1, device=DEV)
            ws = torch.empty(B, device=DEV)
            L.call("vqa_cross_entropy", 0, logits.data_ptr(), tgt.data_ptr(), loss.data_ptr(), None, None, B, N, 1.0, None,
                   ws.data_ptr())
            vals.append(loss)
        torch.cuda.synchronize()
        assert abs(vals[0].item() - ref.item()) < 1e-5 * max(1.0, abs(ref.item()))
        assert torch.equal(vals[0], vals[1]) and torch.equal(vals[0], vals[2])


def test_embedding_gradient_is_a_fixed_order_gather():
    """nn.Embedding(padding_idx=0) backward (models/text_encoder.py:504-510): heavy collisions (V = 50 over 40 960 tokens)."""
    L = sub("_lib")
    B, Lq, D, V = 2048, 20, 256, 50
    rows = B * Lq
    g = torch.Generator().manual_seed(41)
    ids = torch.randint(0, V, (rows,), generator=g)
    ids[:7] = torch.tensor([0, V - 1, 1, 0, V + 3, -2, 1])          # padding, last row, out-of-range ids (ignored)
    dout = torch.randn(rows, D, generator=g)
    ok = (ids > 0) & (ids < V)
    ref = torch.zeros(V, D, dtype=torch.float64).index_add_(0, ids[ok], dout[ok].double() * math.sqrt(D))
    idd, dd = ids.to(DEV), dout.to(DEV)
    res = []
    for rep in range(2):
        demb = torch.zeros(V, D, device=DEV)
        L.call("vqa_embed_bwd", 0, idd.data_ptr(), dd.data_ptr(), demb.data_ptr(), rows, D, V, math.sqrt(D), 0.0, 0)
        res.append(demb)
    torch.cuda.synchronize()
    assert float((res[0].cpu().double() - ref).abs().max() / ref.abs().max()) < 1e-5
    assert (res[0][0] == 0).all()
    assert torch.equal(res[0], res[1])
    # += semantics: a second call on the same buffer doubles it
    L.call("vqa_embed_bwd", 0, idd.data_ptr(), dd.data_ptr(), res[1].data_ptr(), rows, D, V, math.sqrt(D), 0.0, 0)
    torch.cuda.synchronize()
    assert torch.allclose(res[1], 2 * res[0], rtol=1e-6, atol=0)


def _run(dtype, B, steps, cfgkw, batch_kw):
    cfg = O.full_config(**cfgkw)
    sd = O.init_state_dict(cfg, 77, jitter=True)
    m = pkg().load_dropin().VQAModel(**cfg, compute_dtype=dtype)
    m.load_state_dict(sd)
    m = m.to(DEV).train()
    tr = pkg().trainer.HipTrainer(m, lr=1e-3)
    grads, losses = [], []
    for s in range(steps):
        images, ids, mask, answers = O.synthetic_batch(B, seed=900 + s, **batch_kw)
        loss, _ = tr.step(images.to(DEV), ids.to(DEV), mask.to(DEV), answers.to(DEV))
        grads.append(tr.G.clone()); losses.append(loss.clone())
    torch.cuda.synchronize()
    return grads, losses, {k: v.clone() for k, v in m.state_dict().items()}


@pytest.mark.parametrize("dtype,B,cfgkw,batch_kw", [
    ("bf16", 24, dict(), dict()),                                                        # full model, 224x224, dropout ON
    ("fp32", 6, dict(), dict()),
    ("bf16", 16, dict(vocab_size=100, num_answers=10, embed_dim=32), dict(image_size=64, seq_len=10, vocab=100, num_answers=10)),
])
def test_train_step_is_bit_reproducible(dtype, B, cfgkw, batch_kw):
    g1, l1, s1 = _run(dtype, B, 3, cfgkw, batch_kw)
    g2, l2, s2 = _run(dtype, B, 3, cfgkw, batch_kw)
    for step, (a, b) in enumerate(zip(g1, g2)):
        assert float(a.abs().max()) > 0
        if not torch.equal(a, b):
            nz = (a != b).nonzero().view(-1)
            raise AssertionError(f"step {step}: {nz.numel()} of {a.numel()} gradient elements differ, first at flat offset {int(nz[0])}")
        assert torch.equal(l1[step], l2[step])
    for k in s1:
        assert torch.equal(s1[k], s2[k]), k
