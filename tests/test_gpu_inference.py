"""GPU: inference path -- eval-mode Conv+BN folded into the implicit-GEMM epilogue (bias + ReLU, ReLU after the
residual addend).  Parity pin: the folded path is what `torch.no_grad()` evaluation runs, so tests/test_gpu_model.py's
eval goldens (logits within 1e-3 of the real reference) exercise it; here it is additionally compared with the unfolded
BN-as-a-pass path on the same weights and checked kernel by kernel."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from _pkg import pkg, sub
from oracle import vqa_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _model(dtype):
    cfg = O.full_config()
    sd = O.init_state_dict(cfg, 1, jitter=True)          # jitter: non-trivial running_mean / running_var / gamma / beta
    m = pkg().load_dropin().VQAModel(**cfg, compute_dtype=dtype)
    m.load_state_dict(sd)
    return m.to(DEV).eval()


@pytest.mark.parametrize("dtype,tol", [("fp32", 2e-4), ("bf16", 6e-2)])
def test_folded_eval_equals_unfolded_eval(dtype, tol):
    m = _model(dtype)
    images, ids, _, _ = O.synthetic_batch(6, seed=21)
    mask = (torch.arange(20)[None, :] < torch.tensor([20, 3, 9, 20, 1, 14])[:, None]).long()
    args = (images.to(DEV), ids.to(DEV), mask.to(DEV))
    with torch.no_grad():
        a, _ = m(*args)
        m._engine.fold_eval = False
        b, _ = m(*args)
        m._engine.fold_eval = True
    torch.cuda.synchronize()
    assert torch.isfinite(a).all()
    assert (a - b).abs().max().item() < tol * max(1.0, b.abs().max().item())
    if dtype == "fp32":
        assert (a.argmax(-1) == b.argmax(-1)).all()


def test_folded_eval_leaves_bn_buffers_and_training_path_untouched():
    m = _model("fp32")
    before = {k: v.clone() for k, v in m.state_dict().items() if "running" in k or "num_batches" in k}
    images, ids, _, _ = O.synthetic_batch(2, seed=5)
    with torch.no_grad():
        m(images.to(DEV), ids.to(DEV), None)
    after = m.state_dict()
    assert all(torch.equal(before[k], after[k]) for k in before)
    # eval with autograd enabled still records a tape (unfolded path) and back-propagates
    lg, _ = m(images.to(DEV), ids.to(DEV), None)
    lg.sum().backward()
    assert dict(m.named_parameters())["answer_head.classifier.6.weight"].grad is not None


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fold_kernel_and_relu_after_addend(dtype):
    """vqa_fold_bn_batch + vqa_igemm(relu=2) on one 3x3 conv against F.conv2d -> F.batch_norm(eval) -> + residual -> relu."""
    L, K = sub("_lib"), sub("kernels")
    B, C, H, N = 2, 64, 12, 128
    g = torch.Generator().manual_seed(4)
    x = torch.randn(B, C, H, H, generator=g)
    w = torch.randn(N, C, 3, 3, generator=g) * 0.05
    gamma, beta = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g)
    rm, rv = torch.randn(N, generator=g) * 0.1, torch.rand(N, generator=g) + 0.5
    res = torch.randn(B, N, H, H, generator=g)
    rd = (lambda t: t.to(torch.bfloat16).float()) if dtype == torch.bfloat16 else (lambda t: t)
    ref = F.relu(F.batch_norm(F.conv2d(rd(x), w, None, 1, 1), rm, rv, gamma, beta, False, 0.0, 1e-5) + rd(res))
    flat = torch.cat([w.permute(0, 2, 3, 1).reshape(-1), gamma, beta]).to(DEV)          # [N][R][S][C] weight, then gamma, beta
    rmd, rvd = rm.to(DEV), rv.to(DEV)
    Kw = 9 * C
    desc = torch.tensor([[0, N * Kw, N * Kw + N, rmd.data_ptr(), rvd.data_ptr(), N, Kw, 0, 0, 0]], dtype=torch.int64).to(DEV)
    wout = torch.empty(N * Kw, device=DEV, dtype=dtype)
    bout = torch.empty(N, device=DEV)
    L.call("vqa_fold_bn_batch", int(dtype == torch.bfloat16), flat.data_ptr(), wout.data_ptr(), bout.data_ptr(), desc.data_ptr(), 1,
           (N * Kw + 255) // 256, 1e-5)
    xa = x.permute(0, 2, 3, 1).reshape(-1, C).contiguous().to(DEV, dtype)
    ra = res.permute(0, 2, 3, 1).reshape(-1, N).contiguous().to(DEV, dtype)
    out, _, _ = K.igemm(xa, wout.view(N, Kw), B * H * H, N, Kw, (B, H, H, C, H, H, 3, 3, 1, 1), dtype=dtype, bias=bout, addend=ra, relu=2)
    torch.cuda.synchronize()
    got = out.float().cpu().view(B, H, H, N).permute(0, 3, 1, 2)
    tol = 2e-4 if dtype == torch.float32 else 3e-2
    assert (got - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item())
    assert (got >= 0).all()


def test_graphed_forward_replays_the_same_logits_for_new_inputs():
    """forward_graphed: the captured HIP graph must give the eager eval logits bit for bit, also for inputs other than the ones it
    was captured with (static buffers are refilled), and must refuse train mode."""
    m = _model("bf16")
    with torch.no_grad():
        for seed in (31, 32, 33):
            images, ids, _, _ = O.synthetic_batch(3, seed=seed)
            mask = (torch.arange(20)[None, :] < torch.tensor([20, 6, 11])[:, None]).long()
            a = m.forward_graphed(images.to(DEV), ids.to(DEV), mask.to(DEV)).clone()
            m.graph_inference = False                               # eager launches, kernel by kernel
            b, _ = m(images.to(DEV), ids.to(DEV), mask.to(DEV))
            m.graph_inference = True                                # the serving route: model(...) under no_grad replays the graph
            c, _ = m(images.to(DEV), ids.to(DEV), mask.to(DEV))
            torch.cuda.synchronize()
            assert torch.equal(a, b) and torch.equal(a, c), seed
    assert len(m._graphs) == 1
    m.train()
    with pytest.raises(RuntimeError):
        m.forward_graphed(images.to(DEV), ids.to(DEV), mask.to(DEV))


def test_graph_cache_is_lru_and_outputs_are_owned_by_the_caller():
    """api/inference.py:228,296 calls model(...) in eval mode under no_grad at small B: that route replays a captured HIP graph.
    The shape cache is LRU (a hit refreshes the entry), evicting a graph synchronises first, and forward() hands out a copy of
    the graph's static output (two results of the same shape must not alias)."""
    m = _model("bf16")
    m.graph_max_shapes = 2
    images, ids, _, _ = O.synthetic_batch(3, seed=41)
    im, idd = images.to(DEV), ids.to(DEV)
    with torch.no_grad():
        o1, _ = m(im[:1], idd[:1], None)
        o2, _ = m(im[1:2], idd[1:2], None)
        assert o1.data_ptr() != o2.data_ptr() and not torch.equal(o1, o2)     # same shape, same graph, two owned results
        k1 = next(iter(m._graphs))
        m(im[:2], idd[:2], None)                                  # second shape
        m(im[:1], idd[:1], None)                                  # hit on the first: it becomes the youngest
        assert list(m._graphs)[-1] == k1
        m(im[:3], idd[:3], None)                                  # third shape evicts the B=2 graph, not the B=1 one
        assert k1 in m._graphs and len(m._graphs) == 2
        o3, _ = m(im[:1], idd[:1], None)
    torch.cuda.synchronize()
    assert torch.equal(o3, o1)


def test_inference_stem_fusion_keeps_the_logits():
    """eval + no_grad forward with the one-launch stem (default) against the same model with it switched off: logits within the
    bf16 path's tolerance, identical top-1; the HIP-graph replay uses the fused stem too (same logits as the eager call)."""
    m = _model("bf16")
    torch.manual_seed(3)
    B = 8
    images = torch.randn(B, 3, 224, 224, device=DEV)
    ids = torch.randint(1, 900, (B, 20), device=DEV)
    mask = torch.ones(B, 20, dtype=torch.long, device=DEV)
    m.eval()
    with torch.no_grad():
        a, _ = m(images, ids, mask)
        eng = m._ensure_engine()
        assert eng.fuse_stem_eval
        eng.fuse_stem_eval = False
        m._graphs.clear() if hasattr(m, "_graphs") else None
        m.graph_inference = False
        b, _ = m(images, ids, mask)
        eng.fuse_stem_eval = True
        c, _ = m(images, ids, mask)                    # eager, fused stem
    torch.cuda.synchronize()
    assert (a - b).abs().max().item() < 6e-2 * max(1.0, float(b.abs().max()))
    assert torch.equal(a.argmax(-1), b.argmax(-1))
    assert torch.equal(a, c)
