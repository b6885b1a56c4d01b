"""GPU parity of the stem backward the benchmark times: `vqa_stem_wgrad_fused` (BN/ReLU/MaxPool backward rebuilt on the fly
inside the 7x7/2 weight-gradient kernel, the 112x112x64 gradient never written) --
  (a) against the two-launch path (vqa_stem_bwd_apply materialises dy, then the generic weight-gradient kernel) at B=64, and
  (b) against autograd of the CPU oracle's stem (oracle.vqa_oracle.stem == models/cnn_backbone.py:349-354) at 224x224.
bf16 path: the image and the conv weight are rounded to bf16 for the MFMA, y is stored in bf16; tolerances are stated per check."""
import pytest
import torch

from _pkg import pkg, sub
from oracle import vqa_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _engine_after_forward(B, seed):
    cfg = O.full_config(dropout=0.0, answer_dropout=0.0)
    sd = O.init_state_dict(cfg, seed, jitter=True)
    m = pkg().load_dropin().VQAModel(**cfg, compute_dtype="bf16")
    m.load_state_dict(sd)
    m = m.to(DEV).train()
    eng = m._ensure_engine()
    images, ids, mask, _ = O.synthetic_batch(B, seed=seed + 1)
    _, _, tape = eng.forward(images.to(DEV), ids.to(DEV), mask.to(DEV).float(), True, False, need_tape=True)
    return m, eng, tape, sd, images


def _relerr(got, ref):
    return float((got - ref).abs().max() / ref.abs().max().clamp(min=1e-12))


def test_fused_stem_wgrad_equals_two_launch_path_b64():
    m, eng, tape, _, _ = _engine_after_forward(64, seed=31)
    LY = sub("layout")
    g = torch.Generator().manual_seed(5)
    dxc = torch.randn(64 * 56 * 56, 64, generator=g).to(DEV, torch.bfloat16)
    G1, G2 = torch.zeros_like(m._flat), torch.zeros_like(m._flat)
    eng._stem_bwd(tape, dxc, G1, True, fused=True)
    eng._stem_bwd(tape, dxc, G2, True, fused=False)
    torch.cuda.synchronize()
    for name in ("image_encoder.stem.0.weight", "image_encoder.stem.1.weight", "image_encoder.stem.1.bias"):
        e = eng.E[name]
        a, b = G1[e.offset: e.offset + e.numel], G2[e.offset: e.offset + e.numel]
        assert b.abs().max() > 0
        # same bf16 y, same coefficients; the two-launch path rounds dy to bf16 before the contraction, the fused one does not
        assert _relerr(a, b) < 5e-3, name


@pytest.mark.parametrize("B", [4])
def test_fused_stem_wgrad_matches_oracle_autograd_224(B):
    m, eng, tape, sd, images = _engine_after_forward(B, seed=32)
    g = torch.Generator().manual_seed(6)
    dpool = torch.randn(B, 64, 56, 56, generator=g).to(torch.bfloat16).float()
    dxc = dpool.permute(0, 2, 3, 1).reshape(-1, 64).contiguous().to(DEV, torch.bfloat16)
    G = torch.zeros_like(m._flat)
    eng._stem_bwd(tape, dxc, G, True)                         # default route: the fused kernel (asserted below)
    torch.cuda.synchronize()
    K = sub("kernels")
    assert eng.stem_w2 is not None and K.stem_conv_blocks(B, 224, 224) > 0
    sdr = {k: v.clone() for k, v in sd.items()}
    wn = "image_encoder.stem.0.weight"
    sdr[wn] = sd[wn].to(torch.bfloat16).float().requires_grad_(True)
    for k in ("image_encoder.stem.1.weight", "image_encoder.stem.1.bias"):
        sdr[k] = sd[k].clone().requires_grad_(True)
    pooled = O.stem(images.to(torch.bfloat16).float(), sdr, True, {})
    pooled.backward(dpool)
    e = eng.E[wn]
    got = G[e.offset: e.offset + e.numel].view(64, 7, 7, 3).cpu()
    ref = sdr[wn].grad.permute(0, 2, 3, 1)
    # bf16 y (8-bit mantissa) decides ReLU signs / pooling argmax near ties and enters xhat: 2e-2 of the largest |dW| element
    assert _relerr(got, ref) < 2e-2
    assert abs(float(got.norm()) - float(ref.norm())) / float(ref.norm()) < 5e-3
    for k in ("image_encoder.stem.1.weight", "image_encoder.stem.1.bias"):
        e = eng.E[k]
        assert _relerr(G[e.offset: e.offset + e.numel].cpu(), sdr[k].grad) < 2e-2, k
