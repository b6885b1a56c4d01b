"""GPU: the AMP branch of the reference's unchanged train loop (training/train.py:108,146,179-195: fp16 autocast + GradScaler,
ON by default whenever device == 'cuda', which is true on ROCm torch) rehearsed on the drop-in.  The whole-model custom op is
opaque to autocast (its inputs stay fp32, its compute dtype is the module's own policy), so the contract is: the loss scale
passes linearly through the HIP backward, `unscale_` + `clip_grad_norm_` + `scaler.step` see finite, correctly scaled
gradients, no step is skipped, and the update equals the non-AMP branch's (`:197-208`)."""
import pytest
import torch

from _pkg import pkg
from oracle import vqa_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _fresh(cfg, sd, dtype):
    m = pkg().load_dropin().VQAModel(**cfg, compute_dtype=dtype)
    m.load_state_dict(sd)
    return m.to(DEV).train()


@pytest.mark.parametrize("log2_scale", [16, 24, 30, 40, 56])
@pytest.mark.parametrize("dtype,tol", [("fp32", 2e-6), ("bf16", 2.5e-4)])
def test_amp_branch_equals_plain_branch(dtype, tol, log2_scale):
    """log2_scale: GradScaler starts at 2^16 and doubles every 2000 clean steps; a bf16 / fp32 backward never overflows the way fp16
    does, so on this path the scale keeps growing.  Whatever it has grown to, a step must be EITHER the plain branch's
    update OR skipped because the gradients came out non-finite (the scaler then backs off) -- never a finite, wrong update."""
    cfg = O.full_config(dropout=0.0, answer_dropout=0.0)
    sd = O.init_state_dict(cfg, 41, jitter=True)
    images, ids, mask, answers = (t.to(DEV) for t in O.synthetic_batch(4, seed=410))
    criterion = torch.nn.CrossEntropyLoss()

    # --- AMP branch, restated from training/train.py:179-195
    m1 = _fresh(cfg, sd, dtype)
    opt1 = torch.optim.AdamW(m1.parameters(), lr=1e-4, weight_decay=0.01, betas=(0.9, 0.999))
    scaler = torch.amp.GradScaler("cuda", init_scale=float(2 ** log2_scale))
    opt1.zero_grad()
    with torch.amp.autocast("cuda"):
        logits1, _ = m1(images, ids, mask)
        loss1 = criterion(logits1, answers)
    scaler.scale(loss1).backward()
    scaler.unscale_(opt1)
    gn1 = torch.nn.utils.clip_grad_norm_(m1.parameters(), 1.0)
    scaler.step(opt1)
    scale_before = scaler.get_scale()
    scaler.update()

    # --- plain branch, training/train.py:197-208
    m2 = _fresh(cfg, sd, dtype)
    opt2 = torch.optim.AdamW(m2.parameters(), lr=1e-4, weight_decay=0.01, betas=(0.9, 0.999))
    opt2.zero_grad()
    logits2, _ = m2(images, ids, mask)
    loss2 = criterion(logits2, answers)
    loss2.backward()
    gn2 = torch.nn.utils.clip_grad_norm_(m2.parameters(), 1.0)
    opt2.step()
    torch.cuda.synchronize()

    assert logits1.dtype == torch.float32 and torch.equal(logits1, logits2)         # autocast does not reach inside the op
    start = _fresh(cfg, sd, dtype)._flat
    if not torch.isfinite(gn1):
        # the loss scale outgrew the path's range somewhere (fixed-point BatchNorm sums: partials beyond 2^41; fp32 itself beyond
        # 3e38): LOUD -- the scaler skipped the step (parameters untouched) and halved the scale, as it does for an fp16 overflow
        assert log2_scale >= 40 and scaler.get_scale() == scale_before / 2 and torch.equal(m1._flat, start)
        return
    assert scaler.get_scale() == scale_before                                       # no inf/nan found: the step was NOT skipped
    assert log2_scale <= 40 or dtype == "fp32"       # (the fp32 schedule keeps float slabs: its range is fp32's own, 2^56 still fits)
    assert abs(float(gn1) - float(gn2)) / float(gn2) < (1e-4 if dtype == "fp32" else 2e-2)
    # parameters after the step: AdamW's first step moves every element by ~lr, so compare the two branches elementwise
    diff = (m1._flat - m2._flat).abs()
    # bf16: the 65536x loss scale changes which bf16 roundings the gradients take -> a few sign flips of near-zero gradient elements
    assert float((diff > tol).float().mean()) < (1e-4 if dtype == "fp32" else 2e-2), float(diff.max())
    assert float((m1._flat - start).abs().max()) > 5e-5                             # and the step really happened
