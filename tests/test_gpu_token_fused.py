"""Round-4 launch mergers of the fusion tail (models/fusion.py:281-296): both masked means / both backward broadcasts in one launch."""
import pytest
import torch

from _pkg import sub

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(5, 20, 256), (64, 20, 256), (3, 7, 512), (512, 20, 256)])
def test_masked_pool_pair_equals_two_single_launches(shape, dtype):
    L = sub("_lib")
    B, T, D = shape
    g = torch.Generator().manual_seed(B + T)
    x0 = torch.randn(B, T, D, generator=g).to(DEV, dtype)
    x1 = torch.randn(B, T, D, generator=g).to(DEV, dtype)
    lens = torch.randint(1, T + 1, (B,), generator=g)
    lens[0] = 0                                                  # an all-padding row: the count clamps to 1 (models/fusion.py:289)
    mask = (torch.arange(T)[None, :] < lens[:, None]).float().to(DEV)
    d = L.dt(dtype)
    pair = torch.empty(B, 2 * D, device=DEV, dtype=dtype)
    L.call("vqa_masked_pool_pair_fwd", d, x0.data_ptr(), x1.data_ptr(), mask.data_ptr(), pair.data_ptr(), B, T, D)
    two = torch.empty_like(pair)
    L.call("vqa_masked_pool_fwd", d, x0.data_ptr(), mask.data_ptr(), two.data_ptr(), 2 * D, 0, B, T, D)
    L.call("vqa_masked_pool_fwd", d, x1.data_ptr(), mask.data_ptr(), two.data_ptr(), 2 * D, D, B, T, D)
    torch.cuda.synchronize()
    assert torch.equal(pair, two)
    cnt = mask.sum(1, keepdim=True).clamp(min=1.0)
    ref = torch.cat([(x0.float() * mask[..., None]).sum(1) / cnt, (x1.float() * mask[..., None]).sum(1) / cnt], 1)
    assert (pair.float() - ref).abs().max().item() < (1e-5 if dtype == torch.float32 else 2e-2)
    assert (pair[0] == 0).all()

    dcat = torch.randn(B, 2 * D, generator=g).to(DEV, dtype)
    dq, de = torch.full_like(x0, 7.0), torch.full_like(x1, 7.0)
    L.call("vqa_masked_pool_pair_bwd", d, dcat.data_ptr(), mask.data_ptr(), dq.data_ptr(), de.data_ptr(), B, T, D)
    dq2, de2 = torch.empty_like(x0), torch.empty_like(x1)
    L.call("vqa_masked_pool_bwd", d, dcat.data_ptr(), 2 * D, 0, mask.data_ptr(), None, dq2.data_ptr(), B, T, D)
    L.call("vqa_masked_pool_bwd", d, dcat.data_ptr(), 2 * D, D, mask.data_ptr(), None, de2.data_ptr(), B, T, D)
    torch.cuda.synchronize()
    assert torch.equal(dq, dq2) and torch.equal(de, de2)
    ref_q = dcat[:, None, :D].float() * mask[..., None] / cnt[..., None]
    assert (dq.float() - ref_q).abs().max().item() < (1e-6 if dtype == torch.float32 else 2e-2)
    assert (dq[mask == 0] == 0).all() and (de[mask == 0] == 0).all()


def test_pair_entries_refuse_missing_operands():
    L = sub("_lib")
    x = torch.zeros(2, 4, 8, device=DEV)
    with pytest.raises(RuntimeError):
        L.call("vqa_masked_pool_pair_fwd", 0, x.data_ptr(), None, None, x.data_ptr(), 2, 4, 8)
    with pytest.raises(RuntimeError):
        L.call("vqa_linear_dgrad_act", 0, x.data_ptr(), x.data_ptr(), x.data_ptr(), None, None, 0.1, 8, 8, 8)
