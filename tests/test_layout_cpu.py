"""CPU: host-side logic of the drop-in -- state_dict layout, flat storage, init statistics,
checkpoint round trip, gradient buckets."""
import io
import math

import torch

from _pkg import pkg, sub
from oracle import vqa_oracle as O


def test_layout_matches_reference_state_dict_contract():
    LY = sub("layout")
    for kw in ({}, dict(embed_dim=32, vocab_size=100, num_answers=10), dict(use_se_attention=False, use_spatial_attention=False),
               dict(use_gating=False, num_cross_layers=1, num_transformer_layers=2)):
        cfg = O.full_config(**kw)
        ent = LY.build_entries(cfg)
        ref = O.param_shapes(cfg)
        assert [e.name for e in ent] == [n for n, _, _ in ref]
        assert [e.shape for e in ent] == [s for _, s, _ in ref]
        assert [e.is_param for e in ent] == [k not in O.BUFFER_KINDS for _, _, k in ref]
    ent = LY.build_entries(O.full_config())
    assert len(ent) == 225 and sum(e.is_param for e in ent) == 164
    assert sum(e.numel for e in ent if e.is_param) == 19_310_316
    offs = [(e.offset, e.numel) for e in ent if e.is_param]
    assert all(o % LY.ALIGN == 0 for o, _ in offs)
    assert all(offs[i][0] + offs[i][1] <= offs[i + 1][0] for i in range(len(offs) - 1))


def test_bucket_ranges_partition_the_flat_buffer_in_backward_order():
    LY = sub("layout")
    ent = LY.build_entries(O.full_config())
    b = LY.bucket_ranges(ent)
    assert [n for n, _, _ in b] == ["answer_head", "fusion", "text_encoder", "image_encoder.stage4", "image_encoder.stage3",
                                    "image_encoder.stage2", "image_encoder.stage1", "image_encoder.stem"]
    spans = sorted((lo, hi) for _, lo, hi in b)
    assert spans[0][0] == 0 and spans[-1][1] == LY.flat_size(ent)
    assert all(spans[i][1] == spans[i + 1][0] for i in range(len(spans) - 1))
    sizes = {n: hi - lo for n, lo, hi in b}
    assert sizes["image_encoder.stage4"] > 8_400_000          # 44 % of the bytes fly while stage3..stem still run


def test_dropin_state_dict_roundtrip_and_flat_views():
    M = pkg().load_dropin()
    cfg = O.full_config()
    m = M.VQAModel(**cfg, seed=3)
    assert m.get_num_parameters()["total"] == 19_310_316
    assert list(m.state_dict().keys()) == [n for n, _, _ in O.param_shapes(cfg)]
    sd = O.init_state_dict(cfg, 1, jitter=True)
    m.load_state_dict(sd)                                       # reference-layout (OIHW contiguous) checkpoint loads
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd[k]), k
    # parameters are views into one flat buffer; conv weights are physically [Cout][R][S][Cin]
    P = dict(m.named_parameters())
    w = P["image_encoder.stage2.blocks.0.conv1.weight"]
    assert w.shape == (128, 64, 3, 3) and w.stride() == (576, 1, 192, 64)
    base = m._flat.data_ptr()
    assert all(base <= p.data_ptr() < base + m._flat.numel() * 4 for p in P.values())
    buf = io.BytesIO()
    torch.save({"model_state_dict": m.state_dict(), "config": m.config}, buf)
    buf.seek(0)
    ck = torch.load(buf, weights_only=True)
    m2 = M.VQAModel(**ck["config"])
    m2.load_state_dict(ck["model_state_dict"])
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
    assert set(m.config) == set(cfg)


def test_init_distributions_follow_the_reference_recipe():
    M = pkg().load_dropin()
    m = M.VQAModel(seed=0)
    P = dict(m.named_parameters())
    w = P["image_encoder.stage3.blocks.1.conv2.weight"]
    assert abs(w.std().item() - math.sqrt(2.0 / (256 * 9))) / math.sqrt(2.0 / (256 * 9)) < 0.02     # kaiming fan_out
    e = P["text_encoder.token_embedding.weight"]
    assert e[0].abs().sum().item() == 0.0 and abs(e[1:].std().item() - 256 ** -0.5) / 256 ** -0.5 < 0.02
    x = P["answer_head.classifier.6.weight"]
    assert x.abs().max().item() <= math.sqrt(6.0 / (1000 + 256)) + 1e-6 and P["answer_head.classifier.6.bias"].abs().sum() == 0
    assert torch.all(P["image_encoder.stem.1.weight"] == 1) and torch.all(P["text_encoder.final_norm.bias"] == 0)
    assert abs(P["fusion.image_projector.position_embedding"].std().item() - 0.02) < 0.002
    pe = dict(m.named_buffers())["text_encoder.positional_encoding.pe"]
    assert torch.allclose(pe, O.sinusoid_pe(20, 256))


def test_torch_custom_ops_are_registered_with_fake_impl():
    """The HIP forward/backward are torch custom ops (vqa_hip::vqa_forward / vqa_backward); shape inference works on fake tensors."""
    from torch._subclasses.fake_tensor import FakeTensorMode
    M = pkg().load_dropin()
    m = M.VQAModel(seed=0, num_answers=37)
    schema = str(torch.ops.vqa_hip.vqa_forward.default._schema)
    assert "Tensor images" in schema and "Tensor flat_params" in schema
    with FakeTensorMode(allow_non_fake_inputs=True):
        imgs = torch.empty(3, 3, 224, 224, device="cuda")
        ids = torch.empty(3, 20, dtype=torch.long, device="cuda")
        out = torch.ops.vqa_hip.vqa_forward(imgs, ids, None, torch.empty(8, device="cuda"), m._handle, False, False)
        assert tuple(out.shape) == (3, 37) and out.dtype == torch.float32
        g = torch.ops.vqa_hip.vqa_backward(out, m._handle, 0)
        assert g.numel() == m._flat.numel()


def test_flop_model_reproduces_the_survey_figures_and_prices_the_stress_config():
    """forward 3.849 GFLOP (conv 3.627 + GEMM/attention 0.222), train step 11.311 GFLOP per pair at the default
    configuration (measured there with torch's flop counter on the real reference); bench.py prices every configuration, incl.
    BASELINE configs[4], with this function instead of a constant."""
    F_ = sub("flops")
    cfg = O.full_config()
    f = F_.forward_flops(cfg)
    assert abs(f["total"] / 1e9 - 3.849) < 2e-3 and abs(f["conv"] / 1e9 - 3.627) < 2e-3 and abs(f["gemm"] / 1e9 - 0.222) < 1e-3
    assert abs(f["stem"] / 1e6 - 236.0) < 0.1 and f["image_tokens"] == 49
    assert abs(F_.train_flops(cfg) / 1e9 - 11.311) < 2e-3
    assert abs(F_.activation_elements(cfg) - 8_433_599) / 8_433_599 < 1e-3           # SURVEY 8(d) "ALGORITHMIC bytes" element count
    stress = O.full_config(embed_dim=512, num_transformer_layers=8, num_answers=2000)
    fs = F_.forward_flops(stress, 384, 384, 20)
    assert fs["image_tokens"] == 144
    assert abs(fs["conv"] / f["conv"] - (384 / 224) ** 2) < 1e-6                     # the CNN scales with the pixel count exactly
    assert 35.0 < F_.train_flops(stress, 384, 384, 20) / 1e9 < 35.3
