"""Parameter / buffer layout of the drop-in VQAModel and its initialisation.

The table reproduces the reference's 225-entry state_dict (names, shapes, order: SURVEY.md appendix A,
reference models/vqa_model.py:184-223) and the reference's initial distributions
(models/cnn_backbone.py:420-438, models/text_encoder.py:472-477, models/cross_attention.py:111-116,
models/vqa_model.py:87-92, models/fusion.py:78-80; everything else torch defaults).

All parameters live in ONE flat fp32 buffer (8-element aligned slots) so that
  * the bf16 working copy of every weight is a single cast kernel,
  * gradient buckets for the RCCL all-reduce are contiguous slices in backward-completion order,
  * global-norm clip + AdamW are two launches over the whole model.
3x3 / 1x1 / stem conv weights are stored physically as [Cout][R][S][Cin] (torch channels_last strides on
the logical OIHW parameter), which is the [N][K] operand layout of the implicit-GEMM kernels.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Tuple

import torch

STAGE_CHANNELS = (64, 128, 256, 512)
ALIGN = 8   # elements; keeps fp32 and bf16 views 16-byte aligned

CONFIG_KEYS = ("vocab_size", "embed_dim", "num_answers", "use_se_attention", "use_spatial_attention", "se_reduction",
               "num_transformer_layers", "num_attention_heads", "ffn_hidden_dim", "max_question_length",
               "num_cross_layers", "use_gating", "dropout", "answer_dropout")


@dataclass
class Entry:
    name: str
    shape: Tuple[int, ...]
    kind: str          # init kind
    is_param: bool
    offset: int = -1   # offset in the flat parameter buffer (params only)
    krsc: bool = False # stored [Cout][R][S][Cin]

    @property
    def numel(self) -> int:
        n = 1
        for s in self.shape:
            n *= s
        return n


def build_entries(cfg: dict) -> List[Entry]:
    E: List[Entry] = []
    d = cfg["embed_dim"]

    def P(name, shape, kind, krsc=False):
        E.append(Entry(name, tuple(shape), kind, True, krsc=krsc))

    def Bf(name, shape, kind):
        E.append(Entry(name, tuple(shape), kind, False))

    def bn(prefix, c):
        P(prefix + ".weight", (c,), "ones"); P(prefix + ".bias", (c,), "zeros")
        Bf(prefix + ".running_mean", (c,), "zeros"); Bf(prefix + ".running_var", (c,), "ones")
        Bf(prefix + ".num_batches_tracked", (), "count")

    P("image_encoder.stem.0.weight", (64, 3, 7, 7), "kaiming", krsc=True)
    bn("image_encoder.stem.1", 64)
    cin = 64
    for s, cout in enumerate(STAGE_CHANNELS, start=1):
        for b in range(2):
            p = f"image_encoder.stage{s}.blocks.{b}"
            P(p + ".conv1.weight", (cout, cin if b == 0 else cout, 3, 3), "kaiming", krsc=True)
            bn(p + ".bn1", cout)
            P(p + ".conv2.weight", (cout, cout, 3, 3), "kaiming", krsc=True)
            bn(p + ".bn2", cout)
            if b == 0 and s > 1:
                P(p + ".downsample.0.weight", (cout, cin, 1, 1), "kaiming", krsc=True)
                bn(p + ".downsample.1", cout)
        if cfg["use_se_attention"]:
            r = max(cout // cfg["se_reduction"], 1)
            P(f"image_encoder.stage{s}.attention.se.fc1.weight", (r, cout), "xavier")
            P(f"image_encoder.stage{s}.attention.se.fc2.weight", (cout, r), "xavier")
        if cfg["use_spatial_attention"] and s >= 3:
            P(f"image_encoder.stage{s}.attention.spatial.conv.weight", (1, 2, 7, 7), "kaiming")
        cin = cout
    P("text_encoder.token_embedding.weight", (cfg["vocab_size"], d), "embed")
    Bf("text_encoder.positional_encoding.pe", (1, cfg["max_question_length"], d), "pe")
    f = cfg["ffn_hidden_dim"]
    for l in range(cfg["num_transformer_layers"]):
        p = f"text_encoder.layers.{l}"
        for w in "qkvo":
            P(f"{p}.self_attention.W_{w}.weight", (d, d), "linear_w")
        P(p + ".norm1.weight", (d,), "ones"); P(p + ".norm1.bias", (d,), "zeros")
        P(p + ".ffn.fc1.weight", (f, d), "linear_w"); P(p + ".ffn.fc1.bias", (f,), "linear_b")
        P(p + ".ffn.fc2.weight", (d, f), "linear_w"); P(p + ".ffn.fc2.bias", (d,), "linear_b")
        P(p + ".norm2.weight", (d,), "ones"); P(p + ".norm2.bias", (d,), "zeros")
    P("text_encoder.final_norm.weight", (d,), "ones"); P("text_encoder.final_norm.bias", (d,), "zeros")
    P("fusion.image_projector.position_embedding", (1, cfg.get("num_image_tokens", 49), d), "posemb")   # 49: models/fusion.py:66
    P("fusion.image_projector.projection.0.weight", (d, 512), "linear_w")
    P("fusion.image_projector.projection.0.bias", (d,), "linear_b")
    P("fusion.image_projector.projection.1.weight", (d,), "ones"); P("fusion.image_projector.projection.1.bias", (d,), "zeros")
    for l in range(cfg["num_cross_layers"]):
        p = f"fusion.cross_attention.layers.{l}"
        for n in ("norm_query", "norm_kv"):
            P(f"{p}.{n}.weight", (d,), "ones"); P(f"{p}.{n}.bias", (d,), "zeros")
        for w in "qkvo":
            P(f"{p}.cross_attention.W_{w}.weight", (d, d), "xavier")
        P(f"{p}.norm_ffn.weight", (d,), "ones"); P(f"{p}.norm_ffn.bias", (d,), "zeros")
        P(f"{p}.ffn.0.weight", (4 * d, d), "linear_w"); P(f"{p}.ffn.0.bias", (4 * d,), "linear_b")
        P(f"{p}.ffn.3.weight", (d, 4 * d), "linear_w"); P(f"{p}.ffn.3.bias", (d,), "linear_b")
    if cfg["use_gating"]:
        P("fusion.gate.gate.0.weight", (d, 2 * d), "linear_w"); P("fusion.gate.gate.0.bias", (d,), "linear_b")
    P("fusion.output_norm.weight", (d,), "ones"); P("fusion.output_norm.bias", (d,), "zeros")
    h = 2 * d
    P("answer_head.classifier.0.weight", (h, d), "xavier"); P("answer_head.classifier.0.bias", (h,), "zeros")
    P("answer_head.classifier.3.weight", (h // 2, h), "xavier"); P("answer_head.classifier.3.bias", (h // 2,), "zeros")
    P("answer_head.classifier.6.weight", (cfg["num_answers"], h // 2), "xavier")
    P("answer_head.classifier.6.bias", (cfg["num_answers"],), "zeros")
    off = 0
    for e in E:
        if e.is_param:
            e.offset = off
            off += (e.numel + ALIGN - 1) // ALIGN * ALIGN
    return E


def flat_size(entries: List[Entry]) -> int:
    last = [e for e in entries if e.is_param][-1]
    return last.offset + (last.numel + ALIGN - 1) // ALIGN * ALIGN


def view_of(flat: torch.Tensor, e: Entry) -> torch.Tensor:
    """Logical-shape view of entry e inside a flat buffer (channels_last strides for KRSC conv weights)."""
    seg = flat[e.offset: e.offset + e.numel]
    if e.krsc:
        co, ci, r, s = e.shape
        return seg.view(co, r, s, ci).permute(0, 3, 1, 2)
    return seg.view(e.shape)


def mat_of(flat: torch.Tensor, e: Entry) -> torch.Tensor:
    """Physical 2-D [N][K] view (GEMM operand / weight-gradient target)."""
    n = e.shape[0]
    return flat[e.offset: e.offset + e.numel].view(n, e.numel // n)


def sinusoid_pe(max_len: int, d: int) -> torch.Tensor:
    pos = torch.arange(max_len, dtype=torch.float32)[:, None]
    freq = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * (-math.log(10000.0) / d))
    pe = torch.zeros(max_len, d)
    pe[:, 0::2] = torch.sin(pos * freq)
    pe[:, 1::2] = torch.cos(pos * freq)
    return pe[None]


def init_value(e: Entry, cfg: dict, gen: torch.Generator, fan_in_of: Dict[str, int]) -> torch.Tensor:
    sh = e.shape
    if e.kind == "kaiming":            # kaiming_normal_(fan_out, relu)
        return torch.randn(sh, generator=gen) * math.sqrt(2.0 / (sh[0] * sh[2] * sh[3]))
    if e.kind == "xavier":
        a = math.sqrt(6.0 / (sh[0] + sh[1]))
        return (torch.rand(sh, generator=gen) * 2 - 1) * a
    if e.kind == "linear_w":           # nn.Linear default: U(-1/sqrt(fan_in), 1/sqrt(fan_in))
        fan_in_of[e.name[:-len("weight")]] = sh[1]
        return (torch.rand(sh, generator=gen) * 2 - 1) / math.sqrt(sh[1])
    if e.kind == "linear_b":
        return (torch.rand(sh, generator=gen) * 2 - 1) / math.sqrt(fan_in_of[e.name[:-len("bias")]])
    if e.kind == "embed":
        t = torch.randn(sh, generator=gen) * (cfg["embed_dim"] ** -0.5)
        t[0].zero_()
        return t
    if e.kind == "posemb":
        return torch.randn(sh, generator=gen) * 0.02
    if e.kind == "pe":
        return sinusoid_pe(sh[1], sh[2])
    if e.kind == "ones":
        return torch.ones(sh)
    if e.kind == "zeros":
        return torch.zeros(sh)
    if e.kind == "count":
        return torch.zeros((), dtype=torch.long)
    raise KeyError(e.kind)


# gradient buckets in backward-completion order: contiguous flat slices
def bucket_ranges(entries: List[Entry]) -> List[Tuple[str, int, int]]:
    groups = ["answer_head.", "fusion.", "text_encoder.", "image_encoder.stage4.", "image_encoder.stage3.",
              "image_encoder.stage2.", "image_encoder.stage1.", "image_encoder.stem."]
    out = []
    total = flat_size(entries)
    params = [e for e in entries if e.is_param]
    for g in groups:
        es = [e for e in params if e.name.startswith(g)]
        if not es:
            continue
        lo = es[0].offset
        nxt = [e.offset for e in params if e.offset > es[-1].offset]
        hi = min(nxt) if nxt else total
        out.append((g.rstrip("."), lo, hi))
    return out
