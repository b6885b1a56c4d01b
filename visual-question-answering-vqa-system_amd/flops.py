"""Algorithmic work of the hot path per image-question pair, derived from the model configuration.

FLOPs count a multiply-add as 2 and cover every contraction of VQAModel.forward (models/vqa_model.py:243-311): the 20 (+3
shortcut) convolutions of the CNN (models/cnn_backbone.py:349-410), every nn.Linear and the QK^T / PV products of the six
attention blocks.  Elementwise work, BatchNorm / LayerNorm, softmax, SE and the two 2->1 channel spatial-attention convs
(< 0.1 MFLOP) are NOT counted, exactly as in the survey's figure (3.849 GFLOP forward, 11.311 GFLOP per train step at the
default configuration -- tests/test_layout_cpu.py pins this function to those two numbers).

A train step is forward + data gradients + weight gradients of every contraction = 3 x forward, minus the stem's data
gradient (the image needs no gradient).
"""
from __future__ import annotations

from typing import Dict

STAGE_CHANNELS = (64, 128, 256, 512)


def conv_out(n: int, k: int, s: int, p: int) -> int:
    return (n + 2 * p - k) // s + 1


def forward_flops(cfg: dict, H: int = 224, W: int = 224, L: int = 20) -> Dict[str, float]:
    """Forward FLOPs per pair, by part: {'stem', 'stages', 'text', 'fusion', 'head', 'conv', 'gemm', 'total'}."""
    d, heads = cfg["embed_dim"], cfg["num_attention_heads"]
    hd = d // heads
    f = {}
    h, w = conv_out(H, 7, 2, 3), conv_out(W, 7, 2, 3)
    f["stem"] = 2.0 * h * w * 64 * 147
    h, w = conv_out(h, 3, 2, 1), conv_out(w, 3, 2, 1)
    cin, stages = 64, 0.0
    for s, cout in enumerate(STAGE_CHANNELS, start=1):
        for b in range(2):
            stride = 2 if (b == 0 and s > 1) else 1
            ci = cin if b == 0 else cout
            ho, wo = conv_out(h, 3, stride, 1), conv_out(w, 3, stride, 1)
            stages += 2.0 * ho * wo * cout * 9 * ci            # conv1
            stages += 2.0 * ho * wo * cout * 9 * cout          # conv2
            if b == 0 and s > 1:
                stages += 2.0 * ho * wo * cout * ci            # 1x1/2 shortcut
            h, w = ho, wo
        cin = cout
    f["stages"] = stages
    ntok = h * w
    ffn = cfg["ffn_hidden_dim"]
    attn = lambda lq, lk: 2.0 * 2.0 * lq * lk * hd * heads     # QK^T and PV
    lin = lambda m, n, k: 2.0 * m * n * k
    f["text"] = cfg["num_transformer_layers"] * (4 * lin(L, d, d) + attn(L, L) + lin(L, ffn, d) + lin(L, d, ffn))
    cross = lin(L, d, d) * 2 + lin(ntok, d, d) * 2 + attn(L, ntok) + lin(L, 4 * d, d) + lin(L, d, 4 * d)   # cross FFN is 4d (cross_attention.py:255)
    f["fusion"] = lin(ntok, d, 512) + cfg["num_cross_layers"] * cross + (lin(1, d, 2 * d) if cfg.get("use_gating", True) else 0.0)
    f["head"] = lin(1, 2 * d, d) + lin(1, d, 2 * d) + lin(1, cfg["num_answers"], d)
    f["conv"] = f["stem"] + f["stages"]
    f["gemm"] = f["text"] + f["fusion"] + f["head"]
    f["total"] = f["conv"] + f["gemm"]
    f["image_tokens"] = float(ntok)
    return f


def train_flops(cfg: dict, H: int = 224, W: int = 224, L: int = 20) -> float:
    """FLOPs per pair of one train step: 3 x forward minus the stem's (non-existent) data gradient."""
    f = forward_flops(cfg, H, W, L)
    return 3.0 * f["total"] - f["stem"]


def activation_elements(cfg: dict, H: int = 224, W: int = 224, L: int = 20) -> float:
    """Forward activation traffic of an ideally fused schedule, elements per pair (SURVEY 8(d) 'ALGORITHMIC bytes'): conv inputs +
    conv outputs + max-pool read/write + residual identity reads + SE (pool read + scale read/write) + spatial attention
    (3 passes) + Linear inputs/outputs.  x element size x 3 (forward + backward) = bytes per pair per train step."""
    d = cfg["embed_dim"]
    h, w = conv_out(H, 7, 2, 3), conv_out(W, 7, 2, 3)
    el = 3.0 * H * W + 64.0 * h * w                        # stem conv in / out
    hp, wp = conv_out(h, 3, 2, 1), conv_out(w, 3, 2, 1)
    el += 64.0 * h * w + 64.0 * hp * wp                    # max-pool read / write
    h, w, cin = hp, wp, 64
    for s, cout in enumerate(STAGE_CHANNELS, start=1):
        for b in range(2):
            stride = 2 if (b == 0 and s > 1) else 1
            ci = cin if b == 0 else cout
            ho, wo = conv_out(h, 3, stride, 1), conv_out(w, 3, stride, 1)
            el += ci * h * w + cout * ho * wo              # conv1 in / out
            el += 2.0 * cout * ho * wo                     # conv2 in / out
            if b == 0 and s > 1:
                el += ci * h * w + cout * ho * wo          # shortcut in / out
            el += cout * ho * wo                           # identity read at the residual add
            h, w = ho, wo
        if cfg.get("use_se_attention", True):
            el += 3.0 * cout * h * w
        if cfg.get("use_spatial_attention", True) and s >= 3:
            el += 3.0 * cout * h * w
        cin = cout
    ntok, ffn = h * w, cfg["ffn_hidden_dim"]
    lin = lambda m, n, k: float(m * (n + k))
    el += cfg["num_transformer_layers"] * (4 * lin(L, d, d) + lin(L, ffn, d) + lin(L, d, ffn))
    el += lin(ntok, d, 512) + cfg["num_cross_layers"] * (2 * lin(L, d, d) + 2 * lin(ntok, d, d) + lin(L, 4 * d, d) + lin(L, d, 4 * d))
    el += lin(1, d, 2 * d) + lin(1, 2 * d, d) + lin(1, d, 2 * d) + lin(1, cfg["num_answers"], d)
    return el
