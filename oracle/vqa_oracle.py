"""CPU oracle for the VQA forward/backward hot path -- TEST INFRASTRUCTURE ONLY.

This file is a from-scratch, functional, fp32 PyTorch-CPU restatement of the reference
algorithm for the path SURVEY.md section 8(a) names (rows A1-A14 and H).  Every function
cites the reference file:line it follows (paths relative to /root/reference).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it -- as the
checker or as the timed CPU baseline, never as the thing shipped.  The product path
(visual-question-answering-vqa-system_amd/) never imports this module and has no CPU fallback.

Pinning: tests/golden/make_golden.py imports the real reference from /root/reference in the
build container, loads the weights produced by `init_state_dict` below into it and stores the
reference's outputs under tests/golden/*.npz; tests/test_oracle_golden.py then checks this
restatement against those files.  So parity is pinned by outputs of the reference itself.

All tensors are NCHW / [B, L, D] fp32 exactly like the reference.  Parameters come in as a
flat dict keyed by the reference's state_dict names; autograd of this
restatement gives the oracle gradients.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

DEFAULT_CONFIG = dict(  # models/vqa_model.py:132-152
    vocab_size=10000, embed_dim=256, num_answers=1000, use_se_attention=True,
    use_spatial_attention=True, se_reduction=16, num_transformer_layers=4,
    num_attention_heads=8, ffn_hidden_dim=1024, max_question_length=20,
    num_cross_layers=2, use_gating=True, dropout=0.1, answer_dropout=0.3)

BN_EPS = 1e-5       # nn.BatchNorm2d default (models/cnn_backbone.py:151)
BN_MOMENTUM = 0.1
LN_EPS = 1e-5       # nn.LayerNorm default (models/text_encoder.py:360)
STAGE_CHANNELS = (64, 128, 256, 512)   # models/cnn_backbone.py:336-341


def full_config(**kw) -> dict:
    cfg = dict(DEFAULT_CONFIG)
    cfg.update(kw)
    return cfg


# --------------------------------------------------------------------------------------
# A14: parameter / buffer layout and initialisation (same distributions as the reference,
# own RNG order; models/cnn_backbone.py:420-438, text_encoder.py:472-477,
# cross_attention.py:111-116, vqa_model.py:87-92, fusion.py:78-80)
# --------------------------------------------------------------------------------------
def param_shapes(cfg: dict) -> List[Tuple[str, Tuple[int, ...], str]]:
    """Ordered (name, shape, kind) list reproducing the reference state_dict (appendix A).
    kind in {conv, bn_w, bn_b, bn_rm, bn_rv, bn_nbt, xavier, default_w, default_b, zeros, ones,
    embed, pe, posemb}."""
    out: List[Tuple[str, Tuple[int, ...], str]] = []
    d = cfg["embed_dim"]

    def bn(prefix, c):
        out.extend([(prefix + ".weight", (c,), "ones"), (prefix + ".bias", (c,), "zeros"),
                    (prefix + ".running_mean", (c,), "bn_rm"), (prefix + ".running_var", (c,), "bn_rv"),
                    (prefix + ".num_batches_tracked", (), "bn_nbt")])

    out.append(("image_encoder.stem.0.weight", (64, 3, 7, 7), "conv"))
    bn("image_encoder.stem.1", 64)
    cin = 64
    for s, cout in enumerate(STAGE_CHANNELS, start=1):
        for b in range(2):
            p = f"image_encoder.stage{s}.blocks.{b}"
            out.append((p + ".conv1.weight", (cout, cin if b == 0 else cout, 3, 3), "conv"))
            bn(p + ".bn1", cout)
            out.append((p + ".conv2.weight", (cout, cout, 3, 3), "conv"))
            bn(p + ".bn2", cout)
            if b == 0 and s > 1:
                out.append((p + ".downsample.0.weight", (cout, cin, 1, 1), "conv"))
                bn(p + ".downsample.1", cout)
        if cfg["use_se_attention"]:
            r = max(cout // cfg["se_reduction"], 1)
            out.append((f"image_encoder.stage{s}.attention.se.fc1.weight", (r, cout), "xavier"))
            out.append((f"image_encoder.stage{s}.attention.se.fc2.weight", (cout, r), "xavier"))
        if cfg["use_spatial_attention"] and s >= 3:
            out.append((f"image_encoder.stage{s}.attention.spatial.conv.weight", (1, 2, 7, 7), "conv"))
        cin = cout
    out.append(("text_encoder.token_embedding.weight", (cfg["vocab_size"], d), "embed"))
    out.append(("text_encoder.positional_encoding.pe", (1, cfg["max_question_length"], d), "pe"))
    f = cfg["ffn_hidden_dim"]
    for l in range(cfg["num_transformer_layers"]):
        p = f"text_encoder.layers.{l}"
        for w in "qkvo":
            out.append((f"{p}.self_attention.W_{w}.weight", (d, d), "default_w"))
        out += [(p + ".norm1.weight", (d,), "ones"), (p + ".norm1.bias", (d,), "zeros"),
                (p + ".ffn.fc1.weight", (f, d), "default_w"), (p + ".ffn.fc1.bias", (f,), "default_b"),
                (p + ".ffn.fc2.weight", (d, f), "default_w"), (p + ".ffn.fc2.bias", (d,), "default_b"),
                (p + ".norm2.weight", (d,), "ones"), (p + ".norm2.bias", (d,), "zeros")]
    out += [("text_encoder.final_norm.weight", (d,), "ones"), ("text_encoder.final_norm.bias", (d,), "zeros")]
    # the reference hard-codes 49 positions (models/fusion.py:66); "num_image_tokens" is this repo's extension for the 384x384 stress shape
    out.append(("fusion.image_projector.position_embedding", (1, cfg.get("num_image_tokens", 49), d), "posemb"))
    out += [("fusion.image_projector.projection.0.weight", (d, 512), "default_w"),
            ("fusion.image_projector.projection.0.bias", (d,), "default_b"),
            ("fusion.image_projector.projection.1.weight", (d,), "ones"),
            ("fusion.image_projector.projection.1.bias", (d,), "zeros")]
    for l in range(cfg["num_cross_layers"]):
        p = f"fusion.cross_attention.layers.{l}"
        for n in ("norm_query", "norm_kv"):
            out += [(f"{p}.{n}.weight", (d,), "ones"), (f"{p}.{n}.bias", (d,), "zeros")]
        for w in "qkvo":
            out.append((f"{p}.cross_attention.W_{w}.weight", (d, d), "xavier"))
        out += [(f"{p}.norm_ffn.weight", (d,), "ones"), (f"{p}.norm_ffn.bias", (d,), "zeros"),
                (f"{p}.ffn.0.weight", (4 * d, d), "default_w"), (f"{p}.ffn.0.bias", (4 * d,), "default_b"),
                (f"{p}.ffn.3.weight", (d, 4 * d), "default_w"), (f"{p}.ffn.3.bias", (d,), "default_b")]
    if cfg["use_gating"]:
        out += [("fusion.gate.gate.0.weight", (d, 2 * d), "default_w"), ("fusion.gate.gate.0.bias", (d,), "default_b")]
    out += [("fusion.output_norm.weight", (d,), "ones"), ("fusion.output_norm.bias", (d,), "zeros")]
    h = 2 * d
    out += [("answer_head.classifier.0.weight", (h, d), "xavier"), ("answer_head.classifier.0.bias", (h,), "zeros"),
            ("answer_head.classifier.3.weight", (h // 2, h), "xavier"), ("answer_head.classifier.3.bias", (h // 2,), "zeros"),
            ("answer_head.classifier.6.weight", (cfg["num_answers"], h // 2), "xavier"),
            ("answer_head.classifier.6.bias", (cfg["num_answers"],), "zeros")]
    return out


BUFFER_KINDS = ("bn_rm", "bn_rv", "bn_nbt", "pe")


def sinusoid_pe(max_len: int, d: int) -> Tensor:
    """models/text_encoder.py:83-92."""
    pe = torch.zeros(max_len, d)
    position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d, 2).float() * (-math.log(10000.0) / d))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.unsqueeze(0)


def init_state_dict(cfg: dict, seed: int = 0, jitter: bool = False) -> SD:
    """Deterministic (CPU generator) weights with the reference's init distributions.
    jitter=True perturbs BN affine / LN affine / biases / running stats away from 1/0 so that
    parity tests exercise them (a fresh reference model has them at exactly 1 and 0)."""
    g = torch.Generator().manual_seed(seed)
    sd: SD = {}
    for name, shape, kind in param_shapes(cfg):
        if kind == "conv":     # kaiming_normal_(mode='fan_out', nonlinearity='relu')
            fan_out = shape[0] * shape[2] * shape[3]
            t = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_out)
        elif kind == "xavier":
            a = math.sqrt(6.0 / (shape[0] + shape[1]))
            t = (torch.rand(shape, generator=g) * 2 - 1) * a
        elif kind == "default_w":   # nn.Linear default: kaiming_uniform(a=sqrt(5)) = U(-1/sqrt(fan_in), ..)
            a = 1.0 / math.sqrt(shape[1])
            t = (torch.rand(shape, generator=g) * 2 - 1) * a
        elif kind == "default_b":
            fan_in = {"ffn.fc1.bias": cfg["embed_dim"], "ffn.fc2.bias": cfg["ffn_hidden_dim"]}
            # bound = 1/sqrt(fan_in of the matching weight); looked up from the weight just emitted
            wname = name[:-4] + "weight"
            a = 1.0 / math.sqrt(sd[wname].shape[1])
            t = (torch.rand(shape, generator=g) * 2 - 1) * a
        elif kind == "embed":
            t = torch.randn(shape, generator=g) * (cfg["embed_dim"] ** -0.5)
            t[0].zero_()
        elif kind == "posemb":
            t = torch.randn(shape, generator=g) * 0.02
        elif kind == "pe":
            t = sinusoid_pe(shape[1], shape[2])
        elif kind in ("ones", "bn_rv"):
            t = torch.ones(shape)
            if jitter:
                t = t + 0.2 * (torch.rand(shape, generator=g) - 0.5)
        elif kind in ("zeros", "bn_rm"):
            t = torch.zeros(shape)
            if jitter:
                t = t + 0.1 * torch.randn(shape, generator=g)
        elif kind == "bn_nbt":
            t = torch.zeros((), dtype=torch.long)
        else:
            raise KeyError(kind)
        sd[name] = t
    return sd


def parameter_names(cfg: dict) -> List[str]:
    return [n for n, _, k in param_shapes(cfg) if k not in BUFFER_KINDS]


# --------------------------------------------------------------------------------------
# A3: BatchNorm2d (nn.BatchNorm2d defaults; models/cnn_backbone.py:151,158,246,351)
# --------------------------------------------------------------------------------------
def batchnorm2d(x: Tensor, sd: SD, prefix: str, training: bool, new_buffers: Optional[SD]) -> Tensor:
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    if training:
        mean = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        if new_buffers is not None:
            n = x.numel() // x.shape[1]
            with torch.no_grad():
                new_buffers[prefix + ".running_mean"] = (1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean
                new_buffers[prefix + ".running_var"] = (1 - BN_MOMENTUM) * rv + BN_MOMENTUM * var * (n / max(n - 1, 1))
                new_buffers[prefix + ".num_batches_tracked"] = sd[prefix + ".num_batches_tracked"] + 1
    else:
        mean, var = rm, rv
    xhat = (x - mean[None, :, None, None]) * torch.rsqrt(var[None, :, None, None] + BN_EPS)
    return xhat * w[None, :, None, None] + b[None, :, None, None]


# --------------------------------------------------------------------------------------
# A1: stem (models/cnn_backbone.py:349-354)
# --------------------------------------------------------------------------------------
def stem(x: Tensor, sd: SD, training: bool, nb: Optional[SD]) -> Tensor:
    y = F.conv2d(x, sd["image_encoder.stem.0.weight"], None, stride=2, padding=3)
    y = torch.relu(batchnorm2d(y, sd, "image_encoder.stem.1", training, nb))
    return F.max_pool2d(y, kernel_size=3, stride=2, padding=1)


# --------------------------------------------------------------------------------------
# A2: ResidualBlock (models/cnn_backbone.py:164-197) and ResidualStage (:267-279)
# --------------------------------------------------------------------------------------
def residual_block(x: Tensor, sd: SD, p: str, stride: int, training: bool, nb: Optional[SD]) -> Tensor:
    out = F.conv2d(x, sd[p + ".conv1.weight"], None, stride=stride, padding=1)
    out = torch.relu(batchnorm2d(out, sd, p + ".bn1", training, nb))
    out = F.conv2d(out, sd[p + ".conv2.weight"], None, stride=1, padding=1)
    out = batchnorm2d(out, sd, p + ".bn2", training, nb)
    if (p + ".downsample.0.weight") in sd:
        identity = F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride=stride, padding=0)
        identity = batchnorm2d(identity, sd, p + ".downsample.1", training, nb)
    else:
        identity = x
    return torch.relu(out + identity)


# A4: SEAttention.forward (models/attention_modules.py:109-136)
def se_attention(x: Tensor, w1: Tensor, w2: Tensor) -> Tensor:
    squeezed = x.mean(dim=(2, 3))
    excited = torch.relu(squeezed @ w1.t())
    scale = torch.sigmoid(excited @ w2.t())
    return x * scale[:, :, None, None]


# A5: SpatialAttention.forward (models/attention_modules.py:223-243)
def spatial_attention(x: Tensor, w: Tensor) -> Tensor:
    mx = x.max(dim=1, keepdim=True)[0]
    av = x.mean(dim=1, keepdim=True)
    amap = torch.sigmoid(F.conv2d(torch.cat([mx, av], dim=1), w, None, padding=w.shape[-1] // 2))
    return x * amap


def residual_stage(x: Tensor, sd: SD, s: int, training: bool, nb: Optional[SD]) -> Tensor:
    p = f"image_encoder.stage{s}"
    x = residual_block(x, sd, p + ".blocks.0", 1 if s == 1 else 2, training, nb)
    x = residual_block(x, sd, p + ".blocks.1", 1, training, nb)
    if (p + ".attention.se.fc1.weight") in sd:                      # AttentionWrapper :427-433
        x = se_attention(x, sd[p + ".attention.se.fc1.weight"], sd[p + ".attention.se.fc2.weight"])
    if (p + ".attention.spatial.conv.weight") in sd:
        x = spatial_attention(x, sd[p + ".attention.spatial.conv.weight"])
    return x


def image_encoder(images: Tensor, sd: SD, training: bool, nb: Optional[SD] = None) -> Tensor:
    """CustomResNet.forward (models/cnn_backbone.py:440-463): [B,3,H,W] -> [B,512,H/32,W/32]."""
    x = stem(images, sd, training, nb)
    for s in (1, 2, 3, 4):
        x = residual_stage(x, sd, s, training, nb)
    return x


# --------------------------------------------------------------------------------------
# token side
# --------------------------------------------------------------------------------------
def _drop(x: Tensor, p: float, training: bool) -> Tensor:
    return F.dropout(x, p, training) if (training and p > 0) else x


def layer_norm(x: Tensor, sd: SD, prefix: str) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], LN_EPS)


def multi_head_attention(q_in: Tensor, kv_in: Tensor, wq: Tensor, wk: Tensor, wv: Tensor, wo: Tensor,
                         heads: int, key_mask: Optional[Tensor], p: float, training: bool) -> Tuple[Tensor, Tensor]:
    """A7 MultiHeadSelfAttention.forward (models/text_encoder.py:219-265, keys masked with -inf) and
    A10 CrossAttention.forward (models/cross_attention.py:159-205, called with key_value_mask=None)."""
    B, Lq, D = q_in.shape
    Lk = kv_in.shape[1]
    hd = D // heads
    Q = (q_in @ wq.t()).view(B, Lq, heads, hd).transpose(1, 2)
    K = (kv_in @ wk.t()).view(B, Lk, heads, hd).transpose(1, 2)
    V = (kv_in @ wv.t()).view(B, Lk, heads, hd).transpose(1, 2)
    scores = (Q @ K.transpose(-2, -1)) / math.sqrt(hd)
    if key_mask is not None:
        scores = scores.masked_fill(key_mask[:, None, None, :] == 0, float("-inf"))
    weights = torch.softmax(scores, dim=-1)
    ctx = _drop(weights, p, training) @ V
    ctx = ctx.transpose(1, 2).contiguous().view(B, Lq, D)
    return ctx @ wo.t(), weights


def text_encoder(token_ids: Tensor, mask: Optional[Tensor], sd: SD, cfg: dict, training: bool) -> Tuple[Tensor, Tensor]:
    """A6 TransformerTextEncoder.forward (models/text_encoder.py:479-529) with A7/A8 layers (:373-399)."""
    d, p = cfg["embed_dim"], cfg["dropout"]
    x = F.embedding(token_ids, sd["text_encoder.token_embedding.weight"], padding_idx=0) * math.sqrt(d)
    x = _drop(x + sd["text_encoder.positional_encoding.pe"][:, : x.shape[1]], p, training)
    for l in range(cfg["num_transformer_layers"]):
        q = f"text_encoder.layers.{l}"
        n = layer_norm(x, sd, q + ".norm1")
        a, _ = multi_head_attention(n, n, *(sd[f"{q}.self_attention.W_{w}.weight"] for w in "qkvo"),
                                    cfg["num_attention_heads"], mask, p, training)
        x = x + _drop(a, p, training)
        n = layer_norm(x, sd, q + ".norm2")
        h = _drop(torch.relu(n @ sd[q + ".ffn.fc1.weight"].t() + sd[q + ".ffn.fc1.bias"]), p, training)
        x = x + _drop(h @ sd[q + ".ffn.fc2.weight"].t() + sd[q + ".ffn.fc2.bias"], p, training)
    enc = layer_norm(x, sd, "text_encoder.final_norm")
    return enc, masked_mean(enc, mask)


def masked_mean(x: Tensor, mask: Optional[Tensor]) -> Tensor:
    """models/text_encoder.py:522-527, models/fusion.py:303-313."""
    if mask is None:
        return x.mean(dim=1)
    m = mask.unsqueeze(-1).float()
    return (x * m).sum(dim=1) / m.sum(dim=1).clamp(min=1)


def image_projector(feat: Tensor, sd: SD, cfg: dict, training: bool) -> Tensor:
    """A9 ImageFeatureProjector.forward (models/fusion.py:98-112); pos-emb added after dropout."""
    B, C, H, W = feat.shape
    x = feat.view(B, C, H * W).permute(0, 2, 1)
    p = "fusion.image_projector.projection"
    x = x @ sd[p + ".0.weight"].t() + sd[p + ".0.bias"]
    x = _drop(layer_norm(x, sd, p + ".1"), cfg["dropout"], training)
    return x + sd["fusion.image_projector.position_embedding"][:, : H * W]


def fusion(feat: Tensor, text: Tensor, mask: Optional[Tensor], sd: SD, cfg: dict, training: bool):
    """A10/A11 MultimodalFusion.forward (models/fusion.py:252-336)."""
    p = cfg["dropout"]
    img = image_projector(feat, sd, cfg, training)
    q = text
    weights = []
    for l in range(cfg["num_cross_layers"]):        # MultiHeadCrossAttention.forward cross_attention.py:285-299
        c = f"fusion.cross_attention.layers.{l}"
        a, w = multi_head_attention(layer_norm(q, sd, c + ".norm_query"), layer_norm(img, sd, c + ".norm_kv"),
                                    *(sd[f"{c}.cross_attention.W_{k}.weight"] for k in "qkvo"),
                                    cfg["num_attention_heads"], None, p, training)
        weights.append(w)
        q = q + _drop(a, p, training)
        n = layer_norm(q, sd, c + ".norm_ffn")
        h = _drop(torch.relu(n @ sd[c + ".ffn.0.weight"].t() + sd[c + ".ffn.0.bias"]), p, training)
        q = q + _drop(h @ sd[c + ".ffn.3.weight"].t() + sd[c + ".ffn.3.bias"], p, training)
    att_pooled = masked_mean(q, mask)
    txt_pooled = masked_mean(text, mask)
    if cfg["use_gating"]:                            # GatingMechanism.forward fusion.py:160-166
        g = torch.sigmoid(torch.cat([att_pooled, txt_pooled], -1) @ sd["fusion.gate.gate.0.weight"].t()
                          + sd["fusion.gate.gate.0.bias"])
        fused = g * att_pooled + (1 - g) * txt_pooled
    else:
        fused = att_pooled + txt_pooled
    fused = layer_norm(fused, sd, "fusion.output_norm")
    return fused, dict(cross_attention_weights=weights, image_projected=img,
                       attended_pooled=att_pooled, text_pooled=txt_pooled)


def answer_head(x: Tensor, sd: SD, cfg: dict, training: bool) -> Tensor:
    """A12 AnswerHead.forward (models/vqa_model.py:73-104)."""
    p = cfg["answer_dropout"]
    c = "answer_head.classifier"
    x = _drop(torch.relu(x @ sd[c + ".0.weight"].t() + sd[c + ".0.bias"]), p, training)
    x = _drop(torch.relu(x @ sd[c + ".3.weight"].t() + sd[c + ".3.bias"]), p, training)
    return x @ sd[c + ".6.weight"].t() + sd[c + ".6.bias"]


def vqa_forward(images: Tensor, token_ids: Tensor, attention_mask: Optional[Tensor], sd: SD, cfg: dict,
                training: bool = False, new_buffers: Optional[SD] = None):
    """A13 VQAModel.forward (models/vqa_model.py:243-311).  Returns (logits, aux)."""
    feat = image_encoder(images, sd, training, new_buffers)
    text, text_pooled = text_encoder(token_ids, attention_mask, sd, cfg, training)
    fused, faux = fusion(feat, text, attention_mask, sd, cfg, training)
    logits = answer_head(fused, sd, cfg, training)
    aux = dict(image_features=feat, text_features=text, text_pooled=text_pooled, fused=fused)
    aux.update(faux)   # NB: fusion's 'text_pooled' overwrites the encoder's, as `**fusion_aux` does (:303-309)
    return logits, aux


# --------------------------------------------------------------------------------------
# H: one train step = Trainer.train_epoch non-AMP branch (training/train.py:176-208)
# --------------------------------------------------------------------------------------
class OracleTrainer:
    """zero_grad -> forward -> CrossEntropyLoss(mean) -> backward -> clip_grad_norm_(1.0) -> AdamW.step
    with TrainingConfig defaults (training/train.py:120-132; utils/config.py: lr 1e-4, wd 0.01)."""

    def __init__(self, sd: SD, cfg: dict, lr=1e-4, weight_decay=0.01, betas=(0.9, 0.999), max_grad_norm=1.0):
        self.cfg = cfg
        names = set(parameter_names(cfg))
        self.sd = {k: (v.clone().requires_grad_(True) if k in names else v.clone()) for k, v in sd.items()}
        self.params = [self.sd[k] for k in parameter_names(cfg)]
        self.opt = torch.optim.AdamW(self.params, lr=lr, weight_decay=weight_decay, betas=betas)
        self.max_grad_norm = max_grad_norm

    def step(self, images, token_ids, attention_mask, targets):
        self.opt.zero_grad()
        nb: SD = {}
        logits, _ = vqa_forward(images, token_ids, attention_mask, self.sd, self.cfg, True, nb)
        loss = F.cross_entropy(logits, targets)
        loss.backward()
        gnorm = torch.nn.utils.clip_grad_norm_(self.params, self.max_grad_norm)
        self.opt.step()
        self.sd.update(nb)
        return loss.detach(), logits.detach(), gnorm


def synthetic_batch(batch: int, seed: int, image_size: int = 224, seq_len: int = 20, vocab: int = 1000,
                    num_answers: int = 1000):
    """DemoVQADataset.__getitem__ + vqa_collate_fn shapes/dtypes (data/dataset.py:420-436,
    data/preprocess.py:305-315) from one CPU generator."""
    g = torch.Generator().manual_seed(seed)
    images = torch.randn(batch, 3, image_size, image_size, generator=g)
    token_ids = torch.randint(0, vocab, (batch, seq_len), generator=g)
    lens = torch.randint(5, seq_len + 1, (batch,), generator=g)
    mask = (torch.arange(seq_len)[None, :] < lens[:, None]).long()
    answers = torch.randint(0, num_answers, (batch,), generator=g)
    return images, token_ids, mask, answers
