"""Thin tensor-level wrappers over the C ABI (one Python function per exported kernel family).

Layouts: CNN activations are NHWC `[B, H, W, C]`, tokens `[rows, D]`; compute dtype T is float32 (parity
path, fp32 MFMA) or bfloat16 (throughput path, bf16 MFMA with fp32 accumulation).  All statistics,
coefficients, weight gradients and optimizer state are float32.  Functions allocate outputs with torch
(the caching allocator is the only allocator) and launch on the current stream.
"""
from __future__ import annotations

import os

import torch

from . import _lib as L
from ._lib import call, dt, ptr

LOADER_NHWC, LOADER_STEM = 0, 1

# Optional live profiling (bench.py): when PROFILE is a list, igemm/wgrad bracket their launch with events on the
# launch stream and append (kernel symbol, algorithmic FLOPs, start event, end event, algorithmic bytes).
PROFILE = None
PROFILE_VARIANTS = []        # (symbol, variant label, FLOPs, start event, end event) of launches whose symbol covers several code paths
PROFILE_STAGED = {}      # symbol -> bytes staged through LDS-DMA by its profiled launches (conv8p)

# HBM-bound entries (no FLOPs worth counting): algorithmic bytes of one call from its C-ABI argument list -- the tensors the op
# must read and write once; recorded as ("hbm:<class>:<entry>", 0, e0, e1, bytes) while PROFILE is a list.
_ES = lambda d: 2 if d else 4          # element size of the compute dtype argument (0 = f32, 1 = bf16)
_P = lambda p: 1 if p else 0           # optional operand present
HBM_BYTES = {
    # BatchNorm passes (A3): statistics finalize reads the slab; apply reads y (+ residual) and writes out; backward reduce reads
    # dout, y (+ activation, + shortcut y); backward apply reads dout, y (+ activation) and writes dy (+ shortcut: y2 in, dy2 out)
    "vqa_bn_stats_finalize": ("bn", lambda a: a[1] * a[2] * 2 * 4 + a[2] * 4 * 8),
    "vqa_bn_apply": ("bn", lambda a: a[6] * _ES(a[0]) * (2 + _P(a[3]))),
    "vqa_bn_apply_pool": ("bn", lambda a: a[6] * a[7] * a[8] * _ES(a[0]) * (2 + _P(a[3]))),
    "vqa_bn_bwd_reduce": ("bn", lambda a: a[8] * a[9] * _ES(a[0]) * (2 + _P(a[2]) + _P(a[5]))),
    "vqa_bn_apply_acc": ("bn", lambda a: a[18] * a[19] * a[20] * _ES(a[0]) * (2 + _P(a[9]))),
    "vqa_bn_bwd_apply_acc": ("bn", lambda a: a[16] * _ES(a[0]) * (3 + _P(a[2]) + 2 * _P(a[10]))),
    "vqa_bn_bwd_finalize": ("bn", lambda a: a[1] * a[2] * 3 * 4),
    "vqa_bn_bwd_apply": ("bn", lambda a: a[9] * _ES(a[0]) * (3 + _P(a[2]) + 2 * _P(a[6]))),
    "vqa_stem_pool_fwd": ("stem_pool", lambda a: a[5] * a[6] * a[7] * a[8] * _ES(a[0])
                          + a[5] * ((a[6] - 1) // 2 + 1) * ((a[7] - 1) // 2 + 1) * a[8] * (_ES(a[0]) + 1)),
    # SE / spatial attention (A4, A5): pool read + scale read + write forward; dout, x in and dx out backward
    "vqa_se_fwd": ("se_spatial", lambda a: (2 if a[12] else 3) * a[8] * a[9] * a[10] * _ES(a[0])),      # pool sums handed over: no pool read
    "vqa_se_bwd": ("se_spatial", lambda a: (3 + _P(a[17])) * a[12] * a[13] * a[14] * _ES(a[0])),        # + the BatchNorm operand when fused
    "vqa_spatial_fwd": ("se_spatial", lambda a: 3 * a[7] * a[8] * a[9] * a[10] * _ES(a[0])),
    "vqa_spatial_bwd": ("se_spatial", lambda a: 3 * a[10] * a[11] * a[12] * a[13] * _ES(a[0])),
    # token side: LayerNorm, bias / activation backward, pools, gate, adds, embedding, attention (its FLOPs are tiny: latency class)
    "vqa_layernorm_fwd": ("token", lambda a: 2 * a[6] * a[7] * _ES(a[0])),
    "vqa_layernorm_bwd": ("token", lambda a: (3 + _P(a[5])) * a[9] * a[10] * _ES(a[0])),
    "vqa_bias_act_bwd": ("token", lambda a: a[5] * a[6] * _ES(a[0]) * (1 + _P(a[2]) + _P(a[3]))),
    "vqa_add": ("token", lambda a: 3 * a[4] * _ES(a[0])),
    "vqa_embed_fwd": ("token", lambda a: a[5] * a[7] * (4 + _ES(a[0]))),
    "vqa_embed_bwd": ("token", lambda a: a[4] * a[5] * _ES(a[0]) + a[6] * a[5] * 4),
    "vqa_masked_pool_fwd": ("token", lambda a: a[6] * a[7] * a[8] * _ES(a[0])),
    "vqa_masked_pool_bwd": ("token", lambda a: a[7] * a[8] * a[9] * _ES(a[0])),
    "vqa_masked_pool_pair_fwd": ("token", lambda a: 2 * a[5] * a[6] * a[7] * _ES(a[0])),
    "vqa_masked_pool_pair_bwd": ("token", lambda a: 2 * a[5] * a[6] * a[7] * _ES(a[0])),
    "vqa_attention_fwd_mfma": ("attention", lambda a: (a[10] * (a[12] + 2 * a[13]) * a[9] + a[10] * a[12] * a[9]) * 2 + a[10] * a[11] * a[12] * a[13] * 4),
    "vqa_attention_bwd_mfma": ("attention", lambda a: 2 * (a[15] * (a[17] + 2 * a[18]) * a[1]) * 2 + a[15] * a[17] * a[1] * 2 + a[15] * a[16] * a[17] * a[18] * 4),
    "vqa_attention_fwd": ("attention", lambda a: (a[11] * (a[13] + 2 * a[14]) * a[10] + a[11] * a[13] * a[10]) * _ES(a[0]) + a[11] * a[12] * a[13] * a[14] * 4),
    "vqa_attention_bwd": ("attention", lambda a: 2 * (a[16] * (a[18] + 2 * a[19]) * a[2]) * _ES(a[0]) + a[16] * a[18] * a[2] * _ES(a[0]) + a[16] * a[17] * a[18] * a[19] * 4),
    "vqa_cross_entropy": ("token", lambda a: a[6] * a[7] * (_ES(a[0]) + 4)),
    # weight staging and the optimizer tail (H, N1): cast of the flat buffer, packed data-gradient operands, sum of squares, AdamW
    "vqa_convert": ("optimizer", lambda a: a[4] * (_ES(a[0]) + _ES(a[1]))),
    "vqa_sumsq": ("optimizer", lambda a: a[1] * 4),
    "vqa_adamw": ("optimizer", lambda a: a[4] * 4 * 7),
}


def _hbm_hook(name, args):
    if PROFILE is None:
        return None
    ent = HBM_BYTES.get(name)
    if ent is None:
        return None
    cls, fn = ent
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    prof = PROFILE

    def done():
        e1.record()
        prof.append((f"hbm:{cls}:{name[4:]}", 0.0, e0, e1, float(fn(args))))
    return done


L._HOOK[0] = _hbm_hook


def igemm_variant(dtype, loader, M, N, Kw, geom) -> int:
    """The template instantiation the C dispatch (gemm_conv.hip igemm_variant) picks: BM*10000 + BN*10 + flavour."""
    B, H, W, C, Ho, Wo, R, S, stride, pad = geom
    return L.count("vqa_igemm_variant", dt(dtype), loader, M, N, Kw, B, H, W, C, Ho, Wo, R, S, stride, pad)


def igemm_symbol(dtype, loader, var) -> str:
    """Kernel symbol as rocprofv3 prints it: igemm_kernel<T, BM, BN, LOADER, waves, BK, waves/SIMD, ring slots, window loader>."""
    bm, bn, fl = var // 10000, (var % 10000) // 10, var % 10
    return f"igemm_kernel<{_tname(dtype)}, {bm}, {bn}, {loader}, 4, {_bk(dtype)}, 2, 2, {1 if fl == 1 else 0}>"


def _tname(dtype):
    return "unsigned short" if dtype == torch.bfloat16 else "float"


def _bk(dtype):
    return 64 if dtype == torch.bfloat16 else 32


def pack_rows(w2d: torch.Tensor, dtype, kp=None) -> torch.Tensor:
    """[N][K] fp32 -> [N][Kp] T (zero padded)."""
    n, k = w2d.shape
    kp = k if kp is None else kp
    if dtype == torch.float32 and kp == k:
        return w2d
    out = torch.empty((n, kp), device=w2d.device, dtype=dtype)
    call("vqa_pack_rows", dt(dtype), ptr(w2d), ptr(out), n, k, kp)
    return out


def pack_transpose(w3d: torch.Tensor, dtype, out=None, ldo=None, col0=0, flip=False) -> torch.Tensor:
    """[N][T][C] fp32 -> out[c][col0 + t*N + n] T (the data-gradient operand; row stride ldo; flip reverses the taps)."""
    n, t, c = w3d.shape
    if out is None:
        out = torch.empty((c, t, n), device=w3d.device, dtype=dtype)
        ldo = t * n
    call("vqa_pack_transpose", dt(dtype), ptr(w3d), ptr(out), n, t, c, ldo, col0, int(flip))
    return out


def c64_blocks(B, H, W) -> int:
    return L.count("vqa_conv3x3_c64_blocks", B, H, W)


def conv3x3_c64(x, w, B, H, W, *, want_stats=False, addend=None, addmask=None):
    """bf16 3x3/1 conv, 64->64 channels, LDS-patch kernel.  Returns (out [B*H*W, 64], stats slab | None, blocks)."""
    nb = c64_blocks(B, H, W)
    out = torch.empty((B * H * W, 64), device=x.device, dtype=torch.bfloat16)
    stats = torch.empty((nb, 2, 64), device=x.device, dtype=torch.float32) if want_stats else None
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    call("vqa_conv3x3_c64", ptr(x), ptr(w), ptr(out), ptr(stats), ptr(addend), ptr(addmask), B, H, W)
    if PROFILE is not None:
        e1.record()
        PROFILE.append(("conv3x3_c64_kernel", 2.0 * B * H * W * 64 * 576, e0, e1, 2 * B * H * W * 64 * 2))
    return out, stats, nb


def c64p_blocks(B, H, W) -> int:
    return L.count("vqa_conv3x3_c64p_blocks", B, H, W)


def _c64p_sym(H, W, mode):
    """The symbol rocprofv3 prints for a vqa_conv3x3_c64p* launch: <output rows per block (conv_c64.hip c64p_rows), MODE 0 plain | 1 EPI | 2 BNRED>."""
    rbp = 8 if (H % 8 == 0 and 2 * 10 * (W + 2) * 128 + 3072 <= 160 * 1024) else 4
    return "conv3x3_c64p_kernel<%d, %d>" % (rbp, mode)


def conv3x3_c64p(x, w, B, H, W, *, want_stats=False, stats_acc=None):
    """bf16 3x3/1 conv, 64->64 channels, 8-wave LDS-DMA patch kernel (no epilogue inputs).  Returns (out, stats slab | None, blocks).
    stats_acc: a zeroed int64 [2*64 + 1] fixed-point accumulator that receives the BatchNorm sums instead of a slab."""
    nb = c64p_blocks(B, H, W)
    out = torch.empty((B * H * W, 64), device=x.device, dtype=torch.bfloat16)
    stats = torch.empty((nb, 2, 64), device=x.device, dtype=torch.float32) if (want_stats and stats_acc is None) else stats_acc
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    call("vqa_conv3x3_c64p", ptr(x), ptr(w), ptr(out), ptr(stats), B, H, W, int(stats_acc is not None))
    if PROFILE is not None:
        e1.record()
        PROFILE.append((_c64p_sym(H, W, 0), 2.0 * B * H * W * 64 * 576, e0, e1, 2 * B * H * W * 64 * 2))
    return out, stats, nb


def conv3x3_c64p_bnred(x, w, B, H, W, bn_y, bn_coef, bn_facc):
    """conv3x3_c64p (data gradient, w = the flipped pack) whose output is the gradient entering relu(BatchNorm(bn_y)): also adds that
    BatchNorm's backward column sums (sum g | sum g * xhat, g = out * [bn_y * scale + shift > 0]) to the zeroed accumulator bn_facc."""
    out = torch.empty((B * H * W, 64), device=x.device, dtype=torch.bfloat16)
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    call("vqa_conv3x3_c64p_bnred", ptr(x), ptr(w), ptr(out), ptr(bn_y), ptr(bn_coef), ptr(bn_facc), B, H, W)
    if PROFILE is not None:
        e1.record()
        PROFILE.append((_c64p_sym(H, W, 2), 2.0 * B * H * W * 64 * 576, e0, e1, 3 * B * H * W * 64 * 2))
    return out


def conv3x3_c64p_epi(x, w, B, H, W, *, addend, addmask=None, outmask=None):
    """Data gradient of a 64 -> 64 channel 3x3/1 conv (w = the flipped pack) with the residual block's identity path in the epilogue:
    (conv + addend * (addmask > 0)) * (outmask > 0) on the bf16 conv value (vqa_igemm's epilogue), 8-wave LDS-DMA patch kernel."""
    out = torch.empty((B * H * W, 64), device=x.device, dtype=torch.bfloat16)
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    call("vqa_conv3x3_c64p_epi", ptr(x), ptr(w), ptr(out), ptr(addend), ptr(addmask), ptr(outmask), B, H, W)
    if PROFILE is not None:
        e1.record()
        PROFILE.append((_c64p_sym(H, W, 1), 2.0 * B * H * W * 64 * 576, e0, e1, (3 + (addmask is not None) + (outmask is not None)) * B * H * W * 64 * 2))
    return out


def conv3x3_c64p_bn(y, acc, bn, w, B, H, W, count, *, want_stats=False, stats_acc=None, momentum=0.1, eps=1e-5):
    """conv3x3_c64p(relu(BatchNorm_train(y))) without the normalised tensor: y = the previous conv's raw output, acc = its fixed-point
    statistics, bn = (gamma, beta, running_mean, running_var, num_batches_tracked).  Returns (out, stats | None, blocks, coef [4][64])."""
    nb = c64p_blocks(B, H, W)
    out = torch.empty((B * H * W, 64), device=y.device, dtype=torch.bfloat16)
    coef = torch.empty((4, 64), device=y.device, dtype=torch.float32)
    stats = torch.empty((nb, 2, 64), device=y.device, dtype=torch.float32) if (want_stats and stats_acc is None) else stats_acc
    g, b_, rm, rv, nbt = bn
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    call("vqa_conv3x3_c64p_bn", ptr(y), ptr(acc), ptr(g), ptr(b_), ptr(rm), ptr(rv), ptr(nbt), ptr(coef), ptr(w), ptr(out), ptr(stats), B, H, W,
         int(stats_acc is not None), float(count), momentum, eps)
    if PROFILE is not None:
        e1.record()
        PROFILE.append((_c64p_sym(H, W, 0), 2.0 * B * H * W * 64 * 576, e0, e1, 2 * B * H * W * 64 * 2))
    return out, stats, nb, coef


def c64w_bn_ok(B, H, W) -> bool:
    return L.count("vqa_wgrad3x3_c64_bn_ok", B, H, W) > 0


def wgrad3x3_c64_bn(y, coef, dy, dw, B, H, W):
    """dw += dy^T gather(relu(y * coef[0] + coef[1])): weight gradient of the conv that conv3x3_c64p_bn ran (8-wave kernel)."""
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    wsf = c64w_blocks(B, H, W) * 64 * 576
    ws = torch.empty(wsf, device=y.device, dtype=torch.float32)
    call("vqa_wgrad3x3_c64_bn", ptr(y), ptr(coef), ptr(dy), ptr(dw), B, H, W, ptr(ws), wsf)
    if PROFILE is not None:
        e1.record()
        PROFILE.append(("wgrad3x3_c64", 2.0 * B * H * W * 64 * 576, e0, e1, 2 * B * H * W * 64 * 2))


def conv8p_ok(B, H, W, C, N) -> bool:
    return L.count("vqa_conv8p_ok", B, H, W, C, N) > 0


def conv8p(x, w, B, H, W, C, N, *, transposed=0, stride=1, stats_acc=None, out=None, addend=None, addmask=None, outmask=None, bnred=None):
    """3x3 / 1 / pad 1 conv (or its stride-1 data gradient) on the 8-phase 224(196) x 256 x 64 tile (csrc/gemm8p.hip).  bf16 NHWC."""
    if out is None:
        out = torch.empty((B * ((H - 1) // stride + 1) * ((W - 1) // stride + 1), N), device=x.device, dtype=torch.bfloat16)
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    # fused BatchNorm-backward column sums of the stored tile: (y, coef, facc) with the ReLU mask recomputed from y, or
    # (y, coef, facc, False, y2 | None, coef2 | None) for an already masked tile (+ the shortcut BatchNorm sharing the gradient)
    bn_y, bn_coef, bn_facc, bn_self, bn_y2, bn_coef2 = (tuple(bnred) + (True, None, None))[:6] if bnred is not None else (None, None, None, True, None, None)
    call("vqa_conv8p", ptr(x), ptr(w), ptr(out), ptr(stats_acc), ptr(addend), ptr(addmask), ptr(outmask), ptr(bn_y), ptr(bn_coef), ptr(bn_facc),
         int(bool(bn_self)), ptr(bn_y2), ptr(bn_coef2),
         B, H, W, C, N, int(transposed), int(stride))
    if PROFILE is not None:
        e1.record()
        sym = "conv8p_kernel<2, 4>" if N % 256 == 0 else "conv8p_kernel<4, 2>"      # the symbol rocprofv3 prints (the C dispatch: 256 | N)
        # (the same symbol runs with and without epilogue inputs -- identity-path gradient / masks / the fused BatchNorm-backward sums:
        #  PROFILE_VARIANTS lets bench.py show the two groups of launches apart)
        PROFILE_VARIANTS.append((sym, "with epilogue inputs" if (addend is not None or outmask is not None or bnred is not None) else "plain",
                                 2.0 * out.shape[0] * N * 9 * C, e0, e1))
        # bytes the launch stages through LDS-DMA (tiles x K tiles x (tile rows + tile columns) x 128): what prices its K loop
        bmp, bn = (224, 256) if N % 256 == 0 else (448, 128)
        rpt = bmp * 7 // 8 if out.shape[0] % (bmp * 7 // 8) == 0 else bmp
        PROFILE_STAGED[sym] = PROFILE_STAGED.get(sym, 0) + (-(-out.shape[0] // rpt)) * (N // bn) * (9 * C // 64) * (bmp + bn) * 128
        PROFILE.append((sym,
                        2.0 * out.shape[0] * N * 9 * C, e0, e1,
                        (B * H * W * C + out.shape[0] * N * (1 + sum(t is not None for t in (addend, addmask, outmask, bn_y, bn_y2))) + N * 9 * C) * 2))
    return out


def c64w_blocks(B, H, W) -> int:
    """Slabs vqa_wgrad3x3_c64 wants (8-wave LDS-DMA kernel: 4 or 2 rows per block; else the 4-wave kernel); 0: unsupported shape."""
    return L.count("vqa_wgrad3x3_c64_blocks", B, H, W)


def wgrad3x3_c64(x, dy, dw, B, H, W):
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    wsf = c64w_blocks(B, H, W) * 64 * 576                     # one partial dW per persistent workgroup, reduced in a fixed order
    ws = torch.empty(wsf, device=x.device, dtype=torch.float32)
    call("vqa_wgrad3x3_c64", ptr(x), ptr(dy), ptr(dw), B, H, W, ptr(ws), wsf)
    if PROFILE is not None:
        e1.record()
        PROFILE.append(("wgrad3x3_c64", 2.0 * B * H * W * 64 * 576, e0, e1, 2 * B * H * W * 64 * 2))   # prefix of both kernels (4-wave / 8-wave DMA)


def c128_wgrad_blocks(B, H, W):
    return L.count("vqa_wgrad3x3_c128_blocks", B, H, W)


def wgrad3x3_c128(x, dy, dw, B, H, W):
    """Stage-2 shape (128 -> 128 channels, 28 x 28): 8-wave LDS-DMA weight-gradient kernel, per-workgroup slabs + fixed-order reduce."""
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    wsf = c128_wgrad_blocks(B, H, W) * 128 * 576
    ws = torch.empty(wsf, device=x.device, dtype=torch.float32)
    call("vqa_wgrad3x3_c128", ptr(x), ptr(dy), ptr(dw), B, H, W, ptr(ws), wsf)
    if PROFILE is not None:
        e1.record()
        PROFILE.append(("wgrad3x3_c128p_kernel", 2.0 * B * H * W * 128 * 1152, e0, e1, 2 * B * H * W * 128 * 2))


def dgrad_s2(dy, dyd, wt, B, H, W, C, Ho, Wo, N, R, pad, *, dtype):
    """Data gradient of a stride-2 conv (+ optional 1x1/2 shortcut) with rows grouped by parity class."""
    out = torch.empty((B * Ho * Wo, N), device=dy.device, dtype=dtype)
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    call("vqa_dgrad_s2", dt(dtype), ptr(dy), ptr(dyd), ptr(wt), ptr(out), B, H, W, C, Ho, Wo, N, R, pad)
    if PROFILE is not None:
        e1.record()
        flops = 2.0 * B * H * W * C * N * (R * R + (1 if dyd is not None else 0))
        PROFILE.append((f"igemm_kernel<{_tname(dtype)}, 128, {64 if N <= 64 else 128}, 2, 4, {_bk(dtype)}, 2, 2, 0>", flops, e0, e1,
                        (B * H * W * C * (2 if dyd is not None else 1) + B * Ho * Wo * N) * 2))
    return out


def stem_wgrad(img, dy, dw, B, H, W):
    """bf16 stem weight gradient: dw [64][7][7][3] fp32 += dy^T im2col(img)."""
    Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    ws, wsf = stem_wgrad_scratch(img.device, B, H, W)
    call("vqa_stem_wgrad", ptr(img), ptr(dy), ptr(dw), B, H, W, ptr(ws), wsf)
    if PROFILE is not None:
        e1.record()
        PROFILE.append(("stem_wgrad_kernel<false>", 2.0 * B * Ho * Wo * 64 * 147, e0, e1, B * 3 * H * W * 4 + B * Ho * Wo * 64 * 2))


def stem_wgrad_scratch(device, B, H, W):
    """Scratch for the deterministic accumulation of the stem weight-gradient kernels: one [64][147] slab per workgroup."""
    wsf = L.count("vqa_stem_wgrad_blocks", B, H, W) * 64 * 147
    return torch.empty(wsf, device=device, dtype=torch.float32), wsf


def stem_conv_blocks(B, H, W) -> int:
    return L.count("vqa_stem_conv_blocks", B, H, W)


def stem_conv(img, wstem, B, H, W, want_stats):
    """bf16 stem conv 7x7/2 from the NCHW fp32 image.  Returns (y [B*Ho*Wo, 64] bf16, stats slab | None, blocks)."""
    nb = stem_conv_blocks(B, H, W)
    Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    y = torch.empty((B * Ho * Wo, 64), device=img.device, dtype=torch.bfloat16)
    stats = torch.empty((nb, 2, 64), device=img.device, dtype=torch.float32) if want_stats else None
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    call("vqa_stem_conv", ptr(img), ptr(wstem), ptr(y), ptr(stats), B, H, W)
    if PROFILE is not None:
        e1.record()
        PROFILE.append(("stem_conv_kernel", 2.0 * B * Ho * Wo * 64 * 147, e0, e1, B * 3 * H * W * 4 + B * Ho * Wo * 64 * 2))
    return y, stats, nb


def stem_conv_pool(img, wstem, coef, B, H, W):
    """Inference stem in one launch: conv7x7/2 + BatchNorm (running statistics) + ReLU + MaxPool3x3/2 from the NCHW fp32 image.
    Returns the pooled activation [B*Hp*Wp, 64] bf16, or None when the shape is not supported (the caller takes the two-kernel path)."""
    if not L.count("vqa_stem_conv_pool_ok", B, H, W):
        return None
    Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    Hp, Wp = (Ho + 2 - 3) // 2 + 1, (Wo + 2 - 3) // 2 + 1
    x = torch.empty((B * Hp * Wp, 64), device=img.device, dtype=torch.bfloat16)
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    call("vqa_stem_conv_pool", ptr(img), ptr(wstem), ptr(coef), ptr(x), B, H, W)
    if PROFILE is not None:
        e1.record()
        PROFILE.append(("stem_conv_pool_kernel", 2.0 * B * Ho * Wo * 64 * 147, e0, e1, B * 3 * H * W * 4 + B * Hp * Wp * 64 * 2))
    return x


def igemm(a, w, M, N, Kw, geom, *, dtype, loader=LOADER_NHWC, bias=None, addend=None, addmask=None, outmask=None, want_stats=False,
          transposed=0, relu=0, drop_p=0.0, drop_seed=0, out=None, stats_acc=None):
    """out[M][N] = gather(a) @ w[N][Kw]^T with the fused epilogue.  geom = (B, H, W, C, Ho, Wo, R, S, stride, pad).
    Returns (out, stats_slab | None, mtiles).  stats_acc: a zeroed int64 [2*N + 1] fixed-point accumulator that receives the
    BatchNorm sums instead of a slab (returned in the slab's place, mtiles = 0)."""
    B, H, W, C, Ho, Wo, R, S, stride, pad = geom
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=dtype)
    stats, mt = stats_acc, 0
    if want_stats and stats_acc is None:
        mt = L.count("vqa_igemm_mtiles", M, N, loader)
        stats = torch.empty((mt, 2, N), device=a.device, dtype=torch.float32)
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    call("vqa_igemm", dt(dtype), loader, ptr(a), ptr(w), ptr(out), ptr(bias), ptr(addend), ptr(addmask), ptr(outmask), ptr(stats),
         M, N, Kw, B, H, W, C, Ho, Wo, R, S, stride, pad, transposed, relu, float(drop_p), int(drop_seed), int(stats_acc is not None))
    if PROFILE is not None:
        e1.record()
        var = igemm_variant(dtype, loader, M, N, Kw, geom)
        kreal = 147 if loader == LOADER_STEM else Kw
        flops = 2.0 * B * H * W * C * R * S * N if transposed else 2.0 * M * N * kreal
        es = 2 if dtype == torch.bfloat16 else 4
        nbytes = (B * H * W * C * (4 if loader == LOADER_STEM else es)) + (M * N + N * Kw) * es
        PROFILE.append((igemm_symbol(dtype, loader, var), flops, e0, e1, nbytes))
    return out, stats, mt


def linear_dgrad_act(dz, wt, M, Kin, N, *, dtype, outact, drop_p, addend=None):
    """dx[M][Kin] = (dz[M][N] @ wt[Kin][N]^T + addend) * (outact > 0) / (1 - drop_p): a Linear's data gradient that leaves with the ReLU(+dropout)
    mask of the layer in front applied (vqa_linear_dgrad_act; bit-equal to igemm followed by vqa_bias_act_bwd)."""
    dx = torch.empty((M, Kin), device=dz.device, dtype=dtype)
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    call("vqa_linear_dgrad_act", dt(dtype), ptr(dz), ptr(wt), ptr(dx), ptr(addend), ptr(outact), float(drop_p), M, Kin, N)
    if PROFILE is not None:
        e1.record()
        var = igemm_variant(dtype, LOADER_NHWC, M, Kin, N, linear_geom(M, N))
        es = 2 if dtype == torch.bfloat16 else 4
        PROFILE.append((igemm_symbol(dtype, LOADER_NHWC, var), 2.0 * M * N * Kin, e0, e1, (M * N + 2 * M * Kin + N * Kin) * es))
    return dx


_PLAN_CACHE = {}


def wgrad_plan(dtype, loader, M, N, Kw, B, H, W, C, R, S):
    """(kind, tile_n, tile_k, nsplit, workspace floats) of vqa_wgrad for this problem; kind 1: the 8-wave LDS-DMA kernel."""
    key = (dtype, loader, M, N, Kw, B, H, W, C, R, S)
    pl = _PLAN_CACHE.get(key)
    if pl is None:
        import ctypes as C_
        wsf, kind, tn, tk, ns = C_.c_longlong(0), C_.c_int(0), C_.c_int(0), C_.c_int(0), C_.c_int(0)
        L.lib().vqa_wgrad_plan(dt(dtype), loader, M, N, Kw, B, H, W, C, R, S, C_.byref(wsf), C_.byref(kind), C_.byref(tn), C_.byref(tk),
                               C_.byref(ns))
        pl = _PLAN_CACHE[key] = (kind.value, tn.value, tk.value, ns.value, wsf.value)
    return pl


def wgrad(dy, x, dw, M, N, Kw, geom, *, dtype, loader=LOADER_NHWC):
    """dw[N][Kw] (fp32) += dy[M][N]^T @ gather(x)[M][Kw]: deterministic two-pass split (workspace slabs + fixed-order reduce)."""
    B, H, W, C, Ho, Wo, R, S, stride, pad = geom
    kind, tn, tk, nsplit, wsf = wgrad_plan(dtype, loader, M, N, Kw, B, H, W, C, R, S)
    ws = torch.empty(wsf, device=dy.device, dtype=torch.float32) if wsf else None       # torch's caching allocator, stream-ordered
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    call("vqa_wgrad", dt(dtype), loader, ptr(dy), ptr(x), ptr(dw), M, N, Kw, B, H, W, C, Ho, Wo, R, S, stride, pad, ptr(ws), wsf)
    if PROFILE is not None:
        e1.record()
        es = 2 if dtype == torch.bfloat16 else 4
        name = f"wgrad_dma_kernel<{tn}, {tk}, 2>" if kind else f"wgrad_kernel<{_tname(dtype)}, {tn}, {tk}, {loader}>"     # (timing includes the reduce launch)
        PROFILE.append((name, 2.0 * M * N * Kw, e0, e1, (M * N + B * H * W * C) * es + N * Kw * 4))


_GROUP_OK = {}


def wgrad_group_ok(dtype, M, N, Kw):
    """True when vqa_wgrad_group takes this Linear weight gradient (the planner's 4-wave 128x128 split kernel)."""
    key = (dtype, M, N, Kw)
    v = _GROUP_OK.get(key)
    if v is None:
        import ctypes as C_
        one = lambda a: (C_.c_int * 1)(a)
        v = _GROUP_OK[key] = L.count("vqa_wgrad_group_ws", dt(dtype), 1, one(M), one(N), one(Kw)) >= 0
    return v


def wgrad_group(jobs, *, dtype):
    """jobs: up to 8 tuples (dy [M][N], x [M][Kw], dw fp32 [N][Kw] (+=), M, N, Kw) -> one launch + one fixed-order reduce launch;
    every dw is bit-identical to its own wgrad() call."""
    import ctypes as C_
    n = len(jobs)
    VP, IA = C_.c_void_p * n, C_.c_int * n
    dy, x, dw = VP(*[j[0].data_ptr() for j in jobs]), VP(*[j[1].data_ptr() for j in jobs]), VP(*[j[2].data_ptr() for j in jobs])
    Ms, Ns, Ks = IA(*[j[3] for j in jobs]), IA(*[j[4] for j in jobs]), IA(*[j[5] for j in jobs])
    wsf = L.count("vqa_wgrad_group_ws", dt(dtype), n, Ms, Ns, Ks)
    if wsf < 0:
        raise RuntimeError("wgrad_group: a job does not qualify (check wgrad_group_ok first)")
    ws = torch.empty(wsf, device=jobs[0][0].device, dtype=torch.float32) if wsf else None
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    call("vqa_wgrad_group", dt(dtype), n, dy, x, dw, Ms, Ns, Ks, ptr(ws), wsf)
    if PROFILE is not None:
        e1.record()
        es = 2 if dtype == torch.bfloat16 else 4
        PROFILE.append((f"wgrad_group_kernel<{_tname(dtype)}, 128, 128, 0>", sum(2.0 * j[3] * j[4] * j[5] for j in jobs), e0, e1,
                        sum((j[3] * j[4] + j[3] * j[5]) * es + j[4] * j[5] * 4 for j in jobs)))


def linear_geom(M, K):
    return (M, 1, 1, K, 1, 1, 1, 1, 1, 0)


# ----------------------------------------------------------------------------------------------
# BatchNorm
# ----------------------------------------------------------------------------------------------
def bn_train_coef(stats, mtiles, C, count, gamma, beta, rm, rv, nbt, momentum=0.1, eps=1e-5):
    scratch = torch.empty((64 * 2 * C,), device=stats.device, dtype=torch.float64)
    coef = torch.empty((4, C), device=stats.device, dtype=torch.float32)
    call("vqa_bn_stats_finalize", ptr(stats), mtiles, C, float(count), ptr(gamma), ptr(beta), ptr(rm), ptr(rv), ptr(nbt),
         momentum, eps, ptr(scratch), ptr(coef))
    return coef


def bn_eval_coef(C, gamma, beta, rm, rv, eps=1e-5):
    coef = torch.empty((4, C), device=gamma.device, dtype=torch.float32)
    call("vqa_bn_eval_coef", C, ptr(gamma), ptr(beta), ptr(rm), ptr(rv), eps, ptr(coef))
    return coef


def bn_apply(y, coef, C, relu, res=None, rcoef=None):
    out = torch.empty_like(y)
    call("vqa_bn_apply", dt(y), ptr(y), ptr(coef), ptr(res), ptr(rcoef), ptr(out), y.numel(), C, int(relu))
    return out


def bn_apply_acc(y, acc, bn, C, relu, B, HW, count, *, res=None, racc=None, rbn=None, pool=False, momentum=0.1, eps=1e-5):
    """Train-mode BatchNorm apply with the statistics finalize folded in (vqa_bn_apply_acc).  bn / rbn = (gamma, beta, running_mean,
    running_var, num_batches_tracked).  Returns (out, coef [4][C], rcoef | None, pool part | None, chunks)."""
    out = torch.empty_like(y)
    coef = torch.empty((4, C), device=y.device, dtype=torch.float32)
    rcoef = torch.empty((4, C), device=y.device, dtype=torch.float32) if racc is not None else None
    part, chunks = None, 0
    if pool:
        chunks = L.count("vqa_bn_apply_pool_chunks", dt(y), HW, C)
        part = torch.empty((B, chunks, C), device=y.device, dtype=torch.float32)
    g, b_, rm, rv, nbt = bn
    rg, rb, rrm, rrv, rnbt = rbn if rbn is not None else (None,) * 5
    call("vqa_bn_apply_acc", dt(y), ptr(y), ptr(acc), ptr(g), ptr(b_), ptr(rm), ptr(rv), ptr(nbt), ptr(coef), ptr(res), ptr(racc), ptr(rg), ptr(rb),
         ptr(rrm), ptr(rrv), ptr(rnbt), ptr(rcoef), ptr(out), B, HW, C, int(relu), float(count), momentum, eps, ptr(part))
    return out, coef, rcoef, part, chunks


def bn_apply_pool(y, coef, C, relu, B, HW, res=None, rcoef=None):
    """bn_apply of a stage's last block that also leaves the SE pooling sums: returns (out, part [B][chunks][C], chunks)."""
    out = torch.empty_like(y)
    chunks = L.count("vqa_bn_apply_pool_chunks", dt(y), HW, C)
    part = torch.empty((B, chunks, C), device=y.device, dtype=torch.float32)
    call("vqa_bn_apply_pool", dt(y), ptr(y), ptr(coef), ptr(res), ptr(rcoef), ptr(out), B, HW, C, int(relu), ptr(part))
    return out, part, chunks


def bn_bwd(dout, outact, y, coef, gamma, C, training, dgamma, dbeta, y2=None, coef2=None, gamma2=None, dgamma2=None, dbeta2=None,
           self_mask=False, slab=None, nb=0, facc=None, facc_filled=False):
    """BatchNorm backward for g = dout*(outact>0); optional second BN (1x1 shortcut) sharing g.
    self_mask: the ReLU directly follows this BN (no residual), so the mask relu(bn(y)) > 0 is recomputed from y and the
    activation tensor is not read at all (pass outact=None).  Returns dy (and dy2)."""
    rows = y.numel() // C
    if facc is not None:       # fixed-point accumulators + finalize folded into the apply pass (training mode, bf16 schedule)
        if not facc_filled:
            call("vqa_bn_bwd_reduce", dt(y), ptr(dout), ptr(outact), ptr(y), ptr(coef), ptr(y2), ptr(coef2), ptr(facc), rows, C, int(self_mask), 1)
        dy = torch.empty_like(y)
        dy2 = torch.empty_like(y2) if y2 is not None else None
        call("vqa_bn_bwd_apply_acc", dt(y), ptr(dout), ptr(outact), ptr(y), ptr(facc), ptr(gamma), ptr(coef), ptr(dgamma), ptr(dbeta), ptr(dy),
             ptr(y2), ptr(gamma2), ptr(coef2), ptr(dgamma2), ptr(dbeta2), ptr(dy2), y.numel(), C, float(rows), int(self_mask))
        return dy, dy2
    if slab is None:           # (else: the caller already reduced the column sums)
        nb = L.count("vqa_bn_bwd_blocks", rows)
        slab = torch.empty((nb, 3, C), device=y.device, dtype=torch.float32)
        call("vqa_bn_bwd_reduce", dt(y), ptr(dout), ptr(outact), ptr(y), ptr(coef), ptr(y2), ptr(coef2), ptr(slab), rows, C, int(self_mask), 0)
    bc = torch.empty((3, C), device=y.device, dtype=torch.float32)
    call("vqa_bn_bwd_finalize", ptr(slab), nb, C, 1, float(rows), ptr(gamma), ptr(coef), int(training), ptr(dgamma), ptr(dbeta), ptr(bc))
    dy = torch.empty_like(y)
    dy2 = bc2 = None
    if y2 is not None:
        bc2 = torch.empty((3, C), device=y.device, dtype=torch.float32)
        call("vqa_bn_bwd_finalize", ptr(slab), nb, C, 2, float(rows), ptr(gamma2), ptr(coef2), int(training), ptr(dgamma2), ptr(dbeta2), ptr(bc2))
        dy2 = torch.empty_like(y2)
    call("vqa_bn_bwd_apply", dt(y), ptr(dout), ptr(outact), ptr(y), ptr(bc), ptr(dy), ptr(y2), ptr(bc2), ptr(dy2), y.numel(), C,
         ptr(coef) if self_mask else None)
    return dy, dy2


# ----------------------------------------------------------------------------------------------
# token side helpers
# ----------------------------------------------------------------------------------------------
def layernorm_fwd(x, gamma, beta, *, eps=1e-5, drop_p=0.0, seed=0, addrow=None, period=1):
    rows, D = x.shape
    out = torch.empty_like(x)
    stats = torch.empty((rows, 2), device=x.device, dtype=torch.float32)
    call("vqa_layernorm_fwd", dt(x), ptr(x), ptr(gamma), ptr(beta), ptr(out), ptr(stats), rows, D, eps, float(drop_p), int(seed),
         ptr(addrow), period)
    return out, stats


_WS_CACHE = {}


def reduce_ws(name, *args):
    """Scratch floats of a fixed-order reduction entry (vqa_layernorm_bwd_ws / vqa_bias_act_bwd_ws), cached by shape."""
    key = (name,) + args
    v = _WS_CACHE.get(key)
    if v is None:
        v = _WS_CACHE[key] = int(L.count(name, *args))
    return v


_FOLD_DESC = {}


def ln_bwd_folds(dtype, rows, D, period):
    """[(ws offset, rows, stride, columns, n0), ...] of a deferred vqa_layernorm_bwd (cached by shape)."""
    key = (dtype, rows, D, period)
    v = _FOLD_DESC.get(key)
    if v is None:
        import ctypes as C_
        out = (C_.c_longlong * 10)()
        n = L.count("vqa_layernorm_bwd_folds", dt(dtype), rows, D, period, out)
        v = _FOLD_DESC[key] = [tuple(int(out[5 * i + k]) for k in range(5)) for i in range(n)]
    return v


def fold_group(jobs):
    """jobs: (part tensor, element offset, rows, stride, columns, dst0, n0, dst1 | None): the deferred folds of LayerNorm / bias backward
    calls, one launch per 48 jobs; each job keeps its scratch tensor alive until here."""
    import ctypes as C_
    n = len(jobs)
    if not n:
        return
    VP, IA, LA = C_.c_void_p * n, C_.c_int * n, C_.c_longlong * n
    call("vqa_fold_group", n, VP(*[j[0].data_ptr() + 4 * j[1] for j in jobs]), IA(*[j[2] for j in jobs]), LA(*[j[3] for j in jobs]),
         IA(*[j[4] for j in jobs]), VP(*[j[5].data_ptr() for j in jobs]), IA(*[j[6] for j in jobs]),
         VP(*[(j[7].data_ptr() if j[7] is not None else None) for j in jobs]))


def layernorm_bwd(dout, x, gamma, stats, dgamma, dbeta, *, addend=None, drop_p=0.0, seed=0, dadd=None, period=1, fixed_order=True, foldq=None):
    """fixed_order: dgamma / dbeta / dadd are summed through per-workgroup partial rows + an index-order fold (bit-reproducible);
    False -> float atomics.  foldq (a list): the folds are not launched here but appended as fold_group() jobs."""
    rows, D = x.shape
    dx = torch.empty_like(x)
    ws = None
    per = period if dadd is not None else 0
    if fixed_order:
        ws = torch.empty((reduce_ws("vqa_layernorm_bwd_ws", dt(x), rows, D, per),), device=x.device, dtype=torch.float32)
    defer = int(foldq is not None and ws is not None)
    call("vqa_layernorm_bwd", dt(x), ptr(dout), ptr(x), ptr(gamma), ptr(stats), ptr(addend), ptr(dx), ptr(dgamma), ptr(dbeta),
         rows, D, float(drop_p), int(seed), ptr(dadd), period, ptr(ws), defer)
    if defer:
        folds = ln_bwd_folds(x.dtype, rows, D, per)
        off, nr, st, nc, n0 = folds[0]
        foldq.append((ws, off, nr, st, nc, dgamma, n0, dbeta))
        if len(folds) > 1:
            off, nr, st, nc, n0 = folds[1]
            foldq.append((ws, off, nr, st, nc, dadd, n0, None))
    return dx
