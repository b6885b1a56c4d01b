"""ctypes binding of libvqa_hip.so (the C ABI declared in include/vqa_hip.h).

There is no fallback: if the shared library is missing or a call fails, a RuntimeError is raised.
Every entry takes raw device pointers, explicit sizes and a hipStream_t; PyTorch owns all memory.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvqa_hip.so")

P, I, F, D, LL, ULL = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_longlong, C.c_ulonglong

# name -> argument ctypes (all return int; last argument is always the stream unless noted)
SIGNATURES = {
    "vqa_gemm8p": [P, P, P, I, I, I, P],
    "vqa_gemm4w": [P, P, P, I, I, I, P],
    "vqa_conv8p_ok": [I, I, I, I, I],
    "vqa_conv8p": [P, P, P, P, P, P, P, P, P, P, I, P, P, I, I, I, I, I, I, I, P],
    "vqa_igemm_mtiles": [I, I, I],
    "vqa_igemm_variant": [I] * 15,
    "vqa_igemm": [I, I, P, P, P, P, P, P, P, P] + [I] * 15 + [F, ULL, I, P],
    "vqa_linear_dgrad_act": [I, P, P, P, P, P, F, I, I, I, P],
    "vqa_wgrad_plan": [I] * 11 + [P, P, P, P, P],
    "vqa_wgrad": [I, I, P, P, P] + [I] * 13 + [P, LL, P],
    "vqa_pack_rows": [I, P, P, I, I, I, P],
    "vqa_pack_transpose": [I, P, P, I, I, I, I, I, I, P],
    "vqa_pack_transpose_batch": [I, P, P, P, I, I, P],
    "vqa_fold_bn_batch": [I, P, P, P, P, I, I, F, P],
    "vqa_conv3x3_c64_blocks": [I, I, I],
    "vqa_conv3x3_c64": [P, P, P, P, P, P, I, I, I, P],
    "vqa_conv3x3_c64p_blocks": [I, I, I],
    "vqa_conv3x3_c64p": [P, P, P, P, I, I, I, I, P],
    "vqa_conv3x3_c64p_epi": [P, P, P, P, P, P, I, I, I, P],
    "vqa_conv3x3_c64p_bnred": [P, P, P, P, P, P, I, I, I, P],
    "vqa_conv3x3_c64p_bn": [P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, D, F, F, P],
    "vqa_wgrad3x3_c64_bn_ok": [I, I, I],
    "vqa_wgrad3x3_c64_bn": [P, P, P, P, I, I, I, P, LL, P],
    "vqa_wgrad3x3_c128_blocks": [I, I, I],
    "vqa_wgrad3x3_c128": [P, P, P, I, I, I, P, LL, P],
    "vqa_wgrad3x3_c64_blocks": [I, I, I],
    "vqa_wgrad3x3_c64": [P, P, P, I, I, I, P, LL, P],
    "vqa_wgrad_group_ws": [I, I, P, P, P],
    "vqa_wgrad_group": [I, I, P, P, P, P, P, P, P, LL, P],
    "vqa_slab_reduce": [P, P, I, LL, P],
    "vqa_dgrad_s2": [I, P, P, P, P, I, I, I, I, I, I, I, I, I, P],
    "vqa_stem_conv_blocks": [I, I, I],
    "vqa_stem_pack": [P, P, P],
    "vqa_stem_conv": [P, P, P, P, I, I, I, P],
    "vqa_stem_conv_pool_ok": [I, I, I],
    "vqa_stem_conv_pool": [P, P, P, P, I, I, I, P],
    "vqa_stem_wgrad_blocks": [I, I, I],
    "vqa_stem_wgrad": [P, P, P, I, I, I, P, LL, P],
    "vqa_stem_wgrad_fused": [P, P, P, P, P, P, P, I, I, I, P, LL, P],
    "vqa_bn_stats_finalize": [P, I, I, D, P, P, P, P, P, F, F, P, P, P],
    "vqa_bn_eval_coef": [I, P, P, P, P, F, P, P],
    "vqa_bn_apply": [I, P, P, P, P, P, LL, I, I, P],
    "vqa_bn_apply_pool_chunks": [I, I, I],
    "vqa_bn_apply_pool": [I, P, P, P, P, P, I, I, I, I, P, P],
    "vqa_se_bwd_blocks": [I, I, I, I],
    "vqa_se_bwd_scratch": [I, I, I, I, I],
    "vqa_bn_bwd_blocks": [LL],
    "vqa_bn_bwd_reduce": [I, P, P, P, P, P, P, P, LL, I, I, I, P],
    "vqa_bn_acc_words": [I, I],
    "vqa_bn_apply_acc": [I, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, D, F, F, P, P],
    "vqa_bn_bwd_apply_acc": [I, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, LL, I, D, I, P],
    "vqa_bn_bwd_finalize": [P, I, I, I, D, P, P, I, P, P, P, P],
    "vqa_bn_bwd_apply": [I, P, P, P, P, P, P, P, P, LL, I, P, P],
    "vqa_stem_pool_fwd": [I, P, P, P, P, I, I, I, I, P],
    "vqa_stem_bwd_reduce": [I, P, P, P, P, P, I, I, I, I, P],
    "vqa_stem_bwd_apply": [I, P, P, P, P, P, P, I, I, I, I, P],
    "vqa_se_fwd": [I, P, P, P, P, P, P, P, I, I, I, I, P, I, P],
    "vqa_se_bwd": [I, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, P, P, P, I, P],
    "vqa_spatial_fwd": [I, P, P, P, P, P, P, I, I, I, I, P],
    "vqa_spatial_bwd_scratch": [I, I, I],
    "vqa_spatial_bwd": [I, P, P, P, P, P, P, P, P, P, I, I, I, I, P],
    "vqa_nhwc_to_nchw": [I, P, P, I, I, I, P],
    "vqa_nchw_to_nhwc": [I, P, P, I, I, I, P],
    "vqa_embed_fwd": [I, P, P, P, P, I, I, I, I, F, F, ULL, P],
    "vqa_embed_bwd": [I, P, P, P, I, I, I, F, F, ULL, P],
    "vqa_layernorm_fwd": [I, P, P, P, P, P, I, I, F, F, ULL, P, I, P],
    "vqa_layernorm_bwd_ws": [I, I, I, I],
    "vqa_layernorm_bwd": [I, P, P, P, P, P, P, P, P, I, I, F, ULL, P, I, P, I, P],
    "vqa_layernorm_bwd_folds": [I, I, I, I, P],
    "vqa_attention_fwd": [I, P, P, P, I, I, I, P, P, P, I, I, I, I, I, I, F, ULL, P],
    "vqa_attention_fwd_mfma": [P, P, P, I, I, I, P, P, P, I, I, I, I, I, I, F, ULL, P],
    "vqa_attention_bwd": [I, P, I, P, P, P, I, I, I, P, P, P, P, I, I, I, I, I, I, I, I, F, ULL, P],
    "vqa_accuracy_update": [P, P, P, I, I, P],
    "vqa_attention_bwd_mfma": [P, I, P, P, P, I, I, I, P, P, P, P, I, I, I, I, I, I, I, I, F, ULL, P],
    "vqa_masked_pool_fwd": [I, P, P, P, I, I, I, I, I, P],
    "vqa_masked_pool_bwd": [I, P, I, I, P, P, P, I, I, I, P],
    "vqa_masked_pool_pair_fwd": [I, P, P, P, P, I, I, I, P],
    "vqa_masked_pool_pair_bwd": [I, P, P, P, P, I, I, I, P],
    "vqa_gate_fwd": [I, P, P, P, I, I, P],
    "vqa_gate_bwd": [I, P, P, P, P, P, I, I, P],
    "vqa_add": [I, P, P, P, LL, P],
    "vqa_bias_act_bwd_ws": [I, I, I],
    "vqa_bias_act_bwd": [I, P, P, P, P, I, I, F, ULL, P, I, P],
    "vqa_bias_act_bwd_fold_rows": [I, I, I],
    "vqa_fold_group": [I, P, P, P, P, P, P, P, P],
    "vqa_cross_entropy": [I, P, P, P, P, P, I, I, F, P, P, P],
    "vqa_convert": [I, I, P, P, LL, P],
    "vqa_sumsq": [P, LL, P, P],
    "vqa_image_normalize": [P, P, P, I, I, I, F, F, F, F, F, F, P],
    "vqa_image_color_jitter": [P, P, P, I, I, I, P, P, F, F, F, F, F, F, P, P],
    "vqa_image_resize_ws": [I, P, P, I, I, I],
    "vqa_image_resize": [P, P, P, P, P, I, I, I, I, I, P, P, P, F, F, F, F, F, F, P, LL, P],
    "vqa_pack_tokens": [P, P, P, P, I, I, I, I, I, I, P],
    "vqa_adamw": [P, P, P, P, LL, F, F, F, F, F, LL, P, F, F, P, P, P, P],
}
_RET_LL = {"vqa_image_resize_ws", "vqa_wgrad_group_ws", "vqa_spatial_bwd_scratch", "vqa_se_bwd_scratch", "vqa_layernorm_bwd_ws", "vqa_bias_act_bwd_ws"}                       # return a size (long long)
_NO_STATUS = _RET_LL | {"vqa_wgrad3x3_c64_blocks", "vqa_bn_acc_words", "vqa_bn_apply_pool_chunks", "vqa_se_bwd_blocks", "vqa_wgrad3x3_c128_blocks", "vqa_layernorm_bwd_folds", "vqa_bias_act_bwd_fold_rows", "vqa_conv3x3_c64p_blocks", "vqa_stem_wgrad_blocks", "vqa_igemm_mtiles", "vqa_igemm_variant", "vqa_bn_bwd_blocks", "vqa_stem_conv_blocks", "vqa_conv3x3_c64_blocks", "vqa_stem_conv_pool_ok", "vqa_wgrad3x3_c64_bn_ok", "vqa_conv8p_ok"}   # return a count, not a status

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with __graft_entry__.build() "
                               "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = LL if name in _RET_LL else I
        _lib = L
    return _lib


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


# Optional launch hook (kernels.py installs the live profiler of bench.py here): hook(name, args) -> closer | None
_HOOK = [None]


def call(name: str, *args):
    """Invoke a status-returning entry on the current torch stream; raise on non-zero status."""
    fn = getattr(lib(), name)
    hook = _HOOK[0]
    done = hook(name, args) if hook is not None else None
    rc = fn(*args, stream())
    if done is not None:
        done()
    if rc != 0:
        raise RuntimeError(f"{name} failed with status {rc}" + (" (argument/shape error)" if rc == 1000 else " (hipError_t)"))


def count(name: str, *args) -> int:
    assert name in _NO_STATUS
    return getattr(lib(), name)(*args)


def dt(t_or_dtype) -> int:
    d = t_or_dtype.dtype if isinstance(t_or_dtype, torch.Tensor) else t_or_dtype
    if d == torch.float32:
        return 0
    if d == torch.bfloat16:
        return 1
    raise TypeError(f"unsupported dtype {d}")
