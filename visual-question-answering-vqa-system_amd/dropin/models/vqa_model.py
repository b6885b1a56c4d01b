"""`models.vqa_model` drop-in backed by the HIP engine (reference: models/vqa_model.py).

Same constructor kwargs (:132-152), `forward(images, token_ids, attention_mask=None, return_aux=False)` (:243-311),
`.predict` (:313-339), `.get_attention_maps` (:341-369), `.get_num_parameters` (:371-380), `.config` (:226-241),
`create_vqa_model` (:383-407), `load_vqa_model` (:410-432) and the same 225 state_dict entries.
There is no CPU path: calling forward with CPU tensors, or without the built extension, raises.
"""
from __future__ import annotations

import importlib
import os
from typing import Any, Dict, List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

def _pkg():
    import sys
    here = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # .../visual-question-answering-vqa-system_amd
    root = os.path.dirname(here)
    if root not in sys.path:
        sys.path.insert(0, root)
    return importlib.import_module(os.path.basename(here))


class _Node(nn.Module):
    """Anonymous container; the module tree only exists to give parameters their reference names."""


# ----------------------------------------------------------------------------------------------------------------------
# torch custom ops: the whole HIP forward and backward are registered in the `vqa_hip` namespace
#   torch.ops.vqa_hip.vqa_forward(images, token_ids, mask, params, handle, training, want_aux) -> logits
#   torch.ops.vqa_hip.vqa_backward(dlogits, handle, tape_id) -> flat gradient buffer (layout.py slots)
# with a fake (shape-only) implementation and an autograd formula, so the node is visible to the dispatcher, autograd,
# and FakeTensor tracing.  `handle` indexes a registry of live models (ops cannot carry Python objects); forward and
# backward are explicit kernel sequences over the C ABI (engine.py).
# ----------------------------------------------------------------------------------------------------------------------
_MODELS: Dict[int, "VQAModel"] = {}
_NEXT_HANDLE = [1]


@torch.library.custom_op("vqa_hip::vqa_forward", mutates_args=(), device_types="cuda")
def _vqa_forward_op(images: torch.Tensor, token_ids: torch.Tensor, mask: Optional[torch.Tensor], flat_params: torch.Tensor,
                    handle: int, training: bool, want_aux: bool) -> torch.Tensor:
    # flat_params: the model's ONE flat fp32 parameter buffer as an autograd-tracked alias (_FlatParams below).  Handing the dispatcher
    # the 164 parameter views as a List[Tensor] cost ~0.6 ms of host time per call before the first kernel went out -- idle GPU time
    # for a caller that synchronises every step (training/train.py:211)
    model = _MODELS[handle]
    logits, aux, tape = model._engine.forward(images, token_ids, mask, training, want_aux, need_tape=True)
    model._last_aux = aux              # aux tensors are detached by construction (side channel, not graph outputs)
    model._tape_seq += 1
    model._tapes[model._tape_seq] = tape
    while len(model._tapes) > model.max_live_tapes:      # a forward whose backward never ran must not pin its activations forever
        model._tapes.pop(next(iter(model._tapes)))
    return logits


@_vqa_forward_op.register_fake
def _(images, token_ids, mask, flat_params, handle, training, want_aux):
    return images.new_empty((images.shape[0], _MODELS[handle].num_answers), dtype=torch.float32)


@torch.library.custom_op("vqa_hip::vqa_backward", mutates_args=(), device_types="cuda")
def _vqa_backward_op(dlogits: torch.Tensor, handle: int, tape_id: int) -> torch.Tensor:
    """Returns the FLAT gradient buffer (same layout as the flat parameter buffer); the autograd formula slices it."""
    model = _MODELS[handle]
    tape = model._tapes.pop(tape_id, None)
    if tape is None:
        raise RuntimeError(f"vqa_backward: the activations of forward #{tape_id} are gone -- either backward ran twice through it "
                           f"(the reference needs retain_graph=True for that) or more than max_live_tapes = {model.max_live_tapes} "
                           "training forwards were issued before its backward (raise VQAModel.max_live_tapes)")
    G = torch.zeros_like(model._flat)
    model._engine.backward(tape, dlogits.contiguous(), G, on_segment=model._on_segment)
    return G


@_vqa_backward_op.register_fake
def _(dlogits, handle, tape_id):
    return torch.empty_like(_MODELS[handle]._flat)


def _setup_ctx(ctx, inputs, output):
    handle = inputs[4]
    ctx.handle = handle
    ctx.tape_id = _MODELS[handle]._tape_seq


def _backward(ctx, dlogits):
    G = torch.ops.vqa_hip.vqa_backward(dlogits, ctx.handle, ctx.tape_id)
    return None, None, None, G, None, None, None


torch.library.register_autograd("vqa_hip::vqa_forward", _backward, setup_context=_setup_ctx)


class _FlatParams(torch.autograd.Function):
    """The flat parameter buffer as a function of the 164 leaf Parameters that are views into it: forward aliases the buffer (no
    copy), backward hands each Parameter its slice of the flat gradient (views of ONE tensor, as `.grad` wants them for the flat
    optimizer paths).  A plain autograd.Function with 164 inputs costs ~0.15 ms of host time; the custom op behind it takes one tensor."""

    @staticmethod
    def forward(ctx, flat, handle, *params):
        ctx.handle = handle
        return flat.detach()

    @staticmethod
    def backward(ctx, G):
        model = _MODELS[ctx.handle]
        lay = model._pkg.layout
        return (None, None) + tuple(lay.view_of(G, e) for e in model._param_entries)


class VQAModel(nn.Module):
    def __init__(self, vocab_size: int = 10000, embed_dim: int = 256, num_answers: int = 1000,
                 use_se_attention: bool = True, use_spatial_attention: bool = True, se_reduction: int = 16,
                 num_transformer_layers: int = 4, num_attention_heads: int = 8, ffn_hidden_dim: int = 1024,
                 max_question_length: int = 20, num_cross_layers: int = 2, use_gating: bool = True,
                 dropout: float = 0.1, answer_dropout: float = 0.3, compute_dtype: Optional[str] = None, seed: Optional[int] = None,
                 num_image_tokens: int = 49):
        # compute_dtype / seed / num_image_tokens are extensions; the reference hard-codes 49 image positions (models/fusion.py:66),
        # num_image_tokens = 144 is the 384x384 stress shape of BASELINE configs[4]
        super().__init__()
        assert embed_dim % num_attention_heads == 0, \
            f"embed_dim ({embed_dim}) must be divisible by num_heads ({num_attention_heads})"
        self._pkg = _pkg()
        lay = self._pkg.layout
        self.embed_dim, self.num_answers = embed_dim, num_answers
        self.config = dict(vocab_size=vocab_size, embed_dim=embed_dim, num_answers=num_answers,
                           use_se_attention=use_se_attention, use_spatial_attention=use_spatial_attention,
                           se_reduction=se_reduction, num_transformer_layers=num_transformer_layers,
                           num_attention_heads=num_attention_heads, ffn_hidden_dim=ffn_hidden_dim,
                           max_question_length=max_question_length, num_cross_layers=num_cross_layers,
                           use_gating=use_gating, dropout=dropout, answer_dropout=answer_dropout)
        if num_image_tokens != 49:
            self.config["num_image_tokens"] = int(num_image_tokens)
        cd = (compute_dtype or os.environ.get("VQA_HIP_DTYPE", "bf16")).lower()
        self.compute_dtype = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp32": torch.float32, "float32": torch.float32}[cd]
        self._entries = lay.build_entries(self.config)
        self._param_entries = [e for e in self._entries if e.is_param]
        self._flat = torch.zeros(lay.flat_size(self._entries), dtype=torch.float32)
        gen = torch.Generator()
        gen.manual_seed(torch.initial_seed() if seed is None else seed)
        fan: Dict[str, int] = {}
        for e in self._entries:
            val = lay.init_value(e, self.config, gen, fan)
            node = self
            *path, leaf = e.name.split(".")
            for part in path:
                if not hasattr(node, part):
                    node.add_module(part, _Node())
                node = getattr(node, part)
            if e.is_param:
                view = lay.view_of(self._flat, e)
                view.copy_(val)
                node.register_parameter(leaf, nn.Parameter(view))
            else:
                node.register_buffer(leaf, val)
        self.image_encoder.output_channels = 512
        self.image_encoder.output_spatial_size = int(round(num_image_tokens ** 0.5))   # 7 in the reference (models/cnn_backbone.py:415)
        self.fusion.get_attention_visualization = self._attention_visualization
        self._engine = None
        self._on_segment = None
        self._last_aux = None
        self._tapes: Dict[int, Any] = {}
        self._graphs: Dict[Any, Any] = {}           # captured inference graphs, one per input shape (forward_graphed)
        self._tape_seq = 0
        self._handle = _NEXT_HANDLE[0]
        _NEXT_HANDLE[0] += 1
        _MODELS[self._handle] = self

    def _param_list(self):
        # the 164 Parameter objects never change identity (.to() only swaps their .data), but resolving 164 dotted names through
        # nn.Module.__getattr__ on every forward cost the unchanged train.py loop a few hundred host microseconds with the GPU idle
        pl = self.__dict__.get("_params_cache")
        if pl is not None:
            # nn.Module._apply may REPLACE Parameter objects (torch.__future__.set_overwrite_module_params_on_conversion) and a user
            # may re-register one: a stale list would route gradients to orphans (optimizer sees grad = None).  _reflatten drops the
            # cache; this two-lookup probe (first and last parameter) catches the re-registration case
            e0, e1 = self._param_entries[0], self._param_entries[-1]
            if getattr_path(self, e0.name) is not pl[0] or getattr_path(self, e1.name) is not pl[-1]:
                pl = None
        if pl is None:
            pl = [getattr_path(self, e.name) for e in self._param_entries]
            self.__dict__["_params_cache"] = pl
        return pl

    def __del__(self):
        _MODELS.pop(getattr(self, "_handle", -1), None)

    # ---- storage management: parameters are views into one flat buffer; keep that true across .to()/.cuda()
    def _reflatten(self):
        lay = self._pkg.layout
        named = dict(self.named_parameters())
        dev = named[self._param_entries[0].name].device
        flat = torch.zeros(self._flat.numel(), dtype=torch.float32, device=dev)
        for e in self._param_entries:
            p = named[e.name]
            v = lay.view_of(flat, e)
            v.copy_(p.data.to(torch.float32))
            p.data = v
            if p.grad is not None:
                p.grad = None
        self._flat = flat
        self._engine = None
        self.__dict__.pop("_params_cache", None)
        if self._graphs:                             # captured graphs hold the OLD flat buffer's pointers
            torch.cuda.synchronize()
            self._graphs.clear()

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._reflatten()
        return out

    def _ensure_engine(self):
        if self._engine is None:
            if not self._flat.is_cuda:
                raise RuntimeError("VQAModel (HIP) needs its parameters on the GPU: call model.to('cuda'); there is no CPU path")
            bufs = {n: b for n, b in self.named_buffers()}
            self._engine = self._pkg.engine.HipEngine(self.config, self._entries, self._flat, bufs, self.compute_dtype)
        return self._engine

    # ---- reference API
    def forward(self, images: torch.Tensor, token_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                return_aux: bool = False) -> Tuple[torch.Tensor, Optional[Dict]]:
        if not images.is_cuda:
            raise RuntimeError("VQAModel (HIP) got CPU inputs; this implementation only runs on an MI355X (no CPU fallback)")
        eng = self._ensure_engine()
        images = images.contiguous().float()
        token_ids = token_ids.contiguous().long()
        maskf = None if attention_mask is None else attention_mask.contiguous().float()
        params = self._param_list()
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            # tapes are kept by id (gradient accumulation / loss1 + loss2: several forwards, then their backwards), oldest dropped
            # beyond max_live_tapes
            flat = _FlatParams.apply(self._flat, self._handle, *params)
            logits = torch.ops.vqa_hip.vqa_forward(images, token_ids, maskf, flat, self._handle, self.training, return_aux)
            aux, self._last_aux = self._last_aux, None
        elif (self.graph_inference and not self.training and not return_aux and 0 < images.shape[0] <= self.graph_max_batch
              and not torch.cuda.is_current_stream_capturing()):
            # the serving case: api/inference.py:228,296 calls model(image, ids, mask) under no_grad at B = 1 ... a few; ~190 launches
            # of a few microseconds are host-bound there, so the captured HIP graph of this shape is replayed (same kernels, same
            # logits); the result is copied out of the graph's static buffer so callers own what they get, like the reference
            logits, aux = self.forward_graphed(images, token_ids, attention_mask).clone(), None
        else:
            logits, aux, _ = eng.forward(images, token_ids, maskf, self.training, return_aux, need_tape=False)
        return (logits, aux) if return_aux else (logits, None)

    def _forward_eager_eval(self, images, token_ids, maskf):
        logits, _, _ = self._ensure_engine().forward(images, token_ids, maskf, False, False, need_tape=False)
        return logits

    def forward_graphed(self, images: torch.Tensor, token_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Latency mode for serving (api/inference.py:228,296 calls model(...) in eval mode under no_grad at B = 1 ... 64, where
        ~190 launches of a few microseconds each are host-bound): the eval forward (Conv+BN folded) is captured once per input
        shape into a HIP graph and replayed; inputs are copied into the graph's static buffers.  Inference only: eval mode, no
        autograd, logits only.  Returns the graph's STATIC output buffer (valid until the next call with this shape; forward()
        hands out a copy).  The shape cache is LRU; a graph is only dropped after the stream that replayed it has drained."""
        if self.training:
            raise RuntimeError("forward_graphed is the inference path: call model.eval() first")
        if not images.is_cuda:
            raise RuntimeError("VQAModel (HIP) got CPU inputs; this implementation only runs on an MI355X (no CPU fallback)")
        key = (tuple(images.shape), tuple(token_ids.shape), attention_mask is not None, self._flat.data_ptr(), self._ensure_engine().fold_eval,
               getattr(self._engine, "fuse_stem_eval", None))
        g = self._graphs.pop(key, None)
        if g is not None:
            self._graphs[key] = g                    # LRU: a hit moves the shape to the young end
        else:
            st_img = images.detach().clone().contiguous().float()
            st_ids = token_ids.detach().clone().contiguous().long()
            st_msk = None if attention_mask is None else attention_mask.detach().clone().contiguous().float()
            with torch.no_grad():
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):                       # warm-up off the default stream, as graph capture requires
                    for _ in range(2):
                        self._forward_eager_eval(st_img, st_ids, st_msk)
                torch.cuda.current_stream().wait_stream(side)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    out = self._forward_eager_eval(st_img, st_ids, st_msk)
            g = (graph, st_img, st_ids, st_msk, out)
            if len(self._graphs) >= self.graph_max_shapes:
                torch.cuda.synchronize()             # an evicted graph's private pool must outlive its last replay in flight
                while len(self._graphs) >= self.graph_max_shapes:
                    self._graphs.pop(next(iter(self._graphs)))
            self._graphs[key] = g
        graph, st_img, st_ids, st_msk, out = g
        st_img.copy_(images); st_ids.copy_(token_ids)
        if st_msk is not None:
            st_msk.copy_(attention_mask)
        graph.replay()
        return out

    graph_inference = True        # eval-mode no-grad forward()/predict() replay a captured HIP graph up to graph_max_batch
    graph_max_batch = 64          # (the serving case, api/inference.py:196-323: model(...) at B = 1 ... a few)
    graph_max_shapes = 16         # distinct input shapes kept captured (least recently used dropped first)
    max_live_tapes = 4            # training forwards whose backward has not run yet (each pins its activations)

    def predict(self, images, token_ids, attention_mask=None, top_k: int = 5):
        self.eval()
        with torch.no_grad():
            logits, _ = self.forward(images, token_ids, attention_mask)    # (replays the HIP graph of this shape for B <= graph_max_batch)
            probs = F.softmax(logits, dim=-1)
            top_probs, top_indices = probs.topk(top_k, dim=-1)
        return top_indices, top_probs

    def _attention_visualization(self, attention_weights: list, spatial_size: int = 7) -> torch.Tensor:
        avg = torch.stack(attention_weights, dim=0).mean(dim=0).mean(dim=1)
        b, lq, _ = avg.shape
        return avg.view(b, lq, spatial_size, spatial_size)

    def get_attention_maps(self, images, token_ids, attention_mask=None) -> Dict[str, torch.Tensor]:
        _, aux = self.forward(images, token_ids, attention_mask, return_aux=True)
        vis = self._attention_visualization(aux["cross_attention_weights"], self.image_encoder.output_spatial_size)
        return {"cross_attention": aux["cross_attention_weights"], "cross_attention_spatial": vis}

    def get_num_parameters(self) -> Dict[str, int]:
        counts = {k: sum(p.numel() for p in getattr(self, k).parameters())
                  for k in ("image_encoder", "text_encoder", "fusion", "answer_head")}
        counts["total"] = sum(counts.values())
        return counts


def getattr_path(obj, dotted: str):
    for part in dotted.split("."):
        obj = getattr(obj, part)
    return obj


def create_vqa_model(vocab_size: int = 10000, num_answers: int = 1000, use_attention: bool = True, **kwargs) -> VQAModel:
    return VQAModel(vocab_size=vocab_size, num_answers=num_answers, use_se_attention=use_attention,
                    use_spatial_attention=use_attention, **kwargs)


def load_vqa_model(checkpoint_path: str, device: str = "cpu") -> VQAModel:
    # the checkpoint dict of training/train.py:280-288 holds tensors, a config dict and scalars only: the weights-only loader
    # (executes nothing from the file) reads it
    checkpoint = torch.load(checkpoint_path, map_location=device, weights_only=True)
    model = VQAModel(**checkpoint.get("config", {}))
    model.load_state_dict(checkpoint["model_state_dict"])
    return model.to(device)
