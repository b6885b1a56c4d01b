"""Data-parallel train-step harness (this repo's counterpart of Trainer.train_epoch, reference
training/train.py:168-208, non-AMP branch): zero_grad -> forward -> CrossEntropyLoss(mean) -> backward ->
[RCCL all-reduce of gradient buckets, overlapped with the remaining backward] -> clip_grad_norm_(1.0) -> AdamW.

Everything runs on device with no host synchronisation inside a step: the loss stays a device scalar, the
global gradient norm is consumed by the AdamW kernel straight from device memory.
One process per GPU; gradients are summed over ranks with torch.distributed (backend "nccl" = RCCL over xGMI,
or "gloo" in the CPU rehearsal tests) and divided by world size inside the optimizer kernel.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist

from . import layout as LY
from ._lib import call, dt, ptr


# Gradient segments (layout.bucket_ranges, reported by engine.backward in this order: answer_head, fusion, text_encoder, stage4 ...
# stem) are all-reduced in FOUR collectives: adjacent segments that finish close together travel as one message.
#   * head + fusion + text encoder (32.4 MB): the first two finish inside the latency-bound fusion chain right after the forward
#     (~120 dependent launches of 4-40 us).  Issuing a collective there costs host time exactly where the GPU is waiting for the
#     next launch (measured with a one-rank RCCL group at B=512: 9 collectives, three of them inside the chain: +0.9 ms per step);
#     merged, the message leaves when the text encoder's backward (side stream) reports, while the stage-4 convolutions run;
#   * stage 4 (33.7 MB) and stage 3 (8.4 MB): one message each, in flight under the stage-3 ... stem backward;
#   * stage 2 + stage 1 + stem (2.7 MB): one message at the end, together with the 4-byte bad-target counter.
# Larger messages also suit xGMI: a ring / direct all-reduce is per-link bound (7 links x ~153 GB/s per GPU), and a 2 MB message
# is mostly latency.
BUCKET_GROUPS = (("answer_head", "fusion", "text_encoder"), ("image_encoder.stage4",), ("image_encoder.stage3",),
                 ("image_encoder.stage2", "image_encoder.stage1", "image_encoder.stem"))


class GradBucketReducer:
    """Sum-all-reduce of contiguous gradient buckets, issued group by group while backward is still running.
    Device-agnostic (CUDA tensors: side stream + events, RCCL; CPU tensors: gloo) so the N>1 logic is testable on CPU."""

    def __init__(self, flat_grad: torch.Tensor, buckets, process_group=None, overlap=True, force=False, avoid_streams=()):
        """force: run the whole bucket choreography (communication stream, events, async all-reduce per bucket, waits) even in a
        world of ONE rank -- a single-rank RCCL group executes the same code path an 8-GPU job does, so the one-GPU test box can
        run it for real (tests/test_gpu_bench_ranks.py, bench.py --force-reducer); needs an initialised process group."""
        self.G = flat_grad
        self.buckets = list(buckets)
        self._range = {name: (lo, hi) for name, lo, hi in self.buckets}
        # merged groups: name -> (group index); a group is issued when its last segment has been reported
        self.groups = []
        for g in BUCKET_GROUPS:
            names = [n for n in g if n in self._range]
            if not names:
                continue
            lo, hi = min(self._range[n][0] for n in names), max(self._range[n][1] for n in names)
            if sum(self._range[n][1] - self._range[n][0] for n in names) != hi - lo:
                raise RuntimeError(f"gradient segments {names} are not adjacent in the flat buffer")
            self.groups.append((names, lo, hi))
        grouped = {n for names, _, _ in self.groups for n in names}
        for name, lo, hi in self.buckets:              # (a segment outside the table travels alone)
            if name not in grouped:
                self.groups.append(([name], lo, hi))
        self._group_of = {n: i for i, (names, _, _) in enumerate(self.groups) for n in names}
        self._pending = {}                             # group index -> (segments still missing, events collected)
        self._aux: List = []
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        if force and not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("GradBucketReducer(force=True) needs an initialised torch.distributed process group")
        self.active = self.world > 1 or force
        self.overlap = overlap and flat_grad.is_cuda
        self.comm_stream = None
        if self.active and self.overlap:        # a stream that does not share a hardware queue with the compute streams (engine.py)
            from .engine import pick_concurrent_streams
            self.comm_stream = pick_concurrent_streams(flat_grad.device, 1, avoid=avoid_streams)[0]
        self._works: List = []
        self.issued: List[str] = []
        self.bytes_reduced = 0

    def _issue(self, tensor, events=()):
        if self.comm_stream is not None:
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                for e in events:
                    self.comm_stream.wait_event(e)
                self._works.append(dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        else:
            if tensor.is_cuda:
                cur = torch.cuda.current_stream()
                for e in events:
                    cur.wait_event(e)
            self._works.append(dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        self.bytes_reduced += tensor.numel() * tensor.element_size()

    def reduce_aux(self, tensor: torch.Tensor):
        """Sum a small side tensor (the per-step bad-target counter) over the ranks: issued behind the last gradient message of
        the step (ordered after everything enqueued on the current stream by then), waited for in finish() with the buckets."""
        if self.active:
            self._aux.append(tensor)                   # travels with the LAST group: no collective inside the latency-bound chain

    def on_segment(self, name: str, events=()):
        """`events`: HIP events after which every gradient kernel of this segment has been enqueued-and-ordered (the engine
        records one on each stream that wrote the bucket: the data-gradient stream and the weight-gradient side stream).
        Only the COMMUNICATION stream waits for them -- the compute streams are never joined here, so the data-gradient chain
        keeps running ahead of the weight gradients exactly as in the 1-GPU step."""
        if not self.active or name not in self._range:
            return
        gi = self._group_of[name]
        names, lo, hi = self.groups[gi]
        missing, evs = self._pending.get(gi, (set(names), []))
        missing.discard(name)
        evs = evs + list(events)
        if self.G.is_cuda:                             # the stream that reports the segment (main, or the text encoder's side stream)
            e = torch.cuda.Event()
            e.record()
            evs.append(e)
        if missing:
            self._pending[gi] = (missing, evs)
            return
        self._pending.pop(gi, None)
        self.issued.append("+".join(names))
        self._issue(self.G[lo:hi], evs)
        if len(self.issued) == len(self.groups):       # the last message of the step: the side tensors ride behind it
            for t in self._aux:
                self._issue(t)
            self._aux = []

    def finish(self):
        """Make the current stream (or the host, for gloo) wait for every bucket; returns the 1/world gradient scale."""
        if self._pending:
            raise RuntimeError(f"gradient segments never reported: {[sorted(m) for m, _ in self._pending.values()]}")
        for t in self._aux:                            # (a backward that reports no final group: still reduce the side tensors)
            self._issue(t)
        self._aux = []
        for w in self._works:
            w.wait()
        self._works = []
        self.issued = []
        return 1.0 / self.world


class HipTrainer:
    def __init__(self, model, lr=1e-4, weight_decay=0.01, betas=(0.9, 0.999), eps=1e-8, max_grad_norm=1.0,
                 process_group=None, overlap=True, force_reducer=False):
        self.model = model
        self.engine = model._ensure_engine()
        self.lr, self.wd, self.betas, self.eps, self.max_norm = lr, weight_decay, betas, eps, max_grad_norm
        flat = model._flat
        self.G = torch.zeros_like(flat)
        self.m = torch.zeros_like(flat)
        self.v = torch.zeros_like(flat)
        self.sumsq = torch.zeros(1 + 2048, device=flat.device, dtype=torch.float32)   # [0] = sum of squares, rest: block partials
        # one 8-byte scratch zeroed by ONE launch per step: [0] the loss (float32 view), [1] rows of THIS step whose target was
        # outside [0, num_answers) (int32; all-zero bits are 0.0f and 0).  The AdamW kernel skips the update when [1] != 0 and
        # accumulates {rows, steps} into `bad_targets` for check().
        self._scal = torch.zeros(2, device=flat.device, dtype=torch.int32)
        self.loss = self._scal[0:1].view(torch.float32)
        self.bad_step = self._scal[1:2]
        # {rows out of range, steps skipped} since the last check(), and [2]: steps skipped EVER (the AdamW kernel forms Adam's
        # step number as calls - skipped-ever on the device: a skipped step never advances the bias corrections, check() or not)
        self._bad = torch.zeros(3, device=flat.device, dtype=torch.int32)
        self.bad_targets = self._bad[:2]
        self.calls = 0
        self._copy_sig = None                      # parameter-version signature right after the last fused AdamW launch
        self.buckets = LY.bucket_ranges(model._entries)
        self.reducer = GradBucketReducer(self.G, self.buckets, process_group, overlap, force=force_reducer,
                                         avoid_streams=[st for st in (self.engine.side, self.engine.side2) if st is not None])
        self.world = self.reducer.world

    def step(self, images, token_ids, attention_mask, targets, metrics=None):
        """One full train step; returns (loss device scalar, logits fp32).  `metrics`: optional device-side accuracy tracker
        (dropin/utils/metrics.py VQAAccuracy) updated from the logits without a host sync (train.py:211-212 does two)."""
        eng, T = self.engine, self.engine.dtype
        dev = self.G.device
        for name, t in (("images", images), ("token_ids", token_ids), ("targets", targets)):
            if not (isinstance(t, torch.Tensor) and t.device == dev):
                raise RuntimeError(f"HipTrainer.step: `{name}` must be a tensor on {dev} (there is no CPU path)")
        if images.dim() != 4 or images.shape[1] != 3 or token_ids.dim() != 2 or token_ids.shape[0] != images.shape[0] \
                or targets.shape != (images.shape[0],):
            raise RuntimeError("HipTrainer.step: expected images [B,3,H,W], token_ids [B,L], targets [B]")
        if attention_mask is not None and not (isinstance(attention_mask, torch.Tensor) and attention_mask.device == dev
                                               and attention_mask.shape == token_ids.shape):
            raise RuntimeError(f"HipTrainer.step: `attention_mask` must be a [B,L] tensor on {dev} (or None)")
        # the kernels read raw pointers: enforce the dtypes / contiguity VQAModel.forward enforces (vqa_model.py drop-in)
        images = images.contiguous().float()
        token_ids = token_ids.contiguous().long()
        targets = targets.contiguous().long()
        self.G.zero_()
        self._scal.zero_()
        if self._copy_sig is not None and eng.adamw_copy_target() is not None and self._copy_sig == self._param_sig():
            eng._wsrc_fresh = True                 # (one-shot, consumed by the begin_step of the forward below)
        maskf = None if attention_mask is None else attention_mask.contiguous().float()
        logits, _, tape = eng.forward(images, token_ids, maskf, True, False, need_tape=True, lowp_logits=True)
        B, N = logits.shape
        # the loss kernel reads the logits in the compute dtype, writes d logits in it and leaves the fp32 logits the caller gets:
        # the same values as logits.float() -> loss -> d logits.to(bf16), two elementwise launches less between forward and backward
        lowp = logits.dtype != torch.float32
        logits_f = torch.empty((B, N), device=images.device, dtype=torch.float32) if lowp else logits
        dlogits = torch.empty((B, N), device=images.device, dtype=logits.dtype)
        ce_ws = torch.empty((B,), device=images.device, dtype=torch.float32)
        call("vqa_cross_entropy", dt(logits), ptr(logits), ptr(targets), ptr(self.loss), ptr(dlogits), ptr(logits_f) if lowp else None, B, N, 1.0,
             ptr(self.bad_step), ptr(ce_ws))               # per-row loss terms, folded in row order (bit-reproducible)
        self.reducer.reduce_aux(self.bad_step)             # every rank must skip the update of a step ANY rank rejects
        if metrics is not None:
            metrics.update(logits_f, targets)
        eng.backward(tape, dlogits, self.G, on_segment=self.reducer.on_segment if self.reducer.active else None)
        gscale = self.reducer.finish()
        call("vqa_sumsq", ptr(self.G), self.G.numel(), ptr(self.sumsq))
        self.calls += 1
        b1, b2 = self.betas
        call("vqa_adamw", ptr(self.model._flat), ptr(self.G), ptr(self.m), ptr(self.v), self.G.numel(), self.lr, b1, b2, self.eps,
             self.wd, self.calls, ptr(self.sumsq), float(self.max_norm), gscale, ptr(self.bad_step), ptr(self._bad), ptr(eng.adamw_copy_target()))
        # the kernel wrote the bf16 operand copy too: the next step() skips the cast launch if nothing touched the parameters in between
        self._copy_sig = self._param_sig() if eng.adamw_copy_target() is not None else None
        return self.loss, logits_f

    def params_changed(self):
        """Tell the trainer that the parameters were written behind torch's back (through `.data`, a raw pointer, another C-ABI call):
        the next step re-casts the bf16 operand copy instead of trusting the one the last AdamW launch wrote.  Writes through torch
        (load_state_dict, optimizers, in-place ops on the Parameters or the flat buffer) are noticed without this call."""
        self._copy_sig = None

    def _param_sig(self):
        """Version counters of the flat buffer and of every Parameter view: any torch-side write (load_state_dict, an optimizer, .copy_)
        bumps one of them.  (The C-ABI kernels do not: the fused AdamW launch is the one writer this class accounts for itself.)"""
        return (self.model._flat.data_ptr(), self.model._flat._version) + tuple(p._version for p in self.model._param_list())

    @property
    def t(self) -> int:
        """Adam's step number = optimizer updates actually applied (one host sync; the kernels never need it from the host)."""
        return self.calls - int(self._bad[2])

    def grad_norm(self) -> torch.Tensor:
        return self.sumsq[:1].sqrt() / self.world

    def check(self):
        """Host-side error check (one sync; call it per logging interval, not per step): raises like nn.CrossEntropyLoss does
        (training/train.py:120) if any step since the last check saw a target outside [0, num_answers)."""
        n, skipped = (int(x) for x in self.bad_targets.tolist())
        if n:
            self.bad_targets.zero_()
            raise IndexError(f"{n} target(s) out of range [0, {self.model.num_answers}) since the last check: {skipped} step(s) were "
                             "skipped on every rank (parameters and optimizer state untouched; that step's loss and gradients are NaN)")
