"""Explicit forward/backward executor of the VQA model over the HIP kernels (no autograd inside).

`HipEngine.forward` replays VQAModel.forward (reference models/vqa_model.py:243-311) as a fixed sequence of
C-ABI kernel launches on the current stream and records a tape; `HipEngine.backward` walks the tape in
reverse, writing every parameter gradient into a flat fp32 gradient buffer (same layout as the parameters)
and calling `on_segment(name, events)` after each group of layers has enqueued its last gradient kernel
(answer_head, fusion, text_encoder, stage4 ... stem) so a data-parallel driver can start that bucket's
RCCL all-reduce (on its own stream, after `events`) while earlier layers are still running.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional

import torch

from . import kernels as K
from . import layout as LY
from ._lib import call, dt, ptr


def _streams_overlap(a: "torch.cuda.Stream", b: "torch.cuda.Stream", cycles: int = 1_500_000) -> bool:
    """True when work on `b` really runs while `a` is busy.  The HIP runtime multiplexes streams onto a few hardware queues
    (GPU_MAX_HW_QUEUES, 4 by default, least-used queue first): two streams that landed on the same queue execute strictly one
    after the other.  Probe: a ~1 ms spin kernel on a, a tiny one on b; if b's finishes while a's is still running, they overlap."""
    ea, eb = torch.cuda.Event(), torch.cuda.Event()
    with torch.cuda.stream(a):
        torch.cuda._sleep(cycles)
        ea.record()
    with torch.cuda.stream(b):
        torch.cuda._sleep(1000)
        eb.record()
    eb.synchronize()
    ok = not ea.query()
    ea.synchronize()
    return ok


def pick_concurrent_streams(device, n: int, avoid=(), tries: int = 12):
    """n new streams that overlap with the current stream, with `avoid` and with each other (measured, see _streams_overlap).
    Measured on MI355X: a second engine in one process, or an engine built after an RCCL communicator, gets side streams from
    torch's pool that SHARE the main stream's hardware queue -- the weight-gradient stream then serialises with the data-gradient
    chain and a 13.2 ms train step takes 19 ms.  Falls back to plain new streams when no candidate passes."""
    main = torch.cuda.current_stream(device)
    if torch.cuda.is_current_stream_capturing():
        return tuple(torch.cuda.Stream(device=device) for _ in range(n))
    chosen, spare = [], []
    for _ in range(tries):
        if len(chosen) == n:
            break
        c = torch.cuda.Stream(device=device)
        if all(_streams_overlap(o, c) for o in [main, *avoid, *chosen]):
            chosen.append(c)
        else:
            spare.append(c)
    while len(chosen) < n:
        chosen.append(spare.pop(0) if spare else torch.cuda.Stream(device=device))
    return tuple(chosen)


class HipEngine:
    def __init__(self, cfg: dict, entries: List[LY.Entry], flat: torch.Tensor, buffers: Dict[str, torch.Tensor],
                 compute_dtype: torch.dtype):
        self.cfg = cfg
        self.entries = entries
        self.E = {e.name: e for e in entries}
        self.flat = flat
        self.buf = buffers
        self.dtype = compute_dtype
        # dropout seeds: rank << 44 | (seed_base + step_id) << 12 | site.  step_id advances on EVERY training forward (so the
        # reference's unchanged loop model(...) -> loss.backward() -> optimizer.step() draws fresh masks each step); the seeds of a
        # forward are stored in its tape, so its backward regenerates exactly those masks.  Data-parallel replicas carry their
        # rank in disjoint high bits (rank r at step s can never replay rank 0's masks of another step); the rank is read at
        # the first training forward, not here: the engine is rebuilt by every .to() and may predate init_process_group.
        self.seed_base = 0x5EED
        self.seed_rank = None
        self.step_id = 0
        self.wsrc = flat
        self._wsrc_fresh = False                  # one-shot: the bf16 copy wsrc was written by the last fused AdamW launch (trainer.py)
        self._wt: Dict[str, torch.Tensor] = {}
        self._wt_plan: Dict[str, tuple] = {}      # operands packed by begin_step (filled by _packT on first use)
        self._wt_table = None
        self._wt_buf = None
        # schedule switches: plain attributes (tools/ flip them for A/B measurements); the product never reads the environment
        self.defer_tail = True
        self._deferred = []
        self._keep = []
        self._wgq = []
        self.group_wgrad = True
        self._foldq = []                          # deferred folds of the LayerNorm / bias parameter gradients (K.fold_group at segment end)
        self._foldq2 = []
        self.defer_folds = True
        self.bias_offpath = False
        self.chain_trim = False                   # backward: pure bias column sums of the head / gate / projector and the text-gradient add leave the main-stream chain (measured neutral: 12.11 vs 12.09 ms, off)
        self.fuse_act_dgrad = True                # ReLU(+dropout) backward of a Linear -> ReLU -> Dropout -> Linear pair applied by the data-gradient GEMM's epilogue; the bias column sums leave the chain
        self.use_c64_fwd = False                  # 4-wave stage-1 patch kernel for forward / data gradient (superseded by the 8-wave one)
        self.use_c64p = True                      # 8-wave weights-resident stage-1 conv kernel
        self.fuse_se_pool = True                  # SE global-average-pool sums leave the last block's bn_apply (one read of the stage output less)
        self.fuse_se_bnred = True                 # the last block's bn2-backward column sums leave the SE backward apply pass (ditto)
        # BatchNorm finalize folded into the consumers (bf16 training schedule): statistics / backward sums travel as fixed-point
        # integer accumulators (order-free atomics), so conv -> [finalize] -> apply and reduce -> [finalize] -> apply lose their
        # middle launches (60 per step on the critical path; upper bound measured with tools/ab_nofinalize.py: -0.5 ms of 12.9)
        self.fuse_bn_finalize = True
        # round 4: stage-1 conv1 -> bn1 -> relu -> conv2 without the normalised tensor a1: conv2 (8-wave patch kernel) and its weight
        # gradient take conv1's raw output and apply scale / shift / ReLU to their LDS patch (one bn_apply pass less per block)
        self.fuse_bn_conv = True
        # round 4: 3x3 / stride-1 convs and data gradients with >= 256 output channels (stages 3, 4) on the 8-phase 224 x 256 tile
        # (csrc/gemm8p.hip: one 8-wave workgroup per CU, staggered wave groups; 196-of-224-row tiles = exact rounds at B = 512)
        self.use_conv8p = True
        # output channels a multiple of: 128 also routes stage 2 to the 448 x 128 form of the tile.  In isolation that form ties the
        # 128 x 128 window-loader tile there (155 / 143 vs 159 / 137 us); IN THE STEP it is faster: 12.35 -> 12.29 ms with both directions,
        # 12.31 forward only, 12.33 backward only (tools/ab_fusions.py, three alternations on one box; a second box: 12.73 -> 12.64)
        self.conv8p_n_multiple = 128
        self.conv8p_bwd_n_multiple = 128
        self.pack_off_main = True                 # begin_step's backward operands (transposed weights, stem helper) are packed on the weight-gradient stream
        self.hoist_cross = True                   # cross-attention: layer 0's query projection beside the CNN, layers >= 1's K / V path (fwd + bwd) on the text stream
        self.use_c64p_epi = True                  # stage-1 conv1 data gradients (identity addend + masks) on the 8-wave patch kernel
        self.fuse_hand_reduce = False             # ... and the previous block's bn2 (+ shortcut BN) backward sums in the epilogue of conv1's data gradient:
                                                  # measured neutral (12.21-12.27 vs 12.21-12.23 ms/step: two more epilogue streams cost what the pass saves) -> off
        self.fuse_bn1_reduce = True               # bn1-backward column sums in the epilogue of conv2's data gradient (conv8p, stage 1: the patch kernel)
        self.use_conv8p_bwd = True                # ... also for the data gradients (they run beside the weight-gradient stream)
        self._accbuf = None
        self._accpos = 0
        self._stem_fcoef = None
        self.fold_eval = True                     # inference (eval, no tape): Conv+BN folded, BN never runs as its own pass
        self._fold = None                         # (key, table, nd, blocks, wbuf, bbuf, views)
        self.stem_w = None
        # the text encoder (many tiny, latency-bound launches) runs on its own stream beside the CNN, forward and backward;
        # the CNN weight gradients run on a second side stream (off the critical path)
        self.side, self.side2 = pick_concurrent_streams(flat.device, 2) if flat.is_cuda else (None, None)
        self.two_streams = True
        self.wgrad_stream = True
        # test hook (tests/test_gpu_insitu.py): a dict here receives, per residual block, the intermediate gradients of its backward
        # (dout, dy2, dyd, da1, dy1, dx) so that every layer of a LIVE full-size bf16 step can be checked locally against fp32 math
        self.capture = None
        self.fuse_stem_eval = True                # inference (no tape): conv7x7 + BN + ReLU + MaxPool of the stem in one launch
        self.mark = None                          # measurement hook (tools/phase_times.py): called with a label at forward boundaries

    # ------------------------------------------------------------------ parameter access
    def P(self, name):                       # fp32 master, flat 1-D
        e = self.E[name]
        return self.flat[e.offset: e.offset + e.numel]

    def Wm(self, name):                      # [N][K] operand in compute dtype
        return LY.mat_of(self.wsrc, self.E[name])

    def _packT(self, key, pieces, rows, ld):
        """Data-gradient operand [rows][ld] assembled from pieces (src_offset, N, TT, C, col0, flip) of the fp32 master
        (out[c][col0 + t*N + n] = w[n][flip ? TT-1-t : t][c]).  The first step that needs `key` packs it on demand and
        records it in the plan; from then on begin_step() packs EVERY planned operand with one launch."""
        t = self._wt.get(key)
        if t is None:
            t = torch.empty((rows, ld), device=self.flat.device, dtype=self.dtype)
            for off, n, tt, c, col0, flip in pieces:
                K.pack_transpose(self.flat[off: off + n * tt * c].view(n, tt, c), self.dtype, out=t, ldo=ld, col0=col0, flip=flip)
            self._wt[key] = t
            self._wt_plan[key] = (tuple(pieces), rows, ld)
            self._wt_table = None
        return t

    def Wt(self, name):                      # [C][T*N] data-gradient operand, packed once per step
        e = self.E[name]
        n, c = e.shape[0], e.shape[1]
        tt = e.numel // (n * c)
        return self._packT(name, [(e.offset, n, tt, c, 0, False)], c, tt * n)

    def _pack_planned(self):
        """One vqa_pack_transpose_batch launch for every operand recorded by _packT in earlier steps."""
        if not self._wt_plan:
            return
        if self._wt_table is None or self._wt_table[0].device != self.flat.device:
            rowsd, views, dst, blk = [], [], 0, 0
            for key, (pieces, rows, ld) in self._wt_plan.items():
                for off, n, tt, c, col0, flip in pieces:
                    rowsd.append([off, dst, n, tt, c, ld, col0, int(flip), blk, 0])
                    blk += tt * ((n + 31) // 32) * ((c + 31) // 32)      # one workgroup per 32x32 tile of a tap
                views.append((key, dst, rows, ld))
                dst += (rows * ld + 7) // 8 * 8
            self._wt_table = (torch.tensor(rowsd, dtype=torch.int64).to(self.flat.device), views, dst, blk, len(rowsd))
            self._wt_buf = torch.empty(dst, device=self.flat.device, dtype=self.dtype)
        table, views, total, blk, nd = self._wt_table
        if self._wt_buf.dtype != self.dtype or self._wt_buf.device != self.flat.device:
            self._wt_buf = torch.empty(total, device=self.flat.device, dtype=self.dtype)
        call("vqa_pack_transpose_batch", K.dt(self.dtype), ptr(self.flat), ptr(self._wt_buf), ptr(table), nd, blk)
        for key, dst, rows, ld in views:
            self._wt[key] = self._wt_buf[dst: dst + rows * ld].view(rows, ld)

    def _fold_bn(self):
        """Eval-mode Conv+BN folding of every residual-block conv in one launch (BN eval semantics of
        models/cnn_backbone.py:164-197 with running statistics).  Returns {conv weight name: (folded [N][K] operand, bias [N])}."""
        pairs = []
        for s_ in range(1, 5):
            for b in range(2):
                p = f"image_encoder.stage{s_}.blocks.{b}"
                pairs += [(p + ".conv1.weight", p + ".bn1"), (p + ".conv2.weight", p + ".bn2")]
                if (p + ".downsample.0.weight") in self.E:
                    pairs.append((p + ".downsample.0.weight", p + ".downsample.1"))
        key = (self.flat.data_ptr(), str(self.dtype)) + tuple(self.buf[bn + ".running_mean"].data_ptr() for _, bn in pairs)
        if self._fold is None or self._fold[0] != key:
            rows, views, woff, boff, blk = [], {}, 0, 0, 0
            for wname, bn in pairs:
                e = self.E[wname]
                n = e.shape[0]
                k = e.numel // n
                rows.append([e.offset, self.E[bn + ".weight"].offset, self.E[bn + ".bias"].offset,
                             self.buf[bn + ".running_mean"].data_ptr(), self.buf[bn + ".running_var"].data_ptr(), n, k, woff, boff, blk])
                views[wname] = (woff, boff, n, k)
                woff += (n * k + 7) // 8 * 8
                boff += (n + 7) // 8 * 8
                blk += (n * k + 255) // 256
            dev = self.flat.device
            self._fold = (key, torch.tensor(rows, dtype=torch.int64).to(dev), len(rows), blk,
                          torch.empty(woff, device=dev, dtype=self.dtype), torch.empty(boff, device=dev, dtype=torch.float32), views)
        _, table, nd, blk, wbuf, bbuf, views = self._fold
        call("vqa_fold_bn_batch", K.dt(self.dtype), ptr(self.flat), ptr(wbuf), ptr(bbuf), ptr(table), nd, blk, 1e-5)
        return {w: (wbuf[wo: wo + n * k].view(n, k), bbuf[bo: bo + n]) for w, (wo, bo, n, k) in views.items()}

    def _make_stem_fcoef(self):
        gam, bet = self.P("image_encoder.stem.1.weight"), self.P("image_encoder.stem.1.bias")
        fcoef = torch.zeros((4, 64), device=self.flat.device, dtype=torch.float32)
        fcoef[2] = bet
        fcoef[3] = torch.where(gam.abs() > 1e-20, 1.0 / gam, torch.zeros_like(gam))
        return fcoef

    def begin_step(self, for_backward: bool = True):
        """Refresh the working copies of the weights (one cast of the whole flat buffer in bf16 mode)."""
        self._wt = {}
        self._stem_fcoef = None
        self._pack_ev = None
        if for_backward:                          # inference (no tape) needs neither the transposed weights nor the stem helper
            # Both are BACKWARD operands: on the weight-gradient stream (idle until the backward starts) they no longer sit in front of
            # the stem on the main stream (one 43 us pack launch + five small torch launches per step); backward() waits for the event
            s2 = self.side2 if (self.pack_off_main and self.two_streams and self.side2 is not None and self._wt_plan) else None
            if s2 is not None:
                ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream())     # the parameters are final (optimizer step) on this stream
                s2.wait_event(ev)
                with torch.cuda.stream(s2):
                    self._stem_fcoef = self._make_stem_fcoef()
                    self._pack_planned()
                    self._pack_ev = torch.cuda.Event(); self._pack_ev.record(s2)
            else:
                self._stem_fcoef = self._make_stem_fcoef() if self._wt_plan else None    # only once a backward has been seen
                self._pack_planned()
        if self.dtype == torch.bfloat16:
            if self.wsrc is self.flat or self.wsrc.numel() != self.flat.numel():
                self.wsrc = torch.empty(self.flat.numel(), device=self.flat.device, dtype=torch.bfloat16)
                self._wsrc_fresh = False
            # HipTrainer's AdamW launch writes this copy itself (adamw_copy_target) and vouches for it for exactly ONE begin_step, after
            # checking that nothing touched the parameters through torch in between (trainer._param_sig); every other caller casts
            if not self._wsrc_fresh:
                call("vqa_convert", 0, 1, ptr(self.flat), ptr(self.wsrc), self.flat.numel())
            self._wsrc_fresh = False
        else:
            self.wsrc = self.flat
        bk = 64 if self.dtype == torch.bfloat16 else 32
        kp = (147 + bk - 1) // bk * bk
        self.stem_kp = kp
        self.stem_w = K.pack_rows(LY.mat_of(self.flat, self.E["image_encoder.stem.0.weight"]), self.dtype, kp)
        self.stem_w2 = None
        if self.dtype == torch.bfloat16:            # operand of the dedicated bf16 stem kernel: [64][(c,r,s8)]
            self.stem_w2 = torch.empty((64, 192), device=self.flat.device, dtype=torch.bfloat16)
            call("vqa_stem_pack", ptr(self.P("image_encoder.stem.0.weight")), ptr(self.stem_w2))

    def adamw_copy_target(self):
        """bf16 operand buffer the fused AdamW kernel may write next to the fp32 parameters (None: fp32 schedule / not allocated yet)."""
        if self.dtype != torch.bfloat16 or self.wsrc is self.flat or self.wsrc.numel() != self.flat.numel():
            return None
        return self.wsrc


    def _seed(self):
        self._site += 1
        if self.seed_rank is None:
            self.seed_rank = torch.distributed.get_rank() if (torch.distributed.is_available() and torch.distributed.is_initialized()) else 0
        return (self.seed_rank << 44) | (((self.seed_base + self.step_id) & 0xFFFFFFFF) << 12) | self._site

    # ------------------------------------------------------------------ small op helpers
    def _bn_coef(self, prefix, stats, mt, C, count, training):
        if training:
            return K.bn_train_coef(stats, mt, C, count, self.P(prefix + ".weight"), self.P(prefix + ".bias"),
                                   self.buf[prefix + ".running_mean"], self.buf[prefix + ".running_var"],
                                   self.buf[prefix + ".num_batches_tracked"])
        return K.bn_eval_coef(C, self.P(prefix + ".weight"), self.P(prefix + ".bias"),
                              self.buf[prefix + ".running_mean"], self.buf[prefix + ".running_var"])

    # ---- fixed-point accumulators: ONE zeroed int64 buffer per forward (statistics) and per backward (column sums), sliced
    ACC_WORDS = 96 * 1024

    def _acc_reset(self):
        self._accbuf = torch.zeros(self.ACC_WORDS, device=self.flat.device, dtype=torch.int64)
        self._accpos = 0

    def _acc(self, words):
        words = (words + 1) // 2 * 2                       # 16-byte aligned slices
        if self._accbuf is None or self._accpos + words > self._accbuf.numel():
            self._accbuf = torch.zeros(max(self.ACC_WORDS, words), device=self.flat.device, dtype=torch.int64)
            self._accpos = 0
        t = self._accbuf[self._accpos: self._accpos + words]
        self._accpos += words
        return t

    def _bn_params(self, prefix, training=True):
        return (self.P(prefix + ".weight"), self.P(prefix + ".bias"), self.buf[prefix + ".running_mean"], self.buf[prefix + ".running_var"],
                self.buf[prefix + ".num_batches_tracked"])

    def _c64_ok(self, B, H, W, Cin, Cout, R, stride, wgrad=False):
        if not wgrad and not self.use_c64_fwd:
            return False          # since the LDS-DMA rewrite the generic implicit GEMM is as fast for forward / data gradient
        ok = self.dtype == torch.bfloat16 and Cin == 64 and Cout == 64 and R == 3 and stride == 1
        return ok and (K.c64w_blocks(B, H, W) if wgrad else K.c64_blocks(B, H, W)) > 0

    def _c128w_ok(self, B, H, W, Cin, Cout, stride):
        """128 -> 128 channel 3x3/1 conv on 28 x 28 maps (stage 2): the 8-wave LDS-DMA weight-gradient kernel."""
        return (self.dtype == torch.bfloat16 and Cin == 128 and Cout == 128 and stride == 1 and K.c128_wgrad_blocks(B, H, W) > 0)

    def _c64p_ok(self, B, H, W, Cin, Cout, R, stride):
        """64 -> 64 channel 3x3/1 conv without epilogue inputs: the 8-wave LDS-DMA patch kernel (156 vs 215 us forward, 152 vs 188 us
        data gradient at B=512)."""
        return (self.dtype == torch.bfloat16 and Cin == 64 and Cout == 64 and R == 3 and stride == 1 and K.c64p_blocks(B, H, W) > 0
                and self.use_c64p)

    def _c8p_ok(self, B, H, W, Cin, Cout, R, stride, pad, bwd=False):
        """3x3 / pad 1 conv (stride 1 or 2) or stride-1 data gradient routed to the 8-phase tile: bf16, output channels a multiple of 256
        (224 x 256 tile: stages 3, 4) or of 128 (448 x 128 tile: stage 2)."""
        return (self.use_conv8p and self.dtype == torch.bfloat16 and R == 3 and stride in (1, 2) and pad == 1 and Cout % (self.conv8p_bwd_n_multiple if bwd else self.conv8p_n_multiple) == 0
                and K.conv8p_ok(B, H, W, Cin, Cout))

    def _wflip(self, name):                  # [Cin][(2-r,2-s)][Cout] operand of the stride-1 data gradient as a plain 3x3 conv
        e = self.E[name]
        n, c = e.shape[0], e.shape[1]
        return self._packT(name + ".flip", [(e.offset, n, 9, c, 0, True)], c, 9 * n)

    def _conv(self, x, B, H, W, Cin, wname, Cout, R, stride, pad, stats, acc=False):
        """acc: the BatchNorm sums go to a fixed-point accumulator (returned in the slab's place, mtiles = -1) when the kernel that
        runs this shape supports it; otherwise the usual slab comes back and the caller runs the finalize launch."""
        Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
        M = B * Ho * Wo
        geom = (B, H, W, Cin, Ho, Wo, R, R, stride, pad)
        if self._c64p_ok(B, H, W, Cin, Cout, R, stride):
            a = self._acc(K.L.count("vqa_bn_acc_words", 2, Cout)) if (acc and stats) else None
            y, st, mt = K.conv3x3_c64p(x, self.Wm(wname), B, H, W, want_stats=stats, stats_acc=a)
            return y, st, (-1 if a is not None else mt), geom, Ho, Wo
        if self._c64_ok(B, H, W, Cin, Cout, R, stride):
            y, st, mt = K.conv3x3_c64(x, self.Wm(wname), B, H, W, want_stats=stats)
            return y, st, mt, geom, Ho, Wo
        if self._c8p_ok(B, H, W, Cin, Cout, R, stride, pad) and (acc or not stats):
            a = self._acc(K.L.count("vqa_bn_acc_words", 2, Cout)) if stats else None
            y = K.conv8p(x, self.Wm(wname), B, H, W, Cin, Cout, stride=stride, stats_acc=a)
            return y, a, (-1 if a is not None else 0), geom, Ho, Wo
        a = self._acc(K.L.count("vqa_bn_acc_words", 2, Cout)) if (acc and stats) else None
        y, st, mt = K.igemm(x, self.Wm(wname), M, Cout, R * R * Cin, geom, dtype=self.dtype, want_stats=stats, stats_acc=a)
        return y, st, (-1 if a is not None else mt), geom, Ho, Wo

    def _lin(self, x, wname, bname=None, relu=0, p=0.0, seed=0, addend=None):
        e = self.E[wname]
        N, Kin = e.shape[0], e.shape[1]
        M = x.shape[0]
        if N % 8:                                  # rare (e.g. num_answers=10): run the GEMM on a zero-padded copy of W
            Np = (N + 7) // 8 * 8
            w = torch.zeros((Np, Kin), device=x.device, dtype=self.dtype)
            w[:N] = self.Wm(wname)
            b = None
            if bname:
                b = torch.zeros((Np,), device=x.device, dtype=torch.float32)
                b[:N] = self.P(bname)
            assert addend is None and p == 0.0
            out, _, _ = K.igemm(x, w, M, Np, Kin, K.linear_geom(M, Kin), dtype=self.dtype, bias=b, relu=relu)
            return out[:, :N].contiguous()
        out, _, _ = K.igemm(x, self.Wm(wname), M, N, Kin, K.linear_geom(M, Kin), dtype=self.dtype,
                            bias=self.P(bname) if bname else None, relu=relu, drop_p=p, drop_seed=seed, addend=addend)
        return out

    def _lin_multi(self, x, wnames):
        """One GEMM over several bias-free Linear weights that sit back to back in the flat buffer ([sum N][K] view)."""
        e0 = self.E[wnames[0]]
        Kin = e0.shape[1]
        N = sum(self.E[w].shape[0] for w in wnames)
        M = x.shape[0]
        w = self.wsrc[e0.offset: e0.offset + N * Kin].view(N, Kin)
        out, _, _ = K.igemm(x, w, M, N, Kin, K.linear_geom(M, Kin), dtype=self.dtype)
        return out

    def _lin_multi_bwd(self, dz, x_in, wnames, G, addend=None):
        e0 = self.E[wnames[0]]
        Kin = e0.shape[1]
        N = sum(self.E[w].shape[0] for w in wnames)
        M = dz.shape[0]
        self._wgrad_linear(dz, x_in, G[e0.offset: e0.offset + N * Kin].view(N, Kin), M, N, Kin)
        wt = self._packT(wnames[0] + ".multiT%d" % len(wnames), [(e0.offset, N, 1, Kin, 0, False)], Kin, N)
        dx, _, _ = K.igemm(dz, wt, M, Kin, N, K.linear_geom(M, N), dtype=self.dtype, addend=addend)
        return dx

    def _adjacent(self, wnames):
        es = [self.E[w] for w in wnames]
        return all(es[i].offset + es[i].numel == es[i + 1].offset and es[i].shape[1] == es[0].shape[1] for i in range(len(es) - 1)) \
            and all(e.shape[0] % 8 == 0 for e in es)

    # ---- token-side Linear weight gradients are collected and launched up to 8 at a time (vqa_wgrad_group): each is a 40-160
    #      workgroup, latency-bound launch on its own.  The queue is flushed on the stream that produced its operands (when it is
    #      full, before a gradient segment is reported, before the stream context changes) and holds references to dz / x until then.
    def _wgrad_linear(self, dz, x_in, dw, M, N, Kin):
        if self.group_wgrad and K.wgrad_group_ok(self.dtype, M, N, Kin):
            self._wgq.append((dz, x_in, dw, M, N, Kin))
            if len(self._wgq) == 8:
                self._flush_wgq()
        else:
            K.wgrad(dz, x_in, dw, M, N, Kin, K.linear_geom(M, Kin), dtype=self.dtype)

    def _flush_wgq(self):
        q, self._wgq = self._wgq, []
        if len(q) == 1:
            dz, x_in, dw, M, N, Kin = q[0]
            self._off_path([dz, x_in], lambda: K.wgrad(dz, x_in, dw, M, N, Kin, K.linear_geom(M, Kin), dtype=self.dtype))
        elif q:                                   # side outputs: off the data-gradient chain (second side stream, like the conv dW)
            self._off_path([t for j in q for t in (j[0], j[1])], lambda: K.wgrad_group(q, dtype=self.dtype))

    def _lin_act_bwd(self, dz, x_in, wname, bname_prev, G, p):
        """Backward of `x_in = dropout(relu(linear_prev(.) + bias_prev)); y = linear(x_in)` from dz = dL/dy down to the gradient at
        linear_prev's output: dW += dz^T x_in, d(pre-activation) = (dz W) * (x_in > 0) / (1 - p) out of ONE launch on the data-gradient
        chain (vqa_linear_dgrad_act; x_in > 0 encodes both masks as in vqa_bias_act_bwd), bias_prev's gradient = its column sums taken
        off the chain.  Bit-equal to _lin_bwd + _act_bwd."""
        e = self.E[wname]
        N, Kin = e.shape[0], e.shape[1]
        if not self.fuse_act_dgrad or N % 8 or Kin % 8:
            return self._act_bwd(self._lin_bwd(dz, x_in, wname, G), x_in, bname_prev, G, p, 0)
        M = dz.shape[0]
        self._wgrad_linear(dz, x_in, LY.mat_of(G, e), M, N, Kin)
        dpre = K.linear_dgrad_act(dz, self.Wt(wname), M, Kin, N, dtype=self.dtype, outact=x_in, drop_p=p)
        self._act_bwd(dpre, None, bname_prev, G, 0.0, 0, off_chain=True)
        return dpre

    def _lin_bwd(self, dz, x_in, wname, G, need_dx=True, addend=None):
        """dW += dz^T x ; returns dx = dz W (+ addend)."""
        e = self.E[wname]
        N, Kin = e.shape[0], e.shape[1]
        M = dz.shape[0]
        if N % 8:
            Np = (N + 7) // 8 * 8
            dzp = torch.zeros((M, Np), device=dz.device, dtype=self.dtype)
            dzp[:, :N] = dz
            dwp = torch.zeros((Np, Kin), device=dz.device, dtype=torch.float32)
            K.wgrad(dzp, x_in, dwp, M, Np, Kin, K.linear_geom(M, Kin), dtype=self.dtype)
            LY.mat_of(G, e).add_(dwp[:N])
            if not need_dx:
                return None
            wp = torch.zeros((Np, 1, Kin), device=dz.device, dtype=torch.float32)
            wp[:N, 0] = LY.mat_of(self.flat, e)
            dx, _, _ = K.igemm(dzp, K.pack_transpose(wp, self.dtype), M, Kin, Np, K.linear_geom(M, Np), dtype=self.dtype, addend=addend)
            return dx
        self._wgrad_linear(dz, x_in, LY.mat_of(G, e), M, N, Kin)
        if not need_dx:
            return None
        dx, _, _ = K.igemm(dz, self.Wt(wname), M, Kin, N, K.linear_geom(M, N), dtype=self.dtype, addend=addend)
        return dx

    def _act_bwd(self, dout, outact, bname, G, p, seed, off_chain=False):
        """Gradient at the pre-activation of `linear(+bias)(+relu)(+dropout)`; accumulates the bias gradient."""
        M, N = dout.shape
        need_dz = (outact is not None) or p > 0.0
        dz = torch.empty_like(dout) if need_dz else None
        dbias = None
        if bname:
            e = self.E[bname]
            dbias = G[e.offset: e.offset + e.numel]
        if need_dz or dbias is not None:
            def launch():
                ws = None
                if dbias is not None:      # fixed-order column sums (bit-reproducible): per-workgroup rows + index-order fold
                    ws = torch.empty((K.reduce_ws("vqa_bias_act_bwd_ws", dt(dout), M, N),), device=dout.device, dtype=torch.float32)
                defer = int(ws is not None and self.defer_folds)
                call("vqa_bias_act_bwd", dt(dout), ptr(dout), ptr(outact), ptr(dz), ptr(dbias), M, N, float(p), int(seed), ptr(ws), defer)
                if defer:                  # the bias gradient is only read by the optimizer: fold it with the rest of the segment
                    (self._foldq if on_chain else self._foldq2).append((ws, 0, K.L.count("vqa_bias_act_bwd_fold_rows", dt(dout), M, N), N, N, dbias, N, None))
            on_chain = need_dz or not (self.bias_offpath or off_chain)
            if on_chain:
                launch()
            else:                          # a pure column sum (no mask, no dropout): a side output, off the data-gradient chain
                self._off_path([dout], launch)
        return dz if need_dz else dout

    def _ln(self, x, prefix, p=0.0, seed=0, addrow=None, period=1):
        return K.layernorm_fwd(x, self.P(prefix + ".weight"), self.P(prefix + ".bias"), drop_p=p, seed=seed,
                               addrow=addrow, period=period)

    def _ln_bwd(self, dout, x, prefix, stats, G, addend=None, p=0.0, seed=0, dadd=None, period=1):
        eg, eb = self.E[prefix + ".weight"], self.E[prefix + ".bias"]
        return K.layernorm_bwd(dout, x, self.P(prefix + ".weight"), stats, G[eg.offset: eg.offset + eg.numel],
                               G[eb.offset: eb.offset + eb.numel], addend=addend, drop_p=p, seed=seed, dadd=dadd, period=period,
                               foldq=self._foldq if self.defer_folds else None)

    def _off_path(self, tensors, fn, defer=False):
        """Run fn (a weight-gradient launch) on the second side stream: it only needs `tensors` (already produced on the
        current stream) and writes its own slice of G, so it may overlap the data-gradient chain.
        defer: keep it back until _flush_deferred() -- the last residual block's weight gradients are released when the stem
        backward starts, so that the (latency-bound) fused stem weight-gradient kernel does not run alone at the end of the step."""
        if not (self.wgrad_stream and self.two_streams and self.side2 is not None):
            return fn()
        if defer and self.defer_tail:
            self._deferred.append((tensors, fn))
            return
        cur = torch.cuda.current_stream()
        ev = torch.cuda.Event(); ev.record(cur)
        self.side2.wait_event(ev)
        # `tensors` were allocated on the current stream and are read on the side stream: they are kept alive until backward has
        # joined the side stream again (_keep, released at the end of backward) instead of record_stream()'ed.  record_stream
        # defers a block's reuse until an event polled at some LATER allocation completes -- with the host a step ahead of the GPU
        # that made the caching allocator's alloc / free sequence timing-dependent: it never reached a steady state (4 hipMalloc per
        # step after 40 steps, 25 ms host stalls, reserved memory creeping up by ~3 MB per step).  Held references free in a fixed
        # order, so every step replays the same allocation sequence.
        self._keep.extend(tensors)
        with torch.cuda.stream(self.side2):
            fn()

    def _flush_deferred(self):
        d, self._deferred = self._deferred, []
        for tensors, fn in d:
            self._off_path(tensors, fn)

    def _join_off_path(self):
        self._flush_deferred()
        if self.wgrad_stream and self.two_streams and self.side2 is not None:
            ev = torch.cuda.Event(); ev.record(self.side2)
            torch.cuda.current_stream().wait_event(ev)

    def _gslice(self, G, name):
        e = self.E[name]
        return G[e.offset: e.offset + e.numel]

    # ------------------------------------------------------------------ forward
    def forward(self, images: torch.Tensor, token_ids: torch.Tensor, maskf: Optional[torch.Tensor], training: bool,
                want_aux: bool = False, need_tape: bool = True, lowp_logits: bool = False):
        """lowp_logits: return the logits in the compute dtype (the trainer's loss kernel reads bf16 and leaves the fp32 copy itself)."""
        cfg, T = self.cfg, self.dtype
        self._site = 0
        if training:
            self.step_id += 1
        self.begin_step(for_backward=need_tape)
        if training and self.fuse_bn_finalize and T == torch.bfloat16:
            self._acc_reset()                     # one memset for every BatchNorm accumulator of this forward
        tape: dict = {"training": training, "B": images.shape[0]}
        B, _, IH, IW = images.shape
        pdrop = cfg["dropout"] if training else 0.0
        phead = cfg["answer_dropout"] if training else 0.0
        dev = images.device

        # The stem's three launches go out FIRST (0.6 ms of GPU work): a caller that synchronises every step (training/train.py:211
        # loss.item()) starts each forward with an idle GPU, and the ~100 tiny text-encoder launches issued ahead of the stem left
        # it idle for the millisecond the host needs to enqueue them.  The side stream still waits only for what preceded the stem.
        main = torch.cuda.current_stream()
        use_side = self.two_streams and self.side is not None
        if use_side:
            ev0 = torch.cuda.Event(); ev0.record(main)
        # ---- stem: conv7x7/2 (from the NCHW fp32 image) + BN + ReLU + maxpool, A1
        H1, W1 = (IH + 6 - 7) // 2 + 1, (IW + 6 - 7) // 2 + 1
        M = B * H1 * W1
        sgeom = (B, IH, IW, 3, H1, W1, 7, 7, 2, 3)
        Hp, Wp = (H1 + 2 - 3) // 2 + 1, (W1 + 2 - 3) // 2 + 1
        x = None
        if self.fuse_stem_eval and not training and not need_tape and self.stem_w2 is not None:
            # inference: the whole stem in one launch, the 112 x 112 conv output is never stored (no argmax: there is no backward)
            x = K.stem_conv_pool(images, self.stem_w2, self._bn_coef("image_encoder.stem.1", None, 0, 64, M, False), B, IH, IW)
        if x is None:
            if self.stem_w2 is not None and K.stem_conv_blocks(B, IH, IW) > 0:
                y, st, mt = K.stem_conv(images, self.stem_w2, B, IH, IW, training)
            else:
                y, st, mt = K.igemm(images, self.stem_w, M, 64, self.stem_kp, sgeom, dtype=T, loader=K.LOADER_STEM, want_stats=training)
            coef = self._bn_coef("image_encoder.stem.1", st, mt, 64, M, training)
            x = torch.empty((B * Hp * Wp, 64), device=dev, dtype=T)
            idx = torch.empty((B * Hp * Wp, 64), device=dev, dtype=torch.uint8)
            call("vqa_stem_pool_fwd", dt(T), ptr(y), ptr(coef), ptr(x), ptr(idx), B, H1, W1, 64)
            tape["stem"] = dict(images=images, y=y, coef=coef, idx=idx, geom=sgeom, H1=H1, W1=W1)
        H, W, C = Hp, Wp, 64
        # ---- text encoder, A6-A8 (on the side stream; joined before fusion).  Issued behind stage 1: by then the GPU holds > 1 ms of
        #      queued work, which covers the millisecond the host spends on these ~100 launches (see the stem note above).
        d, heads = cfg["embed_dim"], cfg["num_attention_heads"]
        hd = d // heads
        Bt, L = token_ids.shape
        rows = Bt * L
        pe = self.buf["text_encoder.positional_encoding.pe"]
        if L > pe.shape[1]:
            raise RuntimeError(f"sequence length {L} exceeds max_question_length {pe.shape[1]}")
        text = {}

        def issue_text():
            if use_side:
                self.side.wait_event(ev0)                # weights cast + everything earlier on main is visible to the side stream
            with torch.cuda.stream(self.side if use_side else main):
                emb_e = self.E["text_encoder.token_embedding.weight"]
                sd0 = self._seed()
                xt = torch.empty((rows, d), device=dev, dtype=T)
                call("vqa_embed_fwd", dt(T), ptr(token_ids), ptr(self.P(emb_e.name)), ptr(pe), ptr(xt), rows, L, d, emb_e.shape[0],
                     math.sqrt(d), float(pdrop), sd0)
                tape["embed"] = dict(ids=token_ids, seed=sd0, p=pdrop)
                tape["tlayers"] = []
                for l in range(cfg["num_transformer_layers"]):
                    p = f"text_encoder.layers.{l}"
                    rec = self._attn_block_fwd(xt, xt, None, p + ".norm1", None, p + ".self_attention", maskf, Bt, L, L, heads, hd, pdrop,
                                               p + ".norm2", p + ".ffn.fc1", p + ".ffn.fc2", self_attn=True)
                    tape["tlayers"].append(rec)
                    xt = rec["out"]
                text["enc"], enc_st = self._ln(xt, "text_encoder.final_norm")
                tape["final_norm"] = dict(x=xt, st=enc_st)
                if self.hoist_cross and cfg["num_cross_layers"] >= 1:
                    # the first cross-attention layer's norm_query + W_q need the text features only: issued here, beside the CNN,
                    # instead of behind the CNN on the fusion chain (the critical path between CNN forward and CNN backward)
                    p0 = "fusion.cross_attention.layers.0"
                    text["q0"] = self._cross_q_path(text["enc"], p0 + ".norm_query", p0 + ".cross_attention")
                text["ev"] = torch.cuda.Event(); text["ev"].record()

        if self.mark: self.mark("forward: stem")

        # ---- residual stages, A2-A5
        folded = self._fold_bn() if (self.fold_eval and not training and not need_tape) else None
        tape["stages"] = []
        for s, Cout in enumerate(LY.STAGE_CHANNELS, start=1):
            srec = {"blocks": []}
            for b in range(2):
                p = f"image_encoder.stage{s}.blocks.{b}"
                stride = 2 if (b == 0 and s > 1) else 1
                Cin = C
                if folded is not None:
                    # inference: a1 = relu(conv1'(x) + b1); out = relu(conv2'(a1) + b2 + shortcut) -- two or three launches, no BN pass
                    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
                    M = B * Ho * Wo
                    w1, b1 = folded[p + ".conv1.weight"]
                    w2, b2 = folded[p + ".conv2.weight"]
                    a1, _, _ = K.igemm(x, w1, M, Cout, 9 * Cin, (B, H, W, Cin, Ho, Wo, 3, 3, stride, 1), dtype=T, bias=b1, relu=1)
                    res = x
                    if (p + ".downsample.0.weight") in self.E:
                        wd, bd = folded[p + ".downsample.0.weight"]
                        res, _, _ = K.igemm(x, wd, M, Cout, Cin, (B, H, W, Cin, Ho, Wo, 1, 1, stride, 0), dtype=T, bias=bd)
                    out, _, _ = K.igemm(a1, w2, M, Cout, 9 * Cout, (B, Ho, Wo, Cout, Ho, Wo, 3, 3, 1, 1), dtype=T, bias=b2, addend=res, relu=2)
                    x, H, W, C = out, Ho, Wo, Cout
                    continue
                facc = training and self.fuse_bn_finalize and T == torch.bfloat16
                y1, st1, mt1, g1, Ho, Wo = self._conv(x, B, H, W, Cin, p + ".conv1.weight", Cout, 3, stride, 1, training, acc=facc)
                M = B * Ho * Wo
                fuse12 = (mt1 < 0 and self.fuse_bn_conv and need_tape and self._c64p_ok(B, Ho, Wo, Cout, Cout, 3, 1)
                          and K.c64w_bn_ok(B, Ho, Wo))
                if fuse12:                                   # conv2 normalises conv1's raw output in its LDS patch: a1 never exists
                    a1 = None
                    a2 = self._acc(K.L.count("vqa_bn_acc_words", 2, Cout))
                    y2, st2, _, c1 = K.conv3x3_c64p_bn(y1, st1, self._bn_params(p + ".bn1"), self.Wm(p + ".conv2.weight"), B, Ho, Wo, M,
                                                       want_stats=True, stats_acc=a2)
                    mt2, g2 = -1, (B, Ho, Wo, Cout, Ho, Wo, 3, 3, 1, 1)
                elif mt1 < 0:                                # statistics in a fixed-point accumulator: finalize + apply in one launch
                    a1, c1, _, _, _ = K.bn_apply_acc(y1, st1, self._bn_params(p + ".bn1"), Cout, True, B, Ho * Wo, M)
                else:
                    c1 = self._bn_coef(p + ".bn1", st1, mt1, Cout, M, training)
                    a1 = K.bn_apply(y1, c1, Cout, relu=True)
                if not fuse12:
                    y2, st2, mt2, g2, _, _ = self._conv(a1, B, Ho, Wo, Cout, p + ".conv2.weight", Cout, 3, 1, 1, training, acc=facc)
                c2 = None if mt2 < 0 else self._bn_coef(p + ".bn2", st2, mt2, Cout, M, training)
                rec = dict(p=p, x=x, y1=y1, c1=c1, a1=a1, y2=y2, c2=c2, g1=g1, g2=g2, M=M, Cin=Cin, Cout=Cout)
                # the stage's last block hands the SE pooling sums over (its output is the SE input)
                pool_here = (b == 1 and self.fuse_se_pool and (f"image_encoder.stage{s}.attention.se.fc1.weight") in self.E
                             and K.L.count("vqa_bn_apply_pool_chunks", dt(T), Ho * Wo, Cout) > 0 and B <= 65535)
                if (p + ".downsample.0.weight") in self.E:
                    yd, std, mtd, gd, _, _ = self._conv(x, B, H, W, Cin, p + ".downsample.0.weight", Cout, 1, stride, 0, training,
                                                        acc=facc and mt2 < 0)
                    if mt2 < 0 and mtd < 0:                  # both BatchNorms finalized inside the residual-add pass
                        out, c2, cd, _, _ = K.bn_apply_acc(y2, st2, self._bn_params(p + ".bn2"), Cout, True, B, Ho * Wo, M, res=yd, racc=std,
                                                           rbn=self._bn_params(p + ".downsample.1"))
                        rec["c2"] = c2
                    else:
                        if mt2 < 0:
                            raise RuntimeError("fixed-point statistics of conv2 without those of the shortcut conv")
                        cd = self._bn_coef(p + ".downsample.1", std, mtd, Cout, M, training)
                        out = K.bn_apply(y2, c2, Cout, relu=True, res=yd, rcoef=cd)
                    rec.update(yd=yd, cd=cd, gd=gd)
                elif mt2 < 0:
                    out, c2, _, pool_part, pool_chunks = K.bn_apply_acc(y2, st2, self._bn_params(p + ".bn2"), Cout, True, B, Ho * Wo, M, res=x,
                                                                        pool=pool_here)
                    rec["c2"] = c2
                    if pool_here:
                        srec["pool"] = (pool_part, pool_chunks)
                elif pool_here:
                    out, pool_part, pool_chunks = K.bn_apply_pool(y2, c2, Cout, True, B, Ho * Wo, res=x)
                    srec["pool"] = (pool_part, pool_chunks)
                else:
                    out = K.bn_apply(y2, c2, Cout, relu=True, res=x)
                rec["out"] = out
                srec["blocks"].append(rec)
                x, H, W, C = out, Ho, Wo, Cout
            ap = f"image_encoder.stage{s}.attention"
            if (ap + ".se.fc1.weight") in self.E:
                Cr = self.E[ap + ".se.fc1.weight"].shape[0]
                pooled = torch.empty((B, C), device=dev, dtype=torch.float32)
                hidden = torch.empty((B, Cr), device=dev, dtype=torch.float32)
                scale = torch.empty((B, C), device=dev, dtype=torch.float32)
                out = torch.empty_like(x)
                pool_part, pool_chunks = srec.pop("pool", (None, 0))
                call("vqa_se_fwd", dt(T), ptr(x), ptr(self.P(ap + ".se.fc1.weight")), ptr(self.P(ap + ".se.fc2.weight")),
                     ptr(pooled), ptr(hidden), ptr(scale), ptr(out), B, H * W, C, Cr, ptr(pool_part), pool_chunks)
                srec["se"] = dict(x=x, pooled=pooled, hidden=hidden, scale=scale, Cr=Cr, HW=H * W, C=C)
                x = out
            if (ap + ".spatial.conv.weight") in self.E:
                pooled2 = torch.empty((B * H * W, 2), device=dev, dtype=torch.float32)
                amax = torch.empty((B * H * W,), device=dev, dtype=torch.int32)
                amap = torch.empty((B * H * W,), device=dev, dtype=torch.float32)
                out = torch.empty_like(x)
                call("vqa_spatial_fwd", dt(T), ptr(x), ptr(self.P(ap + ".spatial.conv.weight")), ptr(pooled2), ptr(amax),
                     ptr(amap), ptr(out), B, H, W, C)
                srec["spatial"] = dict(x=x, pooled2=pooled2, amax=amax, amap=amap, H=H, W=W, C=C)
                x = out
            tape["stages"].append(srec)
            if s == 1:
                issue_text()
            if self.mark: self.mark(f"forward: stage{s}")
        feat = x                                  # [B*Hf*Wf, 512] == tokens of the projector (NHWC makes the permute free)
        Hf, Wf, Cf = H, W, C
        ntok = Hf * Wf
        tape["feat"] = dict(Hf=Hf, Wf=Wf, Cf=Cf)

        # ---- fusion, A9-A11
        enc, ev_txt = text["enc"], text["ev"]
        if use_side:
            main.wait_event(ev_txt)
        pj = "fusion.image_projector.projection"
        pz = self._lin(feat, pj + ".0.weight", pj + ".0.bias")
        sdp = self._seed()
        posemb = self.P("fusion.image_projector.position_embedding")
        if ntok * d > posemb.numel():            # the reference fails the same way (broadcast error at models/fusion.py:110)
            raise RuntimeError(f"{ntok} image tokens but position_embedding holds {posemb.numel() // d} (num_image_tokens)")
        img, img_st = self._ln(pz, pj + ".1", p=pdrop, seed=sdp, addrow=posemb, period=ntok)
        tape["proj"] = dict(feat=feat, pz=pz, st=img_st, seed=sdp, p=pdrop, ntok=ntok)
        q = enc
        tape["clayers"] = []
        probs_all = []
        ncl = cfg["num_cross_layers"]
        pre_kv = [None] * ncl
        if self.hoist_cross and use_side and ncl > 1:
            # every layer's K / V come from the SAME image tokens: layers >= 1 are projected on the (now idle) text stream while
            # layer 0 runs here.  img lives on this stream and is read there; the results live there and are read here: both are
            # ordered by the events below and kept alive by the tape (forward-only: freed after the last wait)
            ev_img = torch.cuda.Event(); ev_img.record(main)
            self.side.wait_event(ev_img)
            with torch.cuda.stream(self.side):
                for l in range(1, ncl):
                    p = f"fusion.cross_attention.layers.{l}"
                    pre_kv[l] = (self._cross_kv_path(img, p + ".norm_kv", p + ".cross_attention", d), torch.cuda.Event())
                    pre_kv[l][1].record()
        for l in range(ncl):
            p = f"fusion.cross_attention.layers.{l}"
            pkv = None
            if pre_kv[l] is not None:
                pkv, ev_kv = pre_kv[l]
                main.wait_event(ev_kv)
            rec = self._attn_block_fwd(q, img, None, p + ".norm_query", p + ".norm_kv", p + ".cross_attention", None, Bt, L, ntok,
                                       heads, hd, pdrop, p + ".norm_ffn", p + ".ffn.0", p + ".ffn.3", self_attn=False,
                                       pre_q=text.get("q0") if l == 0 else None, pre_kv=pkv)
            tape["clayers"].append(rec)
            probs_all.append(rec["probs"])
            q = rec["out"]
        cat = torch.empty((Bt, 2 * d), device=dev, dtype=T)
        call("vqa_masked_pool_pair_fwd", dt(T), ptr(q), ptr(enc), ptr(maskf), ptr(cat), Bt, L, d)      # [attended | text] means, one launch
        fused_pre = torch.empty((Bt, d), device=dev, dtype=T)
        z = None
        if cfg["use_gating"]:
            z = self._lin(cat, "fusion.gate.gate.0.weight", "fusion.gate.gate.0.bias")
            call("vqa_gate_fwd", dt(T), ptr(z), ptr(cat), ptr(fused_pre), Bt, d)
        else:
            att = cat[:, :d].contiguous(); txt = cat[:, d:].contiguous()
            call("vqa_add", dt(T), ptr(att), ptr(txt), ptr(fused_pre), Bt * d)
        fused, fst = self._ln(fused_pre, "fusion.output_norm")
        tape["pool"] = dict(q=q, enc=enc, cat=cat, z=z, fused_pre=fused_pre, fst=fst, maskf=maskf, L=L, d=d)
        if self.mark: self.mark("forward: fusion")

        # ---- answer head, A12
        c = "answer_head.classifier"
        s1, s2 = self._seed(), self._seed()
        h1 = self._lin(fused, c + ".0.weight", c + ".0.bias", relu=1, p=phead, seed=s1)
        h2 = self._lin(h1, c + ".3.weight", c + ".3.bias", relu=1, p=phead, seed=s2)
        logits = self._lin(h2, c + ".6.weight", c + ".6.bias")
        tape["head"] = dict(fused=fused, h1=h1, h2=h2, s1=s1, s2=s2, p=phead)
        logits_f = logits.float() if (T != torch.float32 and not lowp_logits) else logits

        aux = None
        if want_aux:
            feat_nchw = torch.empty((B, Cf, Hf, Wf), device=dev, dtype=torch.float32)
            call("vqa_nhwc_to_nchw", dt(T), ptr(feat), ptr(feat_nchw), B, Hf * Wf, Cf)
            aux = {
                "image_features": feat_nchw,
                "text_features": enc.float().view(Bt, L, d),
                "text_pooled": cat[:, d:].float(),          # fusion's entry overrides the encoder's (vqa_model.py:303-309)
                "fused": fused.float(),
                "cross_attention_weights": probs_all,
                "image_projected": img.float().view(Bt, ntok, d),
                "attended_pooled": cat[:, :d].float(),
            }
        if not need_tape:
            tape = None
        return logits_f, aux, tape

    def _cross_q_path(self, q_in, norm_q, attn):
        """norm_query + W_q of a cross-attention layer (cross_attention.py:286 and the query projection inside CrossAttention): needs
        the query stream only."""
        nq, stq = self._ln(q_in, norm_q)
        return nq, stq, self._lin(nq, attn + ".W_q.weight")

    def _cross_kv_path(self, kv_in, norm_kv, attn, d):
        """norm_kv + W_k | W_v of a cross-attention layer (cross_attention.py:287): needs the image tokens only -- the same tensor for
        every layer of StackedCrossAttention (cross_attention.py:357-361), so layers >= 1 do not sit on the query chain."""
        nkv, stkv = self._ln(kv_in, norm_kv)
        wk, wv = attn + ".W_k.weight", attn + ".W_v.weight"
        if self._adjacent([wk, wv]):
            kv = self._lin_multi(nkv, [wk, wv])
            return nkv, stkv, kv, kv[:, d:], 2 * d, True
        return nkv, stkv, self._lin(nkv, wk), self._lin(nkv, wv), d, False

    def _attn_block_fwd(self, q_in, kv_in, _unused, norm_q, norm_kv, attn, kmask, B, Lq, Lk, heads, hd, p, norm_f, fc1, fc2, self_attn,
                        pre_q=None, pre_kv=None):
        """pre-norm attention + FFN block (TransformerEncoderLayer.forward text_encoder.py:373-399 and
        MultiHeadCrossAttention.forward cross_attention.py:285-299).  pre_q / pre_kv: the results of _cross_q_path / _cross_kv_path when
        the caller issued them earlier (on another stream, already joined)."""
        T = self.dtype
        d = heads * hd
        wq, wk, wv = attn + ".W_q.weight", attn + ".W_k.weight", attn + ".W_v.weight"
        if self_attn:
            nq, stq = self._ln(q_in, norm_q)
            nkv, stkv = nq, None
            fused = self._adjacent([wq, wk, wv])
            if fused:                                 # one [M][3d] GEMM; Q/K/V are column slices (row stride 3d)
                qkv = self._lin_multi(nq, [wq, wk, wv])
                Q, Kt, V, ldq, ldkv = qkv, qkv[:, d:], qkv[:, 2 * d:], 3 * d, 3 * d
            else:
                Q, Kt, V, ldq, ldkv = self._lin(nq, wq), self._lin(nkv, wk), self._lin(nkv, wv), d, d
        else:
            nq, stq, Q = pre_q if pre_q is not None else self._cross_q_path(q_in, norm_q, attn)
            nkv, stkv, Kt, V, ldkv, fused = pre_kv if pre_kv is not None else self._cross_kv_path(kv_in, norm_kv, attn, d)
            ldq = d
        probs = torch.empty((B, heads, Lq, Lk), device=Q.device, dtype=torch.float32)
        ctx = torch.empty((B * Lq, d), device=Q.device, dtype=T)
        sa = self._seed()
        if T == torch.bfloat16 and Lq <= 32 and Lk <= 160 and hd in (32, 64):
            call("vqa_attention_fwd_mfma", ptr(Q), ptr(Kt), ptr(V), ldq, ldkv, ldkv, ptr(kmask), ptr(probs), ptr(ctx), d, B, heads, Lq, Lk, hd,
                 float(p), sa)
        else:
            call("vqa_attention_fwd", dt(T), ptr(Q), ptr(Kt), ptr(V), ldq, ldkv, ldkv, ptr(kmask), ptr(probs), ptr(ctx), d, B, heads, Lq, Lk, hd,
                 float(p), sa)
        so = self._seed()
        x1 = self._lin(ctx, attn + ".W_o.weight", p=p, seed=so, addend=q_in)
        nf, stf = self._ln(x1, norm_f)
        s1, s2 = self._seed(), self._seed()
        h = self._lin(nf, fc1 + ".weight", fc1 + ".bias", relu=1, p=p, seed=s1)
        out = self._lin(h, fc2 + ".weight", fc2 + ".bias", p=p, seed=s2, addend=x1)
        return dict(q_in=q_in, kv_in=kv_in, nq=nq, stq=stq, nkv=nkv, stkv=stkv, Q=Q, K=Kt, V=V, ldq=ldq, ldkv=ldkv, fused=fused,
                    probs=probs, ctx=ctx, sa=sa, so=so,
                    x1=x1, nf=nf, stf=stf, h=h, s1=s1, s2=s2, out=out, p=p, norm_q=norm_q, norm_kv=norm_kv, attn=attn, norm_f=norm_f,
                    fc1=fc1, fc2=fc2, self_attn=self_attn, B=B, Lq=Lq, Lk=Lk, heads=heads, hd=hd)

    def _attn_block_bwd(self, rec, dout, G, dkv_addend=None, kv_side=False, addend_event=None):
        """Returns (d q_in, d kv_in, event) ; for self-attention d kv_in is folded into d q_in.
        kv_side: the K / V projection's data gradient and norm_kv's backward (they only feed the image-token gradient, which the query
        chain of the remaining layers does not need) run on the text stream; `event` then marks d kv_in.  addend_event: the event of
        the dkv_addend handed in by such a layer."""
        T = self.dtype
        p, B, Lq, Lk, heads, hd = rec["p"], rec["B"], rec["Lq"], rec["Lk"], rec["heads"], rec["hd"]
        d = heads * hd
        attn, fc1, fc2 = rec["attn"], rec["fc1"], rec["fc2"]
        dz2 = self._act_bwd(dout, None, fc2 + ".bias", G, p, rec["s2"])
        dz1 = self._lin_act_bwd(dz2, rec["h"], fc2 + ".weight", fc1 + ".bias", G, p)
        dnf = self._lin_bwd(dz1, rec["nf"], fc1 + ".weight", G)
        dx1 = self._ln_bwd(dnf, rec["x1"], rec["norm_f"], rec["stf"], G, addend=dout)
        dzo = self._act_bwd(dx1, None, None, G, p, rec["so"])
        dctx = self._lin_bwd(dzo, rec["ctx"], attn + ".W_o.weight", G)
        wq, wk, wv = attn + ".W_q.weight", attn + ".W_k.weight", attn + ".W_v.weight"
        ldq, ldkv, fused = rec["ldq"], rec["ldkv"], rec["fused"]
        dev = dctx.device
        if fused and rec["self_attn"]:
            dqkv = torch.empty((B * Lq, 3 * d), device=dev, dtype=T)
            dQ, dK, dV = dqkv, dqkv[:, d:], dqkv[:, 2 * d:]
        elif fused:
            dQ = torch.empty((B * Lq, d), device=dev, dtype=T)
            dkv = torch.empty((B * Lk, 2 * d), device=dev, dtype=T)
            dK, dV = dkv, dkv[:, d:]
        else:
            dQ = torch.empty((B * Lq, d), device=dev, dtype=T); dK = torch.empty((B * Lk, d), device=dev, dtype=T); dV = torch.empty((B * Lk, d), device=dev, dtype=T)
        if T == torch.bfloat16 and Lq <= 32 and Lk <= 160 and hd in (32, 64):
            call("vqa_attention_bwd_mfma", ptr(dctx), d, ptr(rec["Q"]), ptr(rec["K"]), ptr(rec["V"]), ldq, ldkv, ldkv, ptr(rec["probs"]),
                 ptr(dQ), ptr(dK), ptr(dV), ldq, ldkv, ldkv, B, heads, Lq, Lk, hd, float(p), rec["sa"])
        else:
            call("vqa_attention_bwd", dt(T), ptr(dctx), d, ptr(rec["Q"]), ptr(rec["K"]), ptr(rec["V"]), ldq, ldkv, ldkv, ptr(rec["probs"]),
                 ptr(dQ), ptr(dK), ptr(dV), ldq, ldkv, ldkv, B, heads, Lq, Lk, hd, float(p), rec["sa"])
        if rec["self_attn"]:
            if fused:
                dnq = self._lin_multi_bwd(dqkv, rec["nq"], [wq, wk, wv], G)
            else:
                dnq = self._lin_bwd(dQ, rec["nq"], wq, G)
                dnq = self._lin_bwd(dK, rec["nkv"], wk, G, addend=dnq)
                dnq = self._lin_bwd(dV, rec["nkv"], wv, G, addend=dnq)
            dq_in = self._ln_bwd(dnq, rec["q_in"], rec["norm_q"], rec["stq"], G, addend=dx1)
            return dq_in, None, None
        dnq = self._lin_bwd(dQ, rec["nq"], wq, G)
        main = torch.cuda.current_stream()
        if kv_side and fused:
            # the weight gradient is queued HERE (its operands dkv / nkv are this stream's; the queue is flushed on this stream) ...
            e0 = self.E[wk]
            Kin, N2 = e0.shape[1], 2 * d
            self._wgrad_linear(dkv, rec["nkv"], G[e0.offset: e0.offset + N2 * Kin].view(N2, Kin), dkv.shape[0], N2, Kin)
            wt = self._packT(wk + ".multiT2", [(e0.offset, N2, 1, Kin, 0, False)], Kin, N2)
            ev = torch.cuda.Event(); ev.record(main)
            self.side.wait_event(ev)
            self._keep.append(dkv)                    # allocated here, read there
            with torch.cuda.stream(self.side):        # ... the data gradient and the LayerNorm backward over the 49-token rows run there
                dnkv, _, _ = K.igemm(dkv, wt, dkv.shape[0], Kin, N2, K.linear_geom(dkv.shape[0], N2), dtype=self.dtype)
                if addend_event is not None:
                    self.side.wait_event(addend_event)
                dkv_in = self._ln_bwd(dnkv, rec["kv_in"], rec["norm_kv"], rec["stkv"], G, addend=dkv_addend)
                ev_kv = torch.cuda.Event(); ev_kv.record()
            self._keep.extend([dnkv, dkv_in])         # allocated there, read here (as the next layer's addend): alive until the join
            dq_in = self._ln_bwd(dnq, rec["q_in"], rec["norm_q"], rec["stq"], G, addend=dx1)
            return dq_in, dkv_in, ev_kv
        if fused:
            dnkv = self._lin_multi_bwd(dkv, rec["nkv"], [wk, wv], G)
        else:
            dnkv = self._lin_bwd(dK, rec["nkv"], wk, G)
            dnkv = self._lin_bwd(dV, rec["nkv"], wv, G, addend=dnkv)
        dq_in = self._ln_bwd(dnq, rec["q_in"], rec["norm_q"], rec["stq"], G, addend=dx1)
        if addend_event is not None:
            main.wait_event(addend_event)             # (also orders the parameter-gradient folds queued by that layer before seg())
        dkv_in = self._ln_bwd(dnkv, rec["kv_in"], rec["norm_kv"], rec["stkv"], G, addend=dkv_addend)
        return dq_in, dkv_in, None

    # ------------------------------------------------------------------ backward
    def backward(self, tape: dict, dlogits: torch.Tensor, G: torch.Tensor, on_segment: Optional[Callable[[str], None]] = None):
        """G: flat fp32 gradient buffer (same layout as the parameters), accumulated into (+=)."""
        cfg, T = self.cfg, self.dtype
        training = tape["training"]
        B = tape["B"]
        if getattr(self, "_pack_ev", None) is not None:             # the transposed data-gradient operands / stem helper of begin_step
            torch.cuda.current_stream().wait_event(self._pack_ev)
            self._pack_ev = None
        self._deferred = []                       # (a backward that raised must not leak its held-back launches into this one)
        self._keep = []                           # tensors read on a side stream: released after the join at the end of backward
        self._wgq = []
        self._foldq = []
        self._foldq2 = []

        def seg(name):
            """Report a finished gradient segment.  The bucket may be all-reduced once everything enqueued so far on the CURRENT
            stream and on the weight-gradient side stream has run: hand both events to the reducer (its communication stream
            waits for them); the compute streams themselves are NOT joined, so the data-gradient chain is never held back."""
            self._flush_wgq()                     # queued Linear weight gradients belong to the segment being reported
            fq, self._foldq = self._foldq, []     # so do the parameter-gradient folds queued since the last segment: on the side stream,
            K.fold_group(fq)                      # (on this stream: measured 0.1 ms faster per step than on the weight-gradient stream)
            fq2, self._foldq2 = self._foldq2, []  # partials produced on the weight-gradient stream are folded there
            if fq2:
                self._off_path([], lambda: K.fold_group(fq2))
            if on_segment is None:
                return
            evs = []
            cur = torch.cuda.current_stream()
            e = torch.cuda.Event(); e.record(cur); evs.append(e)
            if self.wgrad_stream and self.two_streams and self.side2 is not None:
                e2 = torch.cuda.Event(); e2.record(self.side2); evs.append(e2)
            on_segment(name, evs)
        dl = dlogits.to(T).contiguous() if dlogits.dtype != T else dlogits.contiguous()

        # ---- head
        hdr = tape["head"]; c = "answer_head.classifier"
        dz = self._act_bwd(dl, None, c + ".6.bias", G, 0.0, 0, off_chain=self.chain_trim)
        dz = self._lin_act_bwd(dz, hdr["h2"], c + ".6.weight", c + ".3.bias", G, hdr["p"])
        dz = self._lin_act_bwd(dz, hdr["h1"], c + ".3.weight", c + ".0.bias", G, hdr["p"])
        dfused = self._lin_bwd(dz, hdr["fused"], c + ".0.weight", G)
        seg("answer_head")

        # ---- fusion tail: output norm, gate, pools
        pr = tape["pool"]; d, L = pr["d"], pr["L"]
        dfp = self._ln_bwd(dfused, pr["fused_pre"], "fusion.output_norm", pr["fst"], G)
        dcat = torch.empty_like(pr["cat"])
        if cfg["use_gating"]:
            dzg = torch.empty_like(pr["z"])
            call("vqa_gate_bwd", dt(T), ptr(dfp), ptr(pr["z"]), ptr(pr["cat"]), ptr(dzg), ptr(dcat), B, d)
            dzg = self._act_bwd(dzg, None, "fusion.gate.gate.0.bias", G, 0.0, 0, off_chain=self.chain_trim)
            dcat = self._lin_bwd(dzg, pr["cat"], "fusion.gate.gate.0.weight", G, addend=dcat)
        else:
            dcat[:, :d] = dfp; dcat[:, d:] = dfp
        dq = torch.empty_like(pr["q"])
        denc = torch.empty_like(pr["enc"])
        call("vqa_masked_pool_pair_bwd", dt(T), ptr(dcat), ptr(pr["maskf"]), ptr(dq), ptr(denc), B, L, d)
        # ---- cross-attention layers (reverse); image-token gradient accumulates across layers
        dimg, ev_img = None, None
        side_ok = self.hoist_cross and self.two_streams and self.side is not None
        for li in range(len(tape["clayers"]) - 1, -1, -1):
            dq, dimg, ev_img = self._attn_block_bwd(tape["clayers"][li], dq, G, dkv_addend=dimg, kv_side=side_ok and li >= 1,
                                                    addend_event=ev_img)
        if ev_img is not None:                                    # (a single layer never takes the side path; kept for safety)
            torch.cuda.current_stream().wait_event(ev_img)
        # dq is now the gradient wrt text features through the query path: it joins the pooled-text gradient on the text stream below
        # (only the text encoder's backward reads the sum; the projector / CNN chain does not wait for it)
        add_on_side = self.chain_trim and self.two_streams and self.side is not None
        if not add_on_side:
            call("vqa_add", dt(T), ptr(denc), ptr(dq), ptr(denc), denc.numel())
        # ---- projector
        pj = "fusion.image_projector.projection"; rp = tape["proj"]
        dpos = self._gslice(G, "fusion.image_projector.position_embedding")
        dpz = self._ln_bwd(dimg, rp["pz"], pj + ".1", rp["st"], G, p=rp["p"], seed=rp["seed"], dadd=dpos, period=rp["ntok"])
        dpz = self._act_bwd(dpz, None, pj + ".0.bias", G, 0.0, 0, off_chain=self.chain_trim)
        dfeat = self._lin_bwd(dpz, rp["feat"], pj + ".0.weight", G)
        seg("fusion")

        # ---- text encoder backward on the side stream, concurrently with the CNN backward below
        main = torch.cuda.current_stream()
        use_side = self.two_streams and self.side is not None
        if use_side:
            evf = torch.cuda.Event(); evf.record(main)
            self.side.wait_event(evf)
            self._keep.extend([denc, dq])                 # allocated on main, consumed on the side stream: alive until the join below
        with torch.cuda.stream(self.side if use_side else main):
            if add_on_side:
                call("vqa_add", dt(T), ptr(denc), ptr(dq), ptr(denc), denc.numel())
            fn = tape["final_norm"]
            dx = self._ln_bwd(denc, fn["x"], "text_encoder.final_norm", fn["st"], G)
            for rec in reversed(tape["tlayers"]):
                dx, _, _ = self._attn_block_bwd(rec, dx, G)
            em = tape["embed"]; emb_e = self.E["text_encoder.token_embedding.weight"]
            call("vqa_embed_bwd", dt(T), ptr(em["ids"]), ptr(dx), ptr(self._gslice(G, emb_e.name)), dx.shape[0], d, emb_e.shape[0],
                 math.sqrt(d), float(em["p"]), em["seed"])
            seg("text_encoder")
            ev_tb = torch.cuda.Event(); ev_tb.record()

        # ---- CNN stages (reverse)
        bwd_acc = training and self.fuse_bn_finalize and T == torch.bfloat16
        self._bwd_acc = bwd_acc
        if bwd_acc:
            self._acc_reset()                     # one memset for every BatchNorm-backward accumulator of this backward
        dxc = dfeat
        masked = False           # True: dxc already carries the ReLU mask of the block that consumes it (see _block_bwd)
        for s in (4, 3, 2, 1):
            srec = tape["stages"][s - 1]
            ap = f"image_encoder.stage{s}.attention"
            if "spatial" in srec:
                r = srec["spatial"]
                npix = B * r["H"] * r["W"]
                scratch = torch.empty((K.L.count("vqa_spatial_bwd_scratch", B, r["H"], r["W"]),), device=dxc.device, dtype=torch.float32)
                dxn = torch.empty_like(r["x"])
                call("vqa_spatial_bwd", dt(T), ptr(dxc), ptr(r["x"]), ptr(self.P(ap + ".spatial.conv.weight")), ptr(r["pooled2"]),
                     ptr(r["amax"]), ptr(r["amap"]), ptr(scratch), ptr(dxn), ptr(self._gslice(G, ap + ".spatial.conv.weight")),
                     B, r["H"], r["W"], r["C"])
                dxc = dxn
            if "se" in srec:
                r = srec["se"]
                scratch = torch.empty((K.L.count("vqa_se_bwd_scratch", dt(T), B, r["HW"], r["C"], r["Cr"]),), device=dxc.device, dtype=torch.float32)
                dxn = torch.empty_like(r["x"])
                # dxn is the gradient entering the last block's bn2: its BatchNorm-backward column sums leave the same pass
                lastb = srec["blocks"][-1]
                se_pre = None
                if self.fuse_se_bnred and "yd" not in lastb:
                    nblk = K.L.count("vqa_se_bwd_blocks", dt(T), B, r["HW"], r["C"])
                    if nblk > 0 and bwd_acc:
                        se_pre = (self._acc(K.L.count("vqa_bn_acc_words", 3, r["C"])), -1)      # fixed-point accumulator: the block's apply pass finalizes it
                    elif nblk > 0:
                        se_pre = (torch.empty((nblk, 3, r["C"]), device=dxc.device, dtype=torch.float32), nblk)
                call("vqa_se_bwd", dt(T), ptr(dxc), ptr(r["x"]), ptr(self.P(ap + ".se.fc1.weight")), ptr(self.P(ap + ".se.fc2.weight")),
                     ptr(r["pooled"]), ptr(r["hidden"]), ptr(r["scale"]), ptr(scratch), ptr(dxn),
                     ptr(self._gslice(G, ap + ".se.fc1.weight")), ptr(self._gslice(G, ap + ".se.fc2.weight")), B, r["HW"], r["C"], r["Cr"], 1,
                     ptr(lastb["y2"]) if se_pre else None, ptr(lastb["c2"]) if se_pre else None, ptr(se_pre[0]) if se_pre else None,
                     int(bool(se_pre) and se_pre[1] < 0))
                dxc = dxn
                masked = True                 # the SE input IS the last block's post-ReLU output: its mask was applied on the way out
            else:
                masked = False
                se_pre = None
            nb = len(srec["blocks"])
            pre = se_pre         # BatchNorm-backward sums of the next block's bn2, already reduced by the pass that produced dxc
            for bi in range(nb - 1, -1, -1):
                rec = srec["blocks"][bi]
                # the gradient handed to the previous block of the stage is masked by THAT block's ReLU in this block's epilogue,
                # which also reduces that block's bn2-backward sums (it holds the finished gradient tile anyway)
                hand = srec["blocks"][bi - 1] if (bi > 0 and "yd" not in rec) else None
                dxc, pre = self._block_bwd(rec, dxc, G, training, masked=masked, hand=hand, pre=pre)
                masked = hand is not None
            if not self._deferred:                # stage 1: its held-back weight gradients are released below, report it there
                seg(f"image_encoder.stage{s}")

        # ---- stem
        self._stem_bwd(tape, dxc, G, training, after_reduce=lambda: (self._flush_deferred_and_report(seg)))
        seg("image_encoder.stem")
        self._flush_wgq()                          # (both queues are empty here: every token-side section ends in seg())
        fq, self._foldq = self._foldq, []
        K.fold_group(fq)
        self._join_off_path()
        if use_side:
            main.wait_event(ev_tb)
        self._keep = []                           # main is now ordered after every side-stream reader: plain frees are safe

    def _flush_deferred_and_report(self, seg):
        had_deferred = bool(self._deferred)
        self._flush_deferred()
        if had_deferred:
            seg("image_encoder.stage1")

    def _stem_bwd(self, tape, dxc, G, training, after_reduce=None, fused=None):
        """Backward of conv7x7/2 -> BN -> ReLU -> MaxPool3x3/2 (models/cnn_backbone.py:349-354) given dxc = gradient of the
        pooled output [B*Hp*Wp, 64].  fused=None picks the fused weight-gradient kernel whenever the dedicated bf16 stem path
        is active; fused=False forces the two-launch path (BN/ReLU/pool backward materialised, then the generic wgrad)."""
        T = self.dtype
        B = tape["B"]
        st = tape["stem"]
        Bq, IH, IW, _, H1, W1 = st["geom"][:6]
        if fused is None:
            fused = self.stem_w2 is not None and K.stem_conv_blocks(B, IH, IW) > 0
        # BatchNorm-backward sums of the stem WITHOUT touching the 112x112 tensors: every pooling window routes its gradient to
        # exactly one position (its argmax), whose post-ReLU value is the pooled output itself, so
        #   sum g       = sum_windows dpool * [pooled > 0]
        #   sum g*xhat  = sum_windows dpool * [pooled > 0] * (pooled - beta) / gamma      (pooled = gamma*xhat + beta when > 0)
        # which is the generic BN-backward reduction with y := pooled, mean := beta, invstd := 1/gamma.
        pooled = tape["stages"][0]["blocks"][0]["x"]
        gam, bet = self.P("image_encoder.stem.1.weight"), self.P("image_encoder.stem.1.bias")
        fcoef = self._stem_fcoef                  # (0, 0, beta, 1/gamma), built in begin_step: off the tail of the step
        if fcoef is None:
            fcoef = self._make_stem_fcoef()
        rows_p = pooled.numel() // 64
        nb = K.L.count("vqa_bn_bwd_blocks", rows_p)
        slab = torch.empty((nb, 3, 64), device=dxc.device, dtype=torch.float32)
        call("vqa_bn_bwd_reduce", dt(T), ptr(dxc), ptr(pooled), ptr(pooled), ptr(fcoef), None, None, ptr(slab), rows_p, 64, 0, 0)
        bc = torch.empty((3, 64), device=dxc.device, dtype=torch.float32)
        bnp = "image_encoder.stem.1"
        call("vqa_bn_bwd_finalize", ptr(slab), nb, 64, 1, float(B * H1 * W1), ptr(self.P(bnp + ".weight")), ptr(st["coef"]),
             int(training), ptr(self._gslice(G, bnp + ".weight")), ptr(self._gslice(G, bnp + ".bias")), ptr(bc))
        if after_reduce is not None:
            after_reduce()                        # stage-1 block-0 weight gradients run beside the stem weight gradient
        if fused:
            # dy (B x 112 x 112 x 64) is never written: the weight-gradient kernel rebuilds it row by row
            dwv = self._gslice(G, "image_encoder.stem.0.weight")
            if K.PROFILE is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
            ws, wsf = K.stem_wgrad_scratch(dxc.device, B, IH, IW)
            call("vqa_stem_wgrad_fused", ptr(st["images"]), ptr(st["y"]), ptr(dxc), ptr(st["idx"]), ptr(st["coef"]), ptr(bc), ptr(dwv), B, IH, IW,
                 ptr(ws), wsf)
            if K.PROFILE is not None:
                e1.record(); K.PROFILE.append(("stem_wgrad_kernel<true>", 2.0 * B * H1 * W1 * 64 * 147, e0, e1,
                                                            B * 3 * IH * IW * 4 + B * H1 * W1 * 64 * 2 + dxc.numel() * 3))
        else:
            dy = torch.empty_like(st["y"])
            call("vqa_stem_bwd_apply", dt(T), ptr(dxc), ptr(st["idx"]), ptr(st["y"]), ptr(st["coef"]), ptr(bc), ptr(dy), B, H1, W1, 64)
            K.wgrad(dy, st["images"], LY.mat_of(G, self.E["image_encoder.stem.0.weight"]), B * H1 * W1, 64, 147, st["geom"], dtype=T,
                    loader=K.LOADER_STEM)

    def _block_bwd(self, rec, dout, G, training, masked=False, hand=None, pre=None):
        """ResidualBlock backward (reference forward: models/cnn_backbone.py:164-197).  Returns (dx, pre-reduced bn2 sums for `hand`).
        masked: `dout` was already multiplied by (out > 0) by its producer, so the block output is never re-read here.
        hand: tape record of the PREVIOUS block of the stage (this block's input is its post-ReLU output): the returned gradient
              is masked by that ReLU and the previous block's bn2-backward column sums are reduced in the same epilogue.
        pre: (slab, rows) of this block's bn2-backward sums when the producer of `dout` already reduced them."""
        outmask = hand["out"] if hand is not None else None
        T = self.dtype
        p, Cout, Cin, M = rec["p"], rec["Cout"], rec["Cin"], rec["M"]
        has_ds = "yd" in rec
        last = p == "image_encoder.stage1.blocks.0"           # its weight gradients are released with the stem backward (deferring all of stage 1 measured worse)
        gs = lambda n: self._gslice(G, n)
        out_act = None if masked else rec["out"]
        bacc = getattr(self, "_bwd_acc", False)
        pre_acc = pre is not None and pre[1] < 0             # the SE backward already filled a fixed-point accumulator for bn2
        dy2, dyd = K.bn_bwd(dout, out_act, rec["y2"], rec["c2"], self.P(p + ".bn2.weight"), Cout, training,
                            gs(p + ".bn2.weight"), gs(p + ".bn2.bias"),
                            y2=rec.get("yd"), coef2=rec.get("cd"),
                            gamma2=self.P(p + ".downsample.1.weight") if has_ds else None,
                            dgamma2=gs(p + ".downsample.1.weight") if has_ds else None,
                            dbeta2=gs(p + ".downsample.1.bias") if has_ds else None,
                            slab=pre[0] if (pre and not pre_acc) else None, nb=pre[1] if (pre and not pre_acc) else 0,
                            facc=(pre[0] if pre_acc else (self._acc(K.L.count("vqa_bn_acc_words", 3, Cout)) if bacc else None)), facc_filled=pre_acc)
        g2 = rec["g2"]; B, Ho, Wo = g2[0], g2[1], g2[2]
        c64_2 = self._c64_ok(B, Ho, Wo, Cout, Cout, 3, 1)
        if rec["a1"] is None:                                 # conv2 ran on relu(bn1(y1)) built in LDS (fuse_bn_conv): so does its weight gradient
            self._off_path([dy2], lambda: K.wgrad3x3_c64_bn(rec["y1"], rec["c1"], dy2, LY.mat_of(G, self.E[p + ".conv2.weight"]), B, Ho, Wo), defer=last)
        elif self._c64_ok(B, Ho, Wo, Cout, Cout, 3, 1, wgrad=True):
            self._off_path([dy2], lambda: K.wgrad3x3_c64(rec["a1"], dy2, LY.mat_of(G, self.E[p + ".conv2.weight"]), B, Ho, Wo), defer=last)
        elif self._c128w_ok(B, Ho, Wo, Cout, Cout, 1):
            self._off_path([dy2], lambda: K.wgrad3x3_c128(rec["a1"], dy2, LY.mat_of(G, self.E[p + ".conv2.weight"]), B, Ho, Wo))
        else:
            self._off_path([dy2], lambda: K.wgrad(dy2, rec["a1"], LY.mat_of(G, self.E[p + ".conv2.weight"]), M, Cout, 9 * Cout, g2, dtype=T))
        slab1, nb1, facc1, handed_pre = None, 0, None, None
        if self._c64p_ok(B, Ho, Wo, Cout, Cout, 3, 1) and bacc and self.fuse_bn1_reduce:
            facc1 = self._acc(K.L.count("vqa_bn_acc_words", 3, Cout))      # (as below for conv8p: bn1's backward sums leave this launch)
            da1 = K.conv3x3_c64p_bnred(dy2, self._wflip(p + ".conv2.weight"), B, Ho, Wo, rec["y1"], rec["c1"], facc1)
        elif self._c64p_ok(B, Ho, Wo, Cout, Cout, 3, 1):
            da1, _, _ = K.conv3x3_c64p(dy2, self._wflip(p + ".conv2.weight"), B, Ho, Wo)
        elif c64_2:
            da1, _, _ = K.conv3x3_c64(dy2, self._wflip(p + ".conv2.weight"), B, Ho, Wo)
        elif self.use_conv8p_bwd and self._c8p_ok(B, Ho, Wo, Cout, Cout, 3, 1, 1, bwd=True):
            if bacc and self.fuse_bn1_reduce:     # bn1's backward column sums leave the data-gradient epilogue: no bn_bwd_reduce pass over (da1, y1)
                facc1 = self._acc(K.L.count("vqa_bn_acc_words", 3, Cout))
            da1 = K.conv8p(dy2, self.Wt(p + ".conv2.weight"), B, Ho, Wo, Cout, Cout, transposed=1,
                           bnred=(rec["y1"], rec["c1"], facc1) if facc1 is not None else None)
        else:
            geom_d2 = (B, Ho, Wo, Cout, Ho, Wo, 3, 3, 1, 1)
            da1, _, _ = K.igemm(dy2, self.Wt(p + ".conv2.weight"), M, Cout, 9 * Cout, geom_d2, dtype=T, transposed=1)
        dy1, _ = K.bn_bwd(da1, None, rec["y1"], rec["c1"], self.P(p + ".bn1.weight"), Cout, training,
                          gs(p + ".bn1.weight"), gs(p + ".bn1.bias"), self_mask=True,      # a1 > 0 recomputed from y1: a1 is not read
                          slab=slab1, nb=nb1, facc=facc1 if facc1 is not None else (self._acc(K.L.count("vqa_bn_acc_words", 3, Cout)) if bacc else None),
                          facc_filled=facc1 is not None)
        g1 = rec["g1"]; H, W, stride = g1[1], g1[2], g1[8]
        c64_1 = self._c64_ok(B, H, W, Cin, Cout, 3, stride)
        if self._c64_ok(B, H, W, Cin, Cout, 3, stride, wgrad=True):
            self._off_path([dy1], lambda: K.wgrad3x3_c64(rec["x"], dy1, LY.mat_of(G, self.E[p + ".conv1.weight"]), B, H, W), defer=last)
        elif self._c128w_ok(B, H, W, Cin, Cout, stride):
            self._off_path([dy1], lambda: K.wgrad3x3_c128(rec["x"], dy1, LY.mat_of(G, self.E[p + ".conv1.weight"]), B, H, W))
        else:
            self._off_path([dy1], lambda: K.wgrad(dy1, rec["x"], LY.mat_of(G, self.E[p + ".conv1.weight"]), M, Cout, 9 * Cin, g1, dtype=T))
        Md = B * H * W
        geom_d1 = (B, Ho, Wo, Cout, H, W, 3, 3, stride, 1)
        if has_ds:
            gd = rec["gd"]
            self._off_path([dyd], lambda: K.wgrad(dyd, rec["x"], LY.mat_of(G, self.E[p + ".downsample.0.weight"]), M, Cout, Cin, gd, dtype=T))
            if stride == 2 and H % 2 == 0 and W % 2 == 0:
                # conv1 (3x3/2) and shortcut (1x1/2) data gradients in ONE launch over parity classes: no redundant taps
                wt = self._packT(p + ".dgrad2", [(self.E[p + ".conv1.weight"].offset, Cout, 9, Cin, 0, False),
                                                 (self.E[p + ".downsample.0.weight"].offset, Cout, 1, Cin, 9 * Cout, False)], Cin, 10 * Cout)
                dx = K.dgrad_s2(dy1, dyd, wt, B, Ho, Wo, Cout, H, W, Cin, 3, 1, dtype=T)
            else:
                geom_dd = (B, Ho, Wo, Cout, H, W, 1, 1, stride, 0)
                dxd, _, _ = K.igemm(dyd, self.Wt(p + ".downsample.0.weight"), Md, Cin, Cout, geom_dd, dtype=T, transposed=1)
                dx, _, _ = K.igemm(dy1, self.Wt(p + ".conv1.weight"), Md, Cin, 9 * Cout, geom_d1, dtype=T, transposed=1, addend=dxd)
        elif c64_1 and not masked and outmask is None:
            dx, _, _ = K.conv3x3_c64(dy1, self._wflip(p + ".conv1.weight"), B, H, W, addend=dout, addmask=rec["out"])
        elif self.use_c64p_epi and self._c64p_ok(B, H, W, Cout, Cin, 3, stride):
            # stage 1: the patch kernel with the identity-path gradient and the ReLU masks in its per-tile epilogue (was the 128 x 64 igemm tile)
            dx = K.conv3x3_c64p_epi(dy1, self._wflip(p + ".conv1.weight"), B, H, W, addend=dout, addmask=out_act, outmask=outmask)
        elif self.use_conv8p_bwd and stride == 1 and self._c8p_ok(B, H, W, Cout, Cin, 3, stride, 1, bwd=True):
            bnred = None
            if hand is not None and bacc and self.fuse_hand_reduce:
                # dx is the gradient entering the previous block's bn2 (and its shortcut BatchNorm), already masked by that block's
                # ReLU (outmask): their backward column sums leave this epilogue and that block skips its bn_bwd_reduce pass
                hfacc = self._acc(K.L.count("vqa_bn_acc_words", 3, Cin))
                bnred = (hand["y2"], hand["c2"], hfacc, False, hand.get("yd"), hand.get("cd"))
                handed_pre = (hfacc, -1)
            dx = K.conv8p(dy1, self.Wt(p + ".conv1.weight"), B, H, W, Cout, Cin, transposed=1, addend=dout, addmask=out_act, outmask=outmask,
                          bnred=bnred)
        else:
            dx, _, _ = K.igemm(dy1, self.Wt(p + ".conv1.weight"), Md, Cin, 9 * Cout, geom_d1, dtype=T, transposed=1,
                               addend=dout, addmask=out_act, outmask=outmask)
        if self.capture is not None:
            self.capture[p] = dict(dout=dout, masked=masked, handed=hand is not None, dy2=dy2, dyd=dyd, da1=da1, dy1=dy1, dx=dx)
        return dx, handed_pre
