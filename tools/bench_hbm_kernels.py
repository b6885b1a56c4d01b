"""Micro-benchmark of the HBM-bound CNN passes (BatchNorm apply / backward, SE, spatial attention) on the stage shapes of the
B=512 bf16 train step: microseconds and achieved algorithmic TB/s per launch, back to back on a tensor set larger than the
Infinity Cache when --rotate is given (default: the same operands every iteration, i.e. the in-step situation of stage 3/4).
Measurement tool, not part of the product.
    python tools/bench_hbm_kernels.py [--batch 512] [--iters 20] [--only bn|se|spatial] [--rotate 4]"""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("visual-question-answering-vqa-system_amd")
K, L = pkg.kernels, pkg._lib

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--only", default="")
ap.add_argument("--rotate", type=int, default=1, help="operand sets cycled through (4 sets of stage 1 = 1.6 GB: nothing stays in the 256 MB cache)")
args = ap.parse_args()
B, T, dev = args.batch, torch.bfloat16, "cuda"
STAGES = [(1, 64, 56), (2, 128, 28), (3, 256, 14), (4, 512, 7)]


def timeit(fns, iters):
    for f in fns:
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fns[i % len(fns)]()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def report(name, t, nbytes):
    print(f"{name:46s} {t*1e6:8.1f} us  {nbytes/t/1e12:5.2f} TB/s", flush=True)


tot = {}
for s, C, H in STAGES:
    HW, rows = H * H, B * H * H
    n = rows * C
    R = args.rotate
    rnd = lambda: [torch.randn(rows, C, device=dev).to(T) for _ in range(R)]
    y, x, d = rnd(), [torch.relu(t) for t in rnd()], rnd()
    gam, bet = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
    rm, rv, nbt = torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.zeros((), device=dev, dtype=torch.int64)
    bnp = (gam, bet, rm, rv, nbt)
    coef = torch.stack([gam, bet, torch.zeros(C, device=dev), torch.ones(C, device=dev)]).contiguous()
    if not args.only or "bn" in args.only:
        # forward apply: statistics in accumulators (filled once by the slab-free reference: zeros give mean 0 / var 0 -> fine for timing)
        accs = [torch.zeros(L.count("vqa_bn_acc_words", 2, C), device=dev, dtype=torch.int64) for _ in range(R)]
        t = timeit([lambda i=i: K.bn_apply_acc(y[i], accs[i], bnp, C, True, B, HW, rows) for i in range(R)], args.iters)
        report(f"s{s} bn_apply_acc            (r y, w a)", t, 2 * n * 2); tot["bn"] = tot.get("bn", 0) + 2 * t
        t = timeit([lambda i=i: K.bn_apply_acc(y[i], accs[i], bnp, C, True, B, HW, rows, res=x[i]) for i in range(R)], args.iters)
        report(f"s{s} bn_apply_acc + res      (r y x, w out)", t, 3 * n * 2); tot["bn"] += 2 * t
        dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)

        def bwd(i, self_mask):
            facc = torch.zeros(L.count("vqa_bn_acc_words", 3, C), device=dev, dtype=torch.int64)
            return K.bn_bwd(d[i], None, y[i], coef, gam, C, True, dg, db, self_mask=self_mask, facc=facc)
        t = timeit([lambda i=i: bwd(i, True) for i in range(R)], args.iters)
        report(f"s{s} bn_bwd reduce+apply     (r d y | r d y, w dy)", t, 5 * n * 2); tot["bn"] += 4 * t
    if not args.only or "se" in args.only:
        Cr = max(C // 16, 1)
        w1, w2 = torch.randn(Cr, C, device=dev) * 0.2, torch.randn(C, Cr, device=dev) * 0.2
        pooled, hidden, scale = torch.rand(B, C, device=dev), torch.rand(B, Cr, device=dev), torch.rand(B, C, device=dev)
        out = torch.empty_like(x[0])

        def se_fwd(i):
            call_args = (L.dt(T), x[i].data_ptr(), w1.data_ptr(), w2.data_ptr(), pooled.data_ptr(), hidden.data_ptr(), scale.data_ptr(),
                         out.data_ptr(), B, HW, C, Cr, None, 0)
            L.call("vqa_se_fwd", *call_args)
        t = timeit([lambda i=i: se_fwd(i) for i in range(R)], args.iters)
        report(f"s{s} se_fwd (pool + scale)    (r x | r x, w out)", t, 3 * n * 2); tot["se"] = tot.get("se", 0) + t
        scratch = torch.empty(L.count("vqa_se_bwd_scratch", L.dt(T), B, HW, C, Cr), device=dev)
        dw1, dw2 = torch.zeros_like(w1), torch.zeros_like(w2)
        for mode in (0,):
            def se_bwd(i, bn):
                facc = torch.zeros(L.count("vqa_bn_acc_words", 3, C), device=dev, dtype=torch.int64) if bn else None
                L.call("vqa_se_bwd", L.dt(T), d[i].data_ptr(), x[i].data_ptr(), w1.data_ptr(), w2.data_ptr(), pooled.data_ptr(), hidden.data_ptr(),
                       scale.data_ptr(), scratch.data_ptr(), out.data_ptr(), dw1.data_ptr(), dw2.data_ptr(), B, HW, C, Cr, 1,
                       y[i].data_ptr() if bn else None, coef.data_ptr() if bn else None, facc.data_ptr() if bn else None, mode | int(bn))
            nm = "per-sample"
            t = timeit([lambda i=i: se_bwd(i, False) for i in range(R)], args.iters)
            report(f"s{s} se_bwd {nm:10s}        (r d x | r d x, w dx)", t, 5 * n * 2)
            t = timeit([lambda i=i: se_bwd(i, True) for i in range(R)], args.iters)
            report(f"s{s} se_bwd {nm:10s} + bnred (.. + r y2)", t, 6 * n * 2)
            tot["se"] += t
    if (not args.only or "spatial" in args.only) and s >= 3:
        w = torch.randn(98, device=dev) * 0.1
        pooled2 = torch.empty(rows, 2, device=dev); amax = torch.empty(rows, device=dev, dtype=torch.int32); amap = torch.empty(rows, device=dev)
        out = torch.empty_like(x[0])
        t = timeit([lambda i=i: L.call("vqa_spatial_fwd", L.dt(T), x[i].data_ptr(), w.data_ptr(), pooled2.data_ptr(), amax.data_ptr(), amap.data_ptr(),
                                       out.data_ptr(), B, H, H, C) for i in range(R)], args.iters)
        report(f"s{s} spatial_fwd              (r x | r x, w out)", t, 3 * n * 2); tot["sp"] = tot.get("sp", 0) + t
        scratch = torch.empty(L.count("vqa_spatial_bwd_scratch", B, H, H), device=dev)
        dwv = torch.zeros(98, device=dev)
        t = timeit([lambda i=i: L.call("vqa_spatial_bwd", L.dt(T), d[i].data_ptr(), x[i].data_ptr(), w.data_ptr(), pooled2.data_ptr(), amax.data_ptr(),
                                       amap.data_ptr(), scratch.data_ptr(), out.data_ptr(), dwv.data_ptr(), B, H, H, C) for i in range(R)], args.iters)
        report(f"s{s} spatial_bwd              (r d x | r d, w dx)", t, 4 * n * 2); tot["sp"] += t
    del y, x, d
    torch.cuda.empty_cache()
print({k: f"{v*1e3:.3f} ms per step (2 blocks per stage for bn)" for k, v in tot.items()})
