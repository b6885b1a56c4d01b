"""Does an HBM-bound BatchNorm pass hide under an MFMA-bound conv of the OTHER half of the batch?  (measurement tool)
For each residual stage shape of the B=512 bf16 step: time  conv(full)  and  bn_apply(full + residual)  alone, their sum, and the
skewed micro-batch schedule of DESIGN (round 4)
    main : conv(A)            apply(A)   conv'(A)              apply'(A) ...
    side :          conv(B)              apply(B) || conv'(A)  conv'(B)  ...
i.e. per layer  conv(A) | conv(B) | apply(A) | [apply(B) || conv(A) of the next layer], here as a steady-state loop over one layer.
    python tools/overlap_probe.py [--batch 512] [--iters 20]"""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("visual-question-answering-vqa-system_amd")
K, L, E = pkg.kernels, pkg._lib, pkg.engine

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--iters", type=int, default=20)
args = ap.parse_args()
B, T, dev = args.batch, torch.bfloat16, torch.device("cuda", 0)
torch.cuda.set_device(dev)
side = E.pick_concurrent_streams(dev, 1)[0]
main = torch.cuda.current_stream()


def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for s, C, H in [(1, 64, 56), (2, 128, 28), (3, 256, 14), (4, 512, 7)]:
    HW, rows = H * H, B * H * H
    x = torch.relu(torch.randn(rows, C, device=dev)).to(T)
    res = torch.randn(rows, C, device=dev).to(T)
    w = (torch.randn(C, 9 * C, device=dev) * 0.03).to(T)
    y = torch.empty(rows, C, device=dev, dtype=T)
    out = torch.empty(rows, C, device=dev, dtype=T)
    gam, bet = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
    coef = torch.stack([gam, bet, torch.zeros(C, device=dev), torch.ones(C, device=dev)]).contiguous()
    acc = torch.zeros(L.count("vqa_bn_acc_words", 2, C), device=dev, dtype=torch.int64)
    hb, hr = B // 2, rows // 2

    def conv(lo, nb):
        r0, r1 = lo * HW, (lo + nb) * HW
        geom = (nb, H, H, C, H, H, 3, 3, 1, 1)
        if C == 64:
            L.call("vqa_conv3x3_c64p", x[r0:r1].data_ptr(), w.data_ptr(), y[r0:r1].data_ptr(), acc.data_ptr(), nb, H, H, 1)
        else:
            K.igemm(x[r0:r1], w, nb * HW, C, 9 * C, geom, dtype=T, want_stats=True, stats_acc=acc, out=y[r0:r1])

    def apply(lo, nb):
        r0, r1 = lo * HW, (lo + nb) * HW
        L.call("vqa_bn_apply", L.dt(T), y[r0:r1].data_ptr(), coef.data_ptr(), res[r0:r1].data_ptr(), None, out[r0:r1].data_ptr(), (r1 - r0) * C, C, 1)

    t_conv = timeit(lambda: conv(0, B), args.iters)
    t_app = timeit(lambda: apply(0, B), args.iters)
    t_seq = timeit(lambda: (conv(0, B), apply(0, B)), args.iters)
    t_half = timeit(lambda: (conv(0, hb), conv(hb, hb), apply(0, hb), apply(hb, hb)), args.iters)

    def skewed():
        # steady state of one layer: [apply(B) of the previous iteration || conv(A)] | conv(B) | apply(A)
        ev_a = torch.cuda.Event(); ev_a.record(main)
        with torch.cuda.stream(side):
            side.wait_event(ev_a)
            apply(hb, hb)                    # apply(B) of the previous layer ...
            ev_b = torch.cuda.Event(); ev_b.record(side)
        conv(0, hb)                          # ... under conv(A) of this one
        main.wait_event(ev_b)
        conv(hb, hb)
        apply(0, hb)
    t_skew = timeit(skewed, args.iters)
    print(f"stage {s}: conv {t_conv:7.1f}  apply {t_app:6.1f}  back-to-back {t_seq:7.1f}  halves back-to-back {t_half:7.1f}  skewed 2 streams {t_skew:7.1f} us"
          f"   (hidden {t_half - t_skew:6.1f} of {t_app / 2:5.1f})", flush=True)
