"""Micro-benchmark of the GEMM-class kernels on the layer shapes of the B=512 train step (not part of the product)."""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("visual-question-answering-vqa-system_amd")
K = pkg.kernels

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--only", default="")
args = ap.parse_args()
B, T, dev = args.batch, torch.bfloat16, "cuda"

# (name, Cin, Cout, H, R, stride, pad)
CONVS = [("s1 3x3 64->64 56", 64, 64, 56, 3, 1, 1), ("s2a 3x3/2 64->128", 64, 128, 56, 3, 2, 1), ("s2 3x3 128->128 28", 128, 128, 28, 3, 1, 1),
         ("s3 3x3 256->256 14", 256, 256, 14, 3, 1, 1), ("s4 3x3 512->512 7", 512, 512, 7, 3, 1, 1), ("s3a 3x3/2 128->256", 128, 256, 28, 3, 2, 1)]


def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


for name, Cin, Cout, H, R, stride, pad in CONVS:
    if args.only and args.only not in name:
        continue
    Ho = (H + 2 * pad - R) // stride + 1
    x = torch.randn(B * H * H, Cin, device=dev).to(T)
    w = (torch.randn(Cout, R * R * Cin, device=dev) * 0.05)
    dy = torch.randn(B * Ho * Ho, Cout, device=dev).to(T)
    wp = K.pack_rows(w, T)
    wt = K.pack_transpose(w.view(Cout, R * R, Cin), T)
    M, Kw = B * Ho * Ho, R * R * Cin
    geom = (B, H, H, Cin, Ho, Ho, R, R, stride, pad)
    geom_d = (B, Ho, Ho, Cout, H, H, R, R, stride, pad)
    dw = torch.zeros(Cout, Kw, device=dev)
    fl = 2.0 * M * Cout * Kw
    t = timeit(lambda: K.igemm(x, wp, M, Cout, Kw, geom, dtype=T, want_stats=True), args.iters)
    print(f"{name:22s} fwd   {t*1e6:8.1f} us {fl/t/1e12:7.1f} TF/s   io {(x.numel()+M*Cout)*2/t/1e12:5.2f} TB/s")
    t = timeit(lambda: K.igemm(x, wp, M, Cout, Kw, geom, dtype=T, want_stats=False), args.iters)
    print(f"{name:22s} fwd-ns{t*1e6:8.1f} us {fl/t/1e12:7.1f} TF/s   (no BN statistics)")
    t = timeit(lambda: K.igemm(dy, wt, B * H * H, Cin, R * R * Cout, geom_d, dtype=T, transposed=1), args.iters)
    print(f"{name:22s} dgrad {t*1e6:8.1f} us {fl/t/1e12:7.1f} TF/s")
    t = timeit(lambda: K.wgrad(dy, x, dw, M, Cout, Kw, geom, dtype=T), args.iters)
    print(f"{name:22s} wgrad {t*1e6:8.1f} us {fl/t/1e12:7.1f} TF/s")

if not args.only or "c8p" in args.only:
    for name, C, H in (("s2 128->128 28", 128, 28), ("s3 256->256 14", 256, 14), ("s4 512->512 7", 512, 7)):
        x = torch.randn(B * H * H, C, device=dev).to(T)
        w = (torch.randn(C, 9 * C, device=dev) * 0.03).to(T)
        fl = 2.0 * B * H * H * C * 9 * C
        words = K.L.count("vqa_bn_acc_words", 2, C)
        t = timeit(lambda: K.conv8p(x, w, B, H, H, C, C, stats_acc=torch.zeros(words, device=dev, dtype=torch.int64)), args.iters)
        print(f"{name:22s} conv8p fwd   {t*1e6:8.1f} us {fl/t/1e12:7.1f} TF/s")
        t = timeit(lambda: K.conv8p(x, w, B, H, H, C, C, transposed=1), args.iters)
        print(f"{name:22s} conv8p dgrad {t*1e6:8.1f} us {fl/t/1e12:7.1f} TF/s")
        yb = torch.randn(B * H * H, C, device=dev).to(T)
        cf = torch.stack([torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.3, torch.zeros(C, device=dev), torch.ones(C, device=dev)]).contiguous()
        w3 = K.L.count("vqa_bn_acc_words", 3, C)
        t = timeit(lambda: K.conv8p(x, w, B, H, H, C, C, transposed=1, bnred=(yb, cf, torch.zeros(w3, device=dev, dtype=torch.int64))), args.iters)
        print(f"{name:22s} conv8p dgrad + bn1 backward sums {t*1e6:8.1f} us {fl/t/1e12:7.1f} TF/s")
        t = timeit(lambda: K.conv8p(x, w, B, H, H, C, C, transposed=1, addend=yb, outmask=yb), args.iters)
        print(f"{name:22s} conv8p dgrad + addend + outmask  {t*1e6:8.1f} us {fl/t/1e12:7.1f} TF/s")

if not args.only or "c64" in args.only:
    H = 56
    x = torch.randn(B * H * H, 64, device=dev).to(T)
    w = torch.randn(64, 576, device=dev) * 0.05
    dy = torch.randn(B * H * H, 64, device=dev).to(T)
    wp = K.pack_rows(w, T)
    dw = torch.zeros(64, 576, device=dev)
    fl = 2.0 * B * H * H * 64 * 576
    t = timeit(lambda: K.conv3x3_c64(x, wp, B, H, H, want_stats=True), args.iters)
    print(f"c64 patch conv fwd     {t*1e6:8.1f} us {fl/t/1e12:7.1f} TF/s   io {(2*x.numel())*2/t/1e12:5.2f} TB/s")
    wfl = K.pack_transpose(w.view(64, 9, 64), T, flip=True)
    t = timeit(lambda: K.conv3x3_c64p(x, wp, B, H, H, want_stats=True), args.iters)
    print(f"c64 DMA patch conv fwd {t*1e6:8.1f} us {fl/t/1e12:7.1f} TF/s   io {(2*x.numel())*2/t/1e12:5.2f} TB/s")
    t = timeit(lambda: K.conv3x3_c64p(dy, wfl, B, H, H), args.iters)
    print(f"c64 DMA patch dgrad    {t*1e6:8.1f} us {fl/t/1e12:7.1f} TF/s")
    cf = torch.stack([torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.3, torch.zeros(64, device=dev), torch.ones(64, device=dev)]).contiguous()
    w3 = K.L.count("vqa_bn_acc_words", 3, 64)
    t = timeit(lambda: K.conv3x3_c64p_bnred(dy, wfl, B, H, H, x, cf, torch.zeros(w3, device=dev, dtype=torch.int64)), args.iters)
    print(f"c64 DMA patch dgrad + bn1 backward sums  {t*1e6:8.1f} us {fl/t/1e12:7.1f} TF/s")
    t = timeit(lambda: K.conv3x3_c64p_epi(dy, wfl, B, H, H, addend=x), args.iters)
    print(f"c64 DMA patch dgrad + addend             {t*1e6:8.1f} us {fl/t/1e12:7.1f} TF/s")
    t = timeit(lambda: K.conv3x3_c64p_epi(dy, wfl, B, H, H, addend=x, outmask=x), args.iters)
    print(f"c64 DMA patch dgrad + addend + outmask   {t*1e6:8.1f} us {fl/t/1e12:7.1f} TF/s")
    t = timeit(lambda: K.wgrad3x3_c64(x, dy, dw, B, H, H), args.iters)
    print(f"c64 patch wgrad        {t*1e6:8.1f} us {fl/t/1e12:7.1f} TF/s")

if (not args.only or "c128" in args.only) and K.c128_wgrad_blocks(B, 28, 28) > 0:
    x = torch.randn(B * 28 * 28, 128, device=dev).to(T)
    dy = torch.randn(B * 28 * 28, 128, device=dev).to(T)
    dw = torch.zeros(128, 1152, device=dev)
    fl = 2.0 * B * 28 * 28 * 128 * 1152
    t = timeit(lambda: K.wgrad3x3_c128(x, dy, dw, B, 28, 28), args.iters)
    print(f"c128 resident wgrad    {t*1e6:8.1f} us {fl/t/1e12:7.1f} TF/s   (kernel + fixed-order reduce)")
    t = timeit(lambda: K.wgrad(dy, x, dw, B * 784, 128, 1152, (B, 28, 28, 128, 28, 28, 3, 3, 1, 1), dtype=T), args.iters)
    print(f"c128 split-K wgrad     {t*1e6:8.1f} us {fl/t/1e12:7.1f} TF/s   (generic kernel + reduce)")

if not args.only or "stem" in args.only:
    H = 224
    img = torch.randn(B, 3, H, H, device=dev)
    dy = torch.randn(B * 112 * 112, 64, device=dev).to(T)
    dw = torch.zeros(64, 147, device=dev)
    fl = 2.0 * B * 112 * 112 * 64 * 147
    t = timeit(lambda: K.stem_wgrad(img, dy, dw, B, H, H), args.iters)
    print(f"stem wgrad (unfused)   {t*1e6:8.1f} us {fl/t/1e12:7.1f} TF/s   io {(img.numel()*4+dy.numel()*2)/t/1e12:5.2f} TB/s")

if not args.only or "tok" in args.only:
    # token-side weight gradients of a B=512 step: (rows, K in, N out)
    for M, Kin, N in ((10240, 256, 768), (10240, 256, 256), (10240, 256, 1024), (10240, 1024, 256), (25088, 512, 256), (25088, 256, 512),
                      (25088, 256, 256), (512, 512, 256), (512, 256, 512), (512, 256, 1000)):
        x = torch.randn(M, Kin, device=dev).to(T)
        dy = torch.randn(M, N, device=dev).to(T)
        dw = torch.zeros(N, Kin, device=dev)
        t = timeit(lambda: K.wgrad(dy, x, dw, M, N, Kin, K.linear_geom(M, Kin), dtype=T), args.iters)
        print(f"tok wgrad M={M:6d} K={Kin:5d} N={N:5d}  {t*1e6:8.1f} us {2.0*M*N*Kin/t/1e12:7.1f} TF/s  plan {K.wgrad_plan(T, 0, M, N, Kin, M, 1, 1, Kin, 1, 1)}")
