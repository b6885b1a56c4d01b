"""Stage-1 patch kernels with and without the in-LDS BatchNorm + ReLU prologue (round 4), B=512: microseconds per launch.
    python tools/bench_c64_bn.py [--batch 512]"""
import argparse, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("visual-question-answering-vqa-system_amd")
K, L = pkg.kernels, pkg._lib
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=512); ap.add_argument("--iters", type=int, default=20)
args = ap.parse_args()
B, H, W, dev, bf = args.batch, 56, 56, "cuda", torch.bfloat16


def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


n = B * H * W
y1 = torch.randn(n, 64, device=dev).to(bf)
w = (torch.randn(64, 576, device=dev) * 0.05).to(bf)
dy = torch.randn(n, 64, device=dev).to(bf)
gam, bet = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.3
bnp = (gam, bet, torch.zeros(64, device=dev), torch.ones(64, device=dev), torch.zeros((), device=dev, dtype=torch.int64))
words = L.count("vqa_bn_acc_words", 2, 64)
acc = torch.zeros(words, device=dev, dtype=torch.int64)
K.conv3x3_c64p(torch.randn(n, 64, device=dev).to(bf), w, B, H, W, want_stats=True, stats_acc=acc)        # plausible statistics
a1, coef, _, _, _ = K.bn_apply_acc(y1, acc, bnp, 64, True, B, H * W, n)
dw = torch.zeros(64, 576, device=dev)
t_apply = timeit(lambda: K.bn_apply_acc(y1, acc, bnp, 64, True, B, H * W, n), args.iters)
t_conv = timeit(lambda: K.conv3x3_c64p(a1, w, B, H, W, want_stats=True, stats_acc=torch.zeros(words, device=dev, dtype=torch.int64)), args.iters)
t_convbn = timeit(lambda: K.conv3x3_c64p_bn(y1, acc, bnp, w, B, H, W, n, want_stats=True, stats_acc=torch.zeros(words, device=dev, dtype=torch.int64)), args.iters)
t_wg = timeit(lambda: K.wgrad3x3_c64(a1, dy, dw, B, H, W), args.iters)
t_wgbn = timeit(lambda: K.wgrad3x3_c64_bn(y1, coef, dy, dw, B, H, W), args.iters)
print(f"bn_apply_acc {t_apply:7.1f} us | conv3x3_c64p {t_conv:7.1f} -> with prologue {t_convbn:7.1f} (+{t_convbn - t_conv:5.1f}) | "
      f"wgrad3x3_c64 {t_wg:7.1f} -> with prologue {t_wgbn:7.1f} (+{t_wgbn - t_wg:5.1f}) | net per block {t_convbn - t_conv + t_wgbn - t_wg - t_apply:+7.1f} us")
