"""How a window-loader conv launch's time depends on its tile count (measurement tool, not part of the product): the stage-3 conv
(256->256, 14x14, 128x128 tiles, 2 column tiles) over a sweep of batch sizes, so the tile count crosses multiples of the 512
workgroup slots (profiles/r03_tile_sweep_s{2,3}.txt; the second column there is the rejected two-launch tail split)."""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=40)
ap.add_argument("--C", type=int, default=256)
ap.add_argument("--H", type=int, default=14)
ap.add_argument("--batches", default="84,160,167,168,175,200,250,300,334,335,345,400,450,500,501,502,512,520,600,668,669")
args = ap.parse_args()
pkg = importlib.import_module("visual-question-answering-vqa-system_amd")
K = pkg.kernels
T, dev, C, H = torch.bfloat16, "cuda", args.C, args.H


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


w = torch.randn(C, 9 * C, device=dev) * 0.05
wp = K.pack_rows(w, T)
print(f"conv3x3 {C}->{C} {H}x{H}: batch, tiles (128x128), rounds of 512, us, TF/s")
for B in [int(b) for b in args.batches.split(",")]:
    M, Kw = B * H * H, 9 * C
    x = torch.randn(M, C, device=dev).to(T)
    geom = (B, H, H, C, H, H, 3, 3, 1, 1)
    tiles = (M + 127) // 128 * ((C + 127) // 128)
    fl = 2.0 * M * C * Kw
    t = timeit(lambda: K.igemm(x, wp, M, C, Kw, geom, dtype=T), args.iters)
    print(f"B={B:4d} tiles={tiles:5d} rounds={tiles/512:5.2f} var={K.igemm_variant(T, K.LOADER_NHWC, M, C, Kw, geom)}  {t*1e6:7.1f} us {fl/t/1e12:6.1f} TF/s", flush=True)
