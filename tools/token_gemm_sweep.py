"""Tile choice of the token-side Linear launches (measurement tool): every (M, N, K) of the B = 512 step's dense GEMMs with each
igemm tile (128x128, 128x64, 64x64) forced through VQA_IGEMM_TILE in the ablation build, against what igemm_tile() picks.
    python tools/token_gemm_sweep.py [--iters 30]"""
import argparse
import importlib
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import tools.build_ablation as A
A.build()                                   # (here, on the CPU box, or on the GPU box: hipcc is on both)
A.use()
pkg = importlib.import_module("visual-question-answering-vqa-system_amd")
K = pkg.kernels

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=30)
args = ap.parse_args()
T, dev = torch.bfloat16, "cuda"
SHAPES = [(10240, 256, 256, "Q / W_o / fc2-dgrad-like"), (10240, 768, 256, "QKV"), (10240, 1024, 256, "fc1"), (10240, 256, 1024, "fc2"),
          (25088, 256, 512, "image projector"), (25088, 512, 256, "cross K|V"), (25088, 256, 256, "cross K|V dgrad"), (512, 512, 256, "head 0"),
          (512, 256, 512, "head 3 / gate"), (512, 1000, 256, "head 6")]


def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for M, N, Kd, what in SHAPES:
    x = torch.randn(M, Kd, device=dev).to(T)
    w = (torch.randn(N, Kd, device=dev) * 0.05).to(T)
    geom = K.linear_geom(M, Kd)
    fl = 2.0 * M * N * Kd
    os.environ["VQA_IGEMM_ST"] = "2"
    ref = K.igemm(x, w, M, N, Kd, geom, dtype=T)[0].clone()
    for ring in ("2", "3", "4"):
        os.environ["VQA_IGEMM_ST"] = ring
        for force in ("128128", "128064", "64064"):                # the ring only changes WHEN a tile lands: same bits
            os.environ["VQA_IGEMM_TILE"] = force
            assert torch.equal(K.igemm(x, w, M, N, Kd, geom, dtype=T)[0], ref), (ring, force)
        row = []
        for force in ("0", "128128", "128064", "64064"):
            os.environ["VQA_IGEMM_TILE"] = force
            t = timeit(lambda: K.igemm(x, w, M, N, Kd, geom, dtype=T), args.iters)
            row.append(t)
        os.environ["VQA_IGEMM_TILE"] = "0"
        print(f"M={M:6d} N={N:5d} K={Kd:5d} ring {ring}  picked {row[0]:6.1f} us | 128x128 {row[1]:6.1f} | 128x64 {row[2]:6.1f} | 64x64 {row[3]:6.1f}   ({fl/row[0]/1e6:6.1f} TF/s)  {what}", flush=True)
    os.environ["VQA_IGEMM_ST"] = "2"
