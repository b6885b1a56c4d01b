"""Per-call timing of every generic weight-gradient launch of one B=512 train step, by shape (not part of the product)."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "visual-question-answering-vqa-system_amd"
pkg = importlib.import_module(PKG)
K = pkg.kernels
import bench  # noqa: E402  (synthetic batch)

dev = torch.device("cuda:0")
B = int(os.environ.get("B", "512"))
model = pkg.load_dropin().VQAModel(compute_dtype="bf16", seed=1234).to(dev).train()
tr = pkg.trainer.HipTrainer(model)
tr.engine.two_streams = False
images, ids, mask, answers = bench.synth_batch(B, dev, 1234)
for _ in range(3):
    tr.step(images, ids, mask, answers)
torch.cuda.synchronize()
rec = []
orig = K.wgrad


def wrapped(dy, x, dw, M, N, Kw, geom, *, dtype, loader=K.LOADER_NHWC):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    orig(dy, x, dw, M, N, Kw, geom, dtype=dtype, loader=loader)
    e1.record()
    Bq, H, W, C, Ho, Wo, R, S, stride, pad = geom
    rec.append((M, N, Kw, R, stride, loader, K.wgrad_plan(dtype, loader, M, N, Kw, Bq, H, W, C, R, S), e0, e1))


K.wgrad = wrapped
tr.step(images, ids, mask, answers)
torch.cuda.synchronize()
agg = {}
for M, N, Kw, R, stride, loader, plan, e0, e1 in rec:
    a = agg.setdefault((M, N, Kw, R, stride, loader, plan), [0, 0.0])
    a[0] += 1; a[1] += e0.elapsed_time(e1) * 1e3
tot = 0.0
print(f"{'M':>8} {'N':>5} {'Kw':>5} R s ld  plan(kind,tn,tk,nsplit,ws)             n   us/call   TF/s   total us")
for (M, N, Kw, R, stride, loader, plan), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tot += us
    print(f"{M:8d} {N:5d} {Kw:5d} {R} {stride} {loader}  {str(plan):38s} {n:3d} {us/n:9.1f} {2.0*M*N*Kw/(us/n*1e-6)/1e12:6.1f} {us:9.1f}")
print(f"total {tot:.1f} us over {len(rec)} launches (each = kernel + fixed-order reduce, events on the launch stream)")
