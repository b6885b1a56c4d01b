"""Why a window-loader conv launch takes longer inside the step than in a tight loop (measurement tool, not part of the product):
the stage-3 conv (B=512, 256->256, 14x14) timed with HIP events around EACH launch, (a) back to back, (b) after the GPU idled,
(c) after a kernel that rewrote its input (what BatchNorm-apply does in the step), (d) after 512 MB of unrelated traffic
(cold L2 / Infinity Cache), (e) after an HBM-bound kernel of comparable length."""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("visual-question-answering-vqa-system_amd")
K = pkg.kernels
T, dev = torch.bfloat16, "cuda"
B, C, H = 512, 256, 14
M, Kw = B * H * H, 9 * C
geom = (B, H, H, C, H, H, 3, 3, 1, 1)
x = torch.randn(M, C, device=dev).to(T)
x2 = torch.randn(M, C, device=dev).to(T)
wp = K.pack_rows(torch.randn(C, Kw, device=dev) * 0.05, T)
out = torch.empty(M, C, device=dev, dtype=T)
big = torch.empty(256 << 20, device=dev, dtype=torch.uint8)
big2 = torch.empty(256 << 20, device=dev, dtype=torch.uint8)
fl = 2.0 * M * C * Kw


def conv():
    K.igemm(x, wp, M, C, Kw, geom, dtype=T, out=out)


def timed(pre, n=30):
    ts = []
    for _ in range(n):
        pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); conv(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0], ts[-1]


for _ in range(20):
    conv()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    conv()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) * 1e3 / 200
print(f"back to back, 200 launches            {t:7.1f} us/launch  {fl/t/1e6:6.1f} TF/s")
cases = [("single launch, GPU just synchronised", lambda: None),
         ("after 20 ms of idle", lambda: time.sleep(0.02)),
         ("after a kernel that rewrote its input", lambda: x.copy_(x2)),
         ("after 512 MB of unrelated traffic", lambda: big2.copy_(big)),
         ("after 10 back-to-back convs", lambda: [conv() for _ in range(10)]),
         ("after 10 input rewrites (HBM-bound)", lambda: [x.copy_(x2) for _ in range(10)])]
for name, pre in cases:
    med, lo, hi = timed(pre)
    print(f"{name:38s}{med:7.1f} us (min {lo:.1f}, max {hi:.1f})  {fl/med/1e6:6.1f} TF/s", flush=True)


# ---- sustained activity, no host sync between launches: what the conv sees inside the step
def sustained(label, pre, conv_i, n=40):
    evs = []
    for i in range(n):
        pre(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); conv_i(i); e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs[8:])
    med = ts[len(ts) // 2]
    print(f"{label:62s}{med:7.1f} us (min {ts[0]:.1f}, max {ts[-1]:.1f})  {fl/med/1e6:6.1f} TF/s", flush=True)


NB = 10                                                    # 10 x (51 + 51 MB) = 1 GB: more than the 256 MB Infinity Cache
xs = [torch.randn(M, C, device=dev).to(T) for _ in range(NB)]
outs = [torch.empty(M, C, device=dev, dtype=T) for _ in range(NB)]
ws = [K.pack_rows(torch.randn(C, Kw, device=dev) * 0.05, T) for _ in range(NB)]
print("no host sync between launches (events around each conv):")
sustained("same buffers every launch", lambda i: None, lambda i: K.igemm(x, wp, M, C, Kw, geom, dtype=T, out=out))
sustained("10 input / output / weight sets in turn (cold Infinity Cache)", lambda i: None,
          lambda i: K.igemm(xs[i % NB], ws[i % NB], M, C, Kw, geom, dtype=T, out=outs[i % NB]))
sustained("same buffers, input rewritten by a copy kernel before each launch", lambda i: x.copy_(x2),
          lambda i: K.igemm(x, wp, M, C, Kw, geom, dtype=T, out=out))
sustained("10 sets in turn, each input rewritten just before its launch", lambda i: xs[i % NB].copy_(x2),
          lambda i: K.igemm(xs[i % NB], ws[i % NB], M, C, Kw, geom, dtype=T, out=outs[i % NB]))
sustained("10 sets in turn, 512 MB of unrelated traffic before each launch", lambda i: big2.copy_(big),
          lambda i: K.igemm(xs[i % NB], ws[i % NB], M, C, Kw, geom, dtype=T, out=outs[i % NB]))
