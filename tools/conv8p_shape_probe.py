import importlib, os, sys
import torch
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("visual-question-answering-vqa-system_amd")
K, L = pkg.kernels, pkg._lib
dev, bf = "cuda", torch.bfloat16
def timeit(fn, iters=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
for rep in range(2):
  for name, Bb, H, C, N in [("<4,2> C=128 28x28 B512 (stage 2)", 512, 28, 128, 128), ("<4,2> C=256 28x28 B256", 256, 28, 256, 128), ("<4,2> C=512 28x28 B128", 128, 28, 512, 128),
                          ("<2,4> C=128 14x14 B512", 512, 14, 128, 256), ("<2,4> C=256 14x14 B512 (stage 3)", 512, 14, 256, 256), ("<2,4> C=128 14x14 B1024", 1024, 14, 128, 256),
                          ("<4,2> C=64 28x28 B1024", 1024, 28, 64, 128)]:
    M = Bb * H * H
    x = torch.randn(M, C, device=dev).to(bf)
    w = (torch.randn(N, 9 * C, device=dev) * 0.03).to(bf)
    fl = 2.0 * M * N * 9 * C
    t = timeit(lambda: K.conv8p(x, w, Bb, H, H, C, N, transposed=0))
    tiles = M // (392 if N == 128 else 196)
    nkt = 9 * C // 64
    per_cu = tiles / 256 * nkt
    print(f"{name:36s} {t*1e6:7.1f} us {fl/t/1e12:7.1f} TF/s | tiles/CU {tiles/256:.2f} x {nkt} K tiles = {per_cu:.0f} -> {t*1e6/per_cu:.2f} us per K tile", flush=True)
