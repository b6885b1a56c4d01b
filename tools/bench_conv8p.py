"""conv8p launches of the B = 512 step, isolated, for one or more builds of the kernel library (measurement tool):
    python tools/bench_conv8p.py [lib.so ...]          # the product library first, then each library given
Forward with BatchNorm statistics and the plain stride-1 data gradient of the stage 2 / 3 / 4 3x3 convs, plus the data gradient with
epilogue inputs (identity gradient + mask) for reference."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("visual-question-answering-vqa-system_amd")
K, L = pkg.kernels, pkg._lib
dev, bf = "cuda", torch.bfloat16


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def run(tag):
    print(f"--- {tag}", flush=True)
    for name, C, H in [("stage2 128->128 28x28", 128, 28), ("stage3 256->256 14x14", 256, 14), ("stage4 512->512 7x7", 512, 7)]:
        Bb = 512
        M = Bb * H * H
        x = torch.randn(M, C, device=dev).to(bf)
        w = (torch.randn(C, 9 * C, device=dev) * 0.03).to(bf)
        add = torch.randn(M, C, device=dev).to(bf)
        words = L.count("vqa_bn_acc_words", 2, C)
        acc = torch.zeros(words, device=dev, dtype=torch.int64)
        fl = 2.0 * M * C * 9 * C
        tf = timeit(lambda: K.conv8p(x, w, Bb, H, H, C, C, transposed=0, stats_acc=acc))
        tp = timeit(lambda: K.conv8p(x, w, Bb, H, H, C, C, transposed=0))
        td = timeit(lambda: K.conv8p(x, w, Bb, H, H, C, C, transposed=1))
        te = timeit(lambda: K.conv8p(x, w, Bb, H, H, C, C, transposed=1, addend=add, outmask=add))
        print(f"{name}: fwd+stats {tf*1e6:7.1f} us {fl/tf/1e12:7.1f} TF/s | fwd plain {tp*1e6:7.1f} | dgrad plain {td*1e6:7.1f} us {fl/td/1e12:7.1f} TF/s"
              f" | dgrad + addend + outmask {te*1e6:7.1f}", flush=True)


run("product library")
for lib in sys.argv[1:]:
    L.LIB_PATH = os.path.abspath(lib)
    L._lib = None
    run(lib)
