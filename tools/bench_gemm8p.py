"""The 8-phase 256x256x64 GEMM core against torch.matmul (hipBLASLt) and the 128x128 igemm tile, dense shapes (measurement tool).
    python tools/bench_gemm8p.py"""
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


for M, N, Kd in [(4096, 4096, 4096), (8192, 8192, 8192), (100352, 256, 2304), (65536, 256, 2304)]:
    A = torch.randn(M, Kd, device=dev).to(bf)
    B = torch.randn(N, Kd, device=dev).to(bf)
    C = torch.empty(M, N, device=dev, dtype=bf)
    L.call("vqa_gemm8p", A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, Kd)
    torch.cuda.synchronize()
    rows = torch.randint(0, M, (64,), device=dev)
    ref = A[rows].float() @ B.float().t()
    err = (C[rows].float() - ref).abs().max().item() / ref.abs().max().item()
    fl = 2.0 * M * N * Kd
    t8 = timeit(lambda: L.call("vqa_gemm8p", A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, Kd))
    t4 = timeit(lambda: L.call("vqa_gemm4w", A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, Kd))
    tt = timeit(lambda: torch.matmul(A, B.t()))
    ti = timeit(lambda: K.igemm(A, B, M, N, Kd, K.linear_geom(M, Kd), dtype=bf, out=C))
    print(f"M={M:6d} N={N:5d} K={Kd:5d}: gemm8p {t8*1e6:8.1f} us {fl/t8/1e12:7.1f} TF/s | gemm4w {t4*1e6:8.1f} us {fl/t4/1e12:7.1f} TF/s | torch.matmul {tt*1e6:8.1f} us {fl/tt/1e12:7.1f} | igemm 128x128 {ti*1e6:8.1f} us {fl/ti/1e12:7.1f} | rel err {err:.2e}", flush=True)

print("--- 3x3 convs at B=512 (forward with BN statistics; data gradient plain): conv8p vs the 128x128 window-loader igemm")
for name, C, H in [("stage2 128->128 28x28", 128, 28), ("stage3 256->256 14x14", 256, 14), ("stage4 512->512 7x7", 512, 7)]:
    Bb = 512
    M = Bb * H * H
    x = torch.randn(M, C, device=dev).to(bf)
    w = (torch.randn(C, 9 * C, device=dev) * 0.03).to(bf)
    geom = (Bb, H, H, C, H, H, 3, 3, 1, 1)
    words = L.count("vqa_bn_acc_words", 2, C)
    fl = 2.0 * M * C * 9 * C
    for tr in (0, 1):
        t8 = timeit(lambda: K.conv8p(x, w, Bb, H, H, C, C, transposed=tr, stats_acc=None if tr else torch.zeros(words, device=dev, dtype=torch.int64)))
        ti = timeit(lambda: K.igemm(x, w, M, C, 9 * C, geom, dtype=bf, transposed=tr, want_stats=not tr,
                                    stats_acc=None if tr else torch.zeros(words, device=dev, dtype=torch.int64)))
        print(f"{name} {'dgrad' if tr else 'fwd  '}: conv8p {t8*1e6:7.1f} us {fl/t8/1e12:7.1f} TF/s | igemm {ti*1e6:7.1f} us {fl/ti/1e12:7.1f} TF/s", flush=True)
