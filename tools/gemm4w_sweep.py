"""gemm4w (four waves, one per SIMD) against gemm8p: where the LDS-DMA pieces of the next K tile are issued (measurement tool,
ablation build: VQA_G4_NP3 = pieces issued right behind the barrier, the rest in the next K step's MFMA rows).
    python tools/gemm4w_sweep.py"""
import importlib, os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import tools.build_ablation as A
A.build(); A.use()
pkg = importlib.import_module("visual-question-answering-vqa-system_amd")
L = pkg._lib
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


for M, N, Kd in [(4096, 4096, 4096), (8192, 8192, 8192), (100352, 256, 2304), (65536, 256, 4608)]:
    Am = torch.randn(M, Kd, device=dev).to(bf)
    Bm = torch.randn(N, Kd, device=dev).to(bf)
    C8 = torch.empty(M, N, device=dev, dtype=bf)
    fl = 2.0 * M * N * Kd
    t8 = timeit(lambda: L.call("vqa_gemm8p", Am.data_ptr(), Bm.data_ptr(), C8.data_ptr(), M, N, Kd))
    tt = timeit(lambda: torch.matmul(Am, Bm.t()))
    row = []
    for np3 in ("4", "8", "12", "16"):
        os.environ["VQA_G4_NP3"] = np3
        C4 = torch.empty(M, N, device=dev, dtype=bf)
        L.call("vqa_gemm4w", Am.data_ptr(), Bm.data_ptr(), C4.data_ptr(), M, N, Kd)
        torch.cuda.synchronize()
        assert torch.equal(C4, C8), np3
        t4 = timeit(lambda: L.call("vqa_gemm4w", Am.data_ptr(), Bm.data_ptr(), C4.data_ptr(), M, N, Kd))
        row.append(f"NP3={np3:>2s} {t4*1e6:7.1f} us {fl/t4/1e12:6.1f}")
    print(f"M={M:6d} N={N:5d} K={Kd:5d}: gemm8p {t8*1e6:7.1f} us {fl/t8/1e12:6.1f} TF/s | matmul {fl/tt/1e12:6.1f} | gemm4w " + " | ".join(row), flush=True)
