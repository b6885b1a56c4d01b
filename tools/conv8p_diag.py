"""Where does a K tile of conv8p go?  (measurement tool, ablation build, WRONG results on purpose)
VQA_C8P_DBG: 1 = every A piece of a half-tile re-fetches the first piece's source (same instruction count and LDS writes, no distinct
lines from L2), 2 = the same for B, 4 = no MFMAs.  Stage-2 (448 x 128 tile), stage-3 and stage-4 (224 x 256) shapes at B = 512.
    python tools/conv8p_diag.py"""
import importlib
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import tools.build_ablation as A
A.build(); A.use()
pkg = importlib.import_module("visual-question-answering-vqa-system_amd")
K = pkg.kernels
B, T, dev = 512, torch.bfloat16, "cuda"


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for name, C, H in (("s2 128->128 28", 128, 28), ("s3 256->256 14", 256, 14), ("s4 512->512 7", 512, 7)):
    x = torch.randn(B * H * H, C, device=dev).to(T)
    w = (torch.randn(C, 9 * C, device=dev) * 0.03).to(T)
    row = []
    for dbg in (0, 1, 2, 3, 4, 7):
        os.environ["VQA_C8P_DBG"] = str(dbg)
        row.append(timeit(lambda: K.conv8p(x, w, B, H, H, C, C, transposed=1)))
    row2 = []
    for dbg in (8, 16, 24, 28):              # 8 / 16: every A / B piece moves ONE lane's 16 bytes (same instructions, waits and barriers; 1/64 of the bytes)
        os.environ["VQA_C8P_DBG"] = str(dbg)
        row2.append(timeit(lambda: K.conv8p(x, w, B, H, H, C, C, transposed=1)))
    print(f"{name}  LDS-DMA bytes: A pieces one lane {row2[0]:6.1f} us | B pieces one lane {row2[1]:6.1f} | both {row2[2]:6.1f} | both, no MFMA {row2[3]:6.1f}", flush=True)
    os.environ["VQA_C8P_DBG"] = "0"
    print(f"{name}  data gradient: as shipped {row[0]:6.1f} us | A from one line {row[1]:6.1f} | B from one line {row[2]:6.1f} | both {row[3]:6.1f} | "
          f"no MFMA {row[4]:6.1f} | no MFMA, both from one line {row[5]:6.1f}", flush=True)
