"""Grid caps of the BatchNorm passes whose workgroups finalize the statistics in their prologue (bn_apply_acc, bn_bwd_apply_acc) and of the
backward reduce: microseconds per launch on the stage shapes of the B = 512 bf16 step for each cap (ablation build: VQA_BN_GRID,
VQA_BNB_GRID, VQA_BNR_GRID).  Measurement tool.   python tools/bn_grid_sweep.py"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.build_ablation as A
A.build(); A.use()
pkg = importlib.import_module("visual-question-answering-vqa-system_amd")
K, L = pkg.kernels, pkg._lib
B, T, dev = 512, torch.bfloat16, "cuda"


def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for s, C, H in [(1, 64, 56), (2, 128, 28), (3, 256, 14), (4, 512, 7)]:
    HW, rows = H * H, B * H * H
    y = torch.randn(rows, C, device=dev).to(T); x = torch.relu(torch.randn(rows, C, device=dev)).to(T); d = torch.randn(rows, C, device=dev).to(T)
    gam, bet = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
    rm, rv, nbt = torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.zeros((), device=dev, dtype=torch.int64)
    bnp = (gam, bet, rm, rv, nbt)
    coef = torch.stack([gam, bet, torch.zeros(C, device=dev), torch.ones(C, device=dev)]).contiguous()
    acc = torch.zeros(L.count("vqa_bn_acc_words", 2, C), device=dev, dtype=torch.int64)
    facc = torch.zeros(L.count("vqa_bn_acc_words", 3, C), device=dev, dtype=torch.int64)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    line = f"s{s} C={C:3d}:"
    for cap in (2048, 1024, 512, 256):
        os.environ["VQA_BN_GRID"] = os.environ["VQA_BNB_GRID"] = str(cap)
        t1 = timeit(lambda: K.bn_apply_acc(y, acc, bnp, C, True, B, HW, rows))
        t2 = timeit(lambda: K.bn_apply_acc(y, acc, bnp, C, True, B, HW, rows, res=x))
        t3 = timeit(lambda: K.bn_bwd(d, x, y, coef, gam, C, True, dg, db, facc=facc, facc_filled=True))
        t3s = timeit(lambda: K.bn_bwd(d, None, y, coef, gam, C, True, dg, db, self_mask=True, facc=facc, facc_filled=True))
        line += f"  cap {cap:4d}: apply {t1:5.1f} +res {t2:5.1f} bwd_apply {t3:5.1f} self {t3s:5.1f} |"
    print(line, flush=True)
    line = f"s{s} bn_bwd_reduce (acc mode) cap:"
    for cap in (768, 512, 256):
        os.environ["VQA_BNR_GRID"] = str(cap)
        facc.zero_()
        t4 = timeit(lambda: pkg._lib.call("vqa_bn_bwd_reduce", 1, d.data_ptr(), x.data_ptr(), y.data_ptr(), coef.data_ptr(), None, None, facc.data_ptr(), rows, C, 0, 1))
        line += f"  {cap}: {t4:5.1f}"
    print(line, flush=True)
