"""Inference stem: the one-launch kernel against the two-kernel path at B=512, 224x224 (measurement tool)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--ablation" in sys.argv:
    import tools.build_ablation as A
    A.use()
pkg = importlib.import_module("visual-question-answering-vqa-system_amd")
K, L = pkg.kernels, pkg._lib
dev = "cuda"
for B in (64, 512):
    H = W = 224
    img = torch.randn(B, 3, H, W, device=dev)
    w = torch.randn(64, 7, 7, 3, device=dev) * 0.1
    wst = torch.empty(64, 192, device=dev, dtype=torch.bfloat16)
    L.call("vqa_stem_pack", w.data_ptr(), wst.data_ptr())
    coef = torch.cat([torch.rand(64) + 0.5, torch.randn(64) * 0.5, torch.zeros(128)]).to(dev)
    x2 = torch.empty(B * 56 * 56, 64, device=dev, dtype=torch.bfloat16); idx = torch.empty(B * 56 * 56, 64, device=dev, dtype=torch.uint8)

    def two():
        y, _, _ = K.stem_conv(img, wst, B, H, W, False)
        L.call("vqa_stem_pool_fwd", 1, y.data_ptr(), coef.data_ptr(), x2.data_ptr(), idx.data_ptr(), B, 112, 112, 64)

    def one():
        K.stem_conv_pool(img, wst, coef, B, H, W)

    variants = [("conv + pool (two launches)", two, ""), ("one launch", one, "0")]
    if "--ablation" in sys.argv:
        variants += [("one launch, no MFMA loop", one, "1"), ("one launch, no BN/pool VALU", one, "2"), ("one launch, neither", one, "3"),
                     ("one launch, prologue only", one, "4"), ("one launch, prologue without the patch fill", one, "12"), ("one launch, no patch fill", one, "8")]
    for name, fn, dbg in variants:
        os.environ["VQA_STEMCP_DBG"] = dbg or "0"
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        print(f"B={B:4d} {name:28s} {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us", flush=True)
