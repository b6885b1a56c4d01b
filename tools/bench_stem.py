"""Time the fused stem weight-gradient kernel alone at B=512 (not part of the product)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("visual-question-answering-vqa-system_amd")
if os.environ.get("VQA_TOOLS_LIB"):                      # A/B against another build of the library (tools/_build/...)
    pkg._lib.LIB_PATH = os.environ["VQA_TOOLS_LIB"]; pkg._lib._lib = None
import bench
dev = torch.device("cuda:0")
model = pkg.load_dropin().VQAModel(compute_dtype="bf16", seed=1234).to(dev).train()
eng = model._ensure_engine()
B = 512
images, ids, mask, answers = bench.synth_batch(B, dev, 1234)
_, _, tape = eng.forward(images, ids, mask.float(), True, False, need_tape=True)
dxc = torch.randn(B * 56 * 56, 64, device=dev).to(torch.bfloat16)
G = torch.zeros_like(model._flat)
for _ in range(2):
    eng._stem_bwd(tape, dxc, G, True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30):
    eng._stem_bwd(tape, dxc, G, True)
e1.record(); torch.cuda.synchronize()
print(f"stem backward (reduce + finalize + fused weight gradient + slab reduce): {e0.elapsed_time(e1) / 30 * 1e3:.1f} us   [{os.environ.get('VQA_TOOLS_LIB', 'product library')}]")
