"""Where do the small device copies of a train step come from?  (torch.profiler with stacks; diagnostic only)"""
import importlib, os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from torch.profiler import profile, ProfilerActivity
pkg = importlib.import_module("visual-question-answering-vqa-system_amd")
M = pkg.load_dropin()
dev = torch.device("cuda", 0)
model = M.VQAModel(compute_dtype="bf16", seed=1234).to(dev).train()
tr = pkg.trainer.HipTrainer(model)
batch = bench.synth_batch(64, dev, 1)
for _ in range(3):
    tr.step(*batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.step(*batch)
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    n = ev.name
    if "emcpy" in n or "emset" in n or n.startswith("aten::"):
        st = [s for s in (ev.stack or []) if "visual-question" in s or "bench.py" in s]
        cnt[(n, st[0] if st else "?")] += 1
for (n, s), c in sorted(cnt.items(), key=lambda kv: -kv[1])[:60]:
    print(c, n, s)
