"""The unchanged-caller loop (training/train.py:168-212 restated, as bench.py's extras.dropin_loop) on its own, for profiling:
    rocprofv3 --kernel-trace --stats -d gpurun_out/dropin -- python3 tools/dropin_loop.py
and, without the profiler, a breakdown of what the loop's pieces cost (each variant timed over the same batch)."""
import argparse
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--variants", default="full")
args = ap.parse_args()
pkg = importlib.import_module("visual-question-answering-vqa-system_amd")
M = pkg.load_dropin()
dev = "cuda"
torch.manual_seed(0)
model = M.VQAModel().to(dev).train()
B = args.batch
images = torch.randn(B, 3, 224, 224, device=dev)
ids = torch.randint(0, 1000, (B, 20), device=dev)
lens = torch.randint(5, 21, (B,), device=dev)
mask = (torch.arange(20, device=dev)[None] < lens[:, None]).long()
answers = torch.randint(0, 1000, (B,), device=dev)
crit = torch.nn.CrossEntropyLoss()


def run(name, sync=True, clip=True, opt_kind="adamw", steps=args.steps):
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=0.01, betas=(0.9, 0.999)) if opt_kind == "adamw" else None

    def one():
        if opt is not None:
            opt.zero_grad()
        else:
            model.zero_grad()
        logits, _ = model(images, ids, mask)
        loss = crit(logits, answers)
        loss.backward()
        if clip:
            torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
        if opt is not None:
            opt.step()
        if sync:
            loss.item()
            (logits.detach().argmax(dim=-1).cpu() == answers.cpu()).sum().item()
    for _ in range(5):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print(f"{name:58s} {ms:7.2f} ms/step  {B / ms:6.1f} k pairs/s", flush=True)


V = args.variants.split(",")
if "full" in V:
    run("train.py loop: AdamW + clip_grad_norm_ + 2 host syncs")
if "all" in V:
    run("  without the two host syncs", sync=False)
    run("  without clip_grad_norm_", clip=False)
    run("  without the optimizer (forward + CE + backward + clip)", opt_kind=None)
    run("  forward + CE + backward only", opt_kind=None, clip=False, sync=False)
