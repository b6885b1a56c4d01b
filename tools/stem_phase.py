"""What the main stream does between the start of a train step and the end of the stem (events around begin_step and every launch of the stem),
in the real step order (HipTrainer.step, AdamW of the previous step in front).  Usage: python tools/stem_phase.py"""
import importlib, os, sys, collections
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench
pkg = importlib.import_module("visual-question-answering-vqa-system_amd")
L = importlib.import_module("visual-question-answering-vqa-system_amd._lib")
M = pkg.load_dropin()
model = M.VQAModel(compute_dtype="bf16", seed=1234).to("cuda").train()
tr = pkg.trainer.HipTrainer(model)
eng = tr.engine
data = bench.synth_batch(512, torch.device("cuda", 0), 1234)
for _ in range(8):
    tr.step(*data)
torch.cuda.synchronize()
acc = collections.defaultdict(float)
marks = []
def ev(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((name, e))
orig_call = L.call
K = importlib.import_module("visual-question-answering-vqa-system_amd.kernels")
ENG = importlib.import_module("visual-question-answering-vqa-system_amd.engine")
state = {"on": False}
def traced(name, *a):
    if state["on"] and torch.cuda.current_stream() == main:
        ev("before " + name)
    r = orig_call(name, *a)
    if state["on"] and torch.cuda.current_stream() == main:
        ev("after  " + name)
    return r
main = torch.cuda.current_stream()
for mod in (L, K, ENG):
    if hasattr(mod, "call"):
        mod.call = traced
orig_begin = eng.begin_step
def begin(*a, **k):
    ev("begin_step in"); r = orig_begin(*a, **k); ev("begin_step out"); return r
eng.begin_step = begin
N = 10
for it in range(N):
    marks.clear()
    state["on"] = True
    eng.mark = lambda n: (ev("MARK " + n), state.__setitem__("on", False)) if n == "forward: stem" else None
    ev("step start")
    tr.step(*data)
    torch.cuda.synchronize()
    names = [m[0] for m in marks]
    for (n0, e0), (n1, e1) in zip(marks, marks[1:]):
        acc[f"{n0:40s} -> {n1}"] += e0.elapsed_time(e1)
    acc["TOTAL start -> stem mark"] += marks[0][1].elapsed_time(marks[-1][1])
for k, v in acc.items():
    print(f"{k:90s} {v / N * 1e3:8.1f} us")
