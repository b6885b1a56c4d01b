"""Host-side timeline of the unchanged train.py loop on the drop-in model (measurement tool): where the HOST spends its time per step,
in particular how long after a step's sync the first kernel of the next forward goes out (that stretch is idle GPU time)."""
import importlib, os, sys, time, torch
sys.path.insert(0, "/root/repo")
pkg = importlib.import_module("visual-question-answering-vqa-system_amd")
import bench
M = pkg.load_dropin()
dev = "cuda"
model = M.VQAModel(seed=1).to(dev).train()
images, ids, mask, answers = bench.synth_batch(512, torch.device(dev), 1)
crit = torch.nn.CrossEntropyLoss()
opt = torch.optim.AdamW(model.parameters(), lr=1e-4)
eng = model._ensure_engine()
T = {}
orig = eng.forward
def fwd(*a, **k):
    T["enter"] = time.perf_counter()
    r = orig(*a, **k)
    T["exit"] = time.perf_counter()
    return r
eng.forward = fwd
L = pkg._lib
_first = {"t": None, "name": None}
def hook(name, args):
    if _first["t"] is None:
        _first["t"] = time.perf_counter(); _first["name"] = name
    return None
L._HOOK[0] = hook
orig_b = eng.backward
def bwd(*a, **k):
    T["b_enter"] = time.perf_counter()
    r = orig_b(*a, **k)
    T["b_exit"] = time.perf_counter()
    return r
eng.backward = bwd
acc = {}
def add(k, v): acc.setdefault(k, []).append(v * 1e6)
for it in range(25):
    _first["t"] = None
    t0 = time.perf_counter(); opt.zero_grad(); t1 = time.perf_counter()
    logits, _ = model(images, ids, mask); t2 = time.perf_counter()
    loss = crit(logits, answers); t3 = time.perf_counter()
    loss.backward(); t4 = time.perf_counter()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0); t5 = time.perf_counter()
    opt.step(); t6 = time.perf_counter()
    loss.item(); t7 = time.perf_counter()
    (logits.detach().argmax(-1).cpu() == answers.cpu()).sum().item(); t8 = time.perf_counter()
    if it >= 5:
        add("zero_grad", t1 - t0); add("model() before engine.forward", T["enter"] - t1); add("engine.forward (host)", T["exit"] - T["enter"])
        add("  engine.forward until its first C-ABI launch (" + str(_first["name"]) + ")", _first["t"] - T["enter"])
        add("model() after engine.forward", t2 - T["exit"]); add("CE", t3 - t2); add("backward() before engine.backward", T["b_enter"] - t3)
        add("engine.backward (host)", T["b_exit"] - T["b_enter"]); add("backward() after engine.backward", t4 - T["b_exit"])
        add("clip_grad_norm_", t5 - t4); add("opt.step", t6 - t5); add("loss.item() wait", t7 - t6); add("argmax/cpu compare", t8 - t7)
for k, v in acc.items():
    v.sort(); print(f"{k:70s} {v[len(v)//2]:9.1f} us")
