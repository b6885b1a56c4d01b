"""Where the main stream spends a train step (timing events at the segment reports of engine.backward; side streams on as usual).
Usage: python tools/phase_times.py [B]"""
import importlib, os, sys, collections
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench
pkg = importlib.import_module("visual-question-answering-vqa-system_amd")
from importlib import import_module
L = import_module("visual-question-answering-vqa-system_amd._lib")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
M = pkg.load_dropin()
model = M.VQAModel(compute_dtype="bf16", seed=1234).to("cuda").train()
tr = pkg.trainer.HipTrainer(model)
eng = tr.engine
images, ids, mask, answers = bench.synth_batch(B, torch.device("cuda", 0), 1234)
maskf = mask.float()
if len(sys.argv) > 2 and sys.argv[2] == "serial":
    eng.two_streams = False
for _ in range(8):
    tr.step(images, ids, mask, answers)
torch.cuda.synchronize()
acc = collections.defaultdict(float)
N = 20
G = tr.G
for it in range(N):
    marks = []
    def mark(name):
        e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((name, e))
    G.zero_()
    mark("start")
    eng.mark = mark
    logits, _, tape = eng.forward(images, ids, maskf, True, False, need_tape=True)
    eng.mark = None
    mark("forward: head")
    dl = torch.empty_like(logits); loss = torch.zeros(1, device="cuda"); ws = torch.empty(B, device="cuda")
    L.call("vqa_cross_entropy", 0, logits.data_ptr(), answers.data_ptr(), loss.data_ptr(), dl.data_ptr(), None, B, logits.shape[1], 1.0, None, ws.data_ptr())
    main = torch.cuda.current_stream()
    def on_seg(name, evs):
        if torch.cuda.current_stream() == main:      # (the text encoder reports on its side stream)
            mark("backward: " + name)
    eng.backward(tape, dl, G, on_segment=on_seg)
    mark("backward: joins (side streams)")
    L.call("vqa_sumsq", G.data_ptr(), G.numel(), tr.sumsq.data_ptr())
    mark("sumsq")
    torch.cuda.synchronize()
    for (n0, e0), (n1, e1) in zip(marks, marks[1:]):
        acc[n1] += e0.elapsed_time(e1)
    acc["TOTAL"] += marks[0][1].elapsed_time(marks[-1][1])
for k, v in acc.items():
    print(f"{k:40s} {v / N:8.3f} ms")
if len(sys.argv) > 2 and sys.argv[2] == "serial":
    sys.exit(0)
