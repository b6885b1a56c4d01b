"""Per-tensor err / moved of the skipped-step trainer test (tests/test_gpu_trainer.py) with the product library or another build:
    python tools/ratio_diag.py [product | path/to/libvqa_hip_other.so]"""
import os, sys, importlib
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
pkg = importlib.import_module("visual-question-answering-vqa-system_amd")
if len(sys.argv) > 1 and sys.argv[1] != "product":
    pkg._lib.LIB_PATH = os.path.abspath(sys.argv[1]); pkg._lib._lib = None
from oracle import vqa_oracle as O
DEV = "cuda"
cfg = O.full_config(dropout=0.0, answer_dropout=0.0, vocab_size=100, num_answers=10, embed_dim=32)
sd = O.init_state_dict(cfg, 21, jitter=True)
M = pkg.load_dropin()
m = M.VQAModel(**cfg, compute_dtype="fp32"); m.load_state_dict(sd); m = m.to(DEV).train()
tr = pkg.trainer.HipTrainer(m, lr=1e-3)
ot = O.OracleTrainer(sd, cfg, lr=1e-3)
names = O.parameter_names(cfg)
start = {n: sd[n].clone() for n in names}
for step in range(5):
    images, ids, mask, answers = O.synthetic_batch(4, seed=700 + step, image_size=64, seq_len=10, vocab=100, num_answers=10)
    if step == 1:
        answers[2] = 10
        nb = {}
        with torch.no_grad():
            O.vqa_forward(images, ids, mask, ot.sd, cfg, True, nb)
        ot.sd.update(nb)
    else:
        ot.step(images, ids, mask, answers)
    tr.step(images.to(DEV), ids.to(DEV), mask.to(DEV), answers.to(DEV))
torch.cuda.synchronize()
P = dict(m.named_parameters())
r = []
for n in names:
    moved = (ot.sd[n].detach() - start[n]).norm().item()
    err = (P[n].detach().cpu() - ot.sd[n].detach()).norm().item()
    r.append((err / (moved + 1e-12), n, err, moved))
r.sort(reverse=True)
print(sys.argv[1:] or "product")
for x in r[:8]:
    print("  %.4f  %s  err %.3e moved %.3e" % x)
