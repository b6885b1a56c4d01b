"""The launcher on the GPU: a project laid out like the reference whose OWN models/vqa_model.py raises on import, an entry script that
puts the project root first on sys.path and then imports the model (training/train.py:41-49) and runs the reference's step recipe
(training/train.py:168-212: zero_grad -> forward -> CrossEntropyLoss -> backward -> clip_grad_norm_ -> AdamW.step) -- through
`python run_reference.py training/train.py` it must get the HIP model, and libvqa_hip.so must be the code that ran."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

from test_binding_cpu import REPO, _fake_root

pytestmark = pytest.mark.gpu

_STEP = textwrap.dedent("""
    import sys, json
    from pathlib import Path
    PROJECT_ROOT = Path(__file__).parent.parent
    sys.path.insert(0, str(PROJECT_ROOT))
    from models.vqa_model import VQAModel, create_vqa_model
    from utils.config import WHO
    import torch, torch.nn as nn, torch.optim as optim

    if __name__ == "__main__":
        torch.manual_seed(0)
        device = "cuda"
        model = create_vqa_model(vocab_size=1000, num_answers=100).to(device)
        criterion = nn.CrossEntropyLoss()
        optimizer = optim.AdamW(model.parameters(), lr=1e-4, weight_decay=0.01, betas=(0.9, 0.999))
        images = torch.randn(8, 3, 224, 224, device=device)
        ids = torch.randint(1, 1000, (8, 20), device=device)
        mask = torch.ones(8, 20, dtype=torch.long, device=device)
        answers = torch.randint(0, 100, (8,), device=device)
        losses = []
        model.train()
        first = next(iter(model.parameters())).detach().clone()
        for _ in range(3):
            optimizer.zero_grad()
            logits, _ = model(images, ids, mask)
            loss = criterion(logits, answers)
            loss.backward()
            gn = float(torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0))
            optimizer.step()
            losses.append(loss.item())
        maps = open("/proc/self/maps").read()
        print(json.dumps(dict(model_file=sys.modules["models.vqa_model"].__file__, who=WHO, losses=losses, grad_norm=gn,
                              moved=float((next(iter(model.parameters())).detach() - first).abs().max()),
                              native="libvqa_hip.so" in maps, ops=hasattr(torch.ops.vqa_hip, "vqa_forward"),
                              n_opt=sum(len(g["params"]) for g in optimizer.param_groups))))
""")


def test_launcher_runs_a_train_step_on_the_hip_model(tmp_path):
    root = _fake_root(tmp_path)
    (root / "training" / "train.py").write_text(_STEP)
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    env.pop("PYTHONPATH", None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "run_reference.py"), "--dtype=fp32", "training/train.py"],
                       cwd=str(root), env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["model_file"].endswith("dropin/models/vqa_model.py") and out["who"] == "project utils.config"
    assert out["native"] and out["ops"] and out["n_opt"] == 164
    assert all(l == l and 0 < l < 20 for l in out["losses"]) and out["losses"][-1] < out["losses"][0]
    assert out["grad_norm"] > 0 and out["moved"] > 0
