"""Per-kernel-class times (live events, one single-stream step) of two schedule variants on one box, to explain a step-time gap."""
import importlib, os, sys, time, collections
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench
pkg = importlib.import_module("visual-question-answering-vqa-system_amd")
K = pkg.kernels
M = pkg.load_dropin()
data = bench.synth_batch(512, torch.device("cuda", 0), 1234)
for tag, kw in (("default", {}), ("no bn_finalize fusion", dict(fuse_bn_finalize=False))):
    model = M.VQAModel(compute_dtype="bf16", seed=1234).to("cuda").train()
    tr = pkg.trainer.HipTrainer(model)
    for k, v in kw.items():
        setattr(tr.engine, k, v)
    for _ in range(8):
        tr.step(*data)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        tr.step(*data)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / 20 * 1e3
    tr.engine.two_streams = False
    tr.step(*data)
    K.PROFILE = []
    tr.step(*data)
    torch.cuda.synchronize()
    agg = collections.defaultdict(float)
    for name, fl, e0, e1, nb in K.PROFILE:
        agg[name] += e0.elapsed_time(e1)
    K.PROFILE = None
    top = sorted(agg.items(), key=lambda kv: -kv[1])[:24]
    print(f"== {tag}: {el:.3f} ms/step overlapped; serial sum {sum(agg.values()):.3f} ms")
    for n, t in top:
        print(f"     {t:7.3f}  {n[:80]}")
    sys.stdout.flush()
    del tr, model
    torch.cuda.empty_cache()
