"""Upper bound of what folding the BatchNorm finalize launches into their consumers could save: time the step with the 60 finalize
launches replaced by cached coefficients of an earlier step (WRONG numerics, measurement only)."""
import importlib, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench
pkg = importlib.import_module("visual-question-answering-vqa-system_amd")
K = pkg.kernels
M = pkg.load_dropin()
data = bench.synth_batch(512, torch.device("cuda", 0), 1234)
model = M.VQAModel(compute_dtype="bf16", seed=1234).to("cuda").train()
tr = pkg.trainer.HipTrainer(model)

def timeit(n=30):
    for _ in range(5):
        tr.step(*data)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        tr.step(*data)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

base = timeit()
orig_coef, orig_call = K.bn_train_coef, K.call
cache = {}
def fake_coef(stats, mtiles, C, count, gamma, beta, rm, rv, nbt, momentum=0.1, eps=1e-5):
    key = gamma.data_ptr()
    if key not in cache:
        cache[key] = orig_coef(stats, mtiles, C, count, gamma, beta, rm, rv, nbt, momentum, eps)
    return cache[key]
K.bn_train_coef = fake_coef
a = timeit()
bc_cache = {}
def fake_call(name, *args):
    if name == "vqa_bn_bwd_finalize":
        key = (args[5], args[3])          # gamma ptr, which
        if key in bc_cache:
            return
        bc_cache[key] = True
    return orig_call(name, *args)
K.call = fake_call
b = timeit()
print(f"RESULT baseline {base:.3f} ms   no stats-finalize {a:.3f} ms   no finalize at all {b:.3f} ms")
