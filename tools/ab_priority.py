"""Does a high-priority main stream shorten the step?  (The weight-gradient side stream competes with the data-gradient chain for CUs:
backward measures as the SUM of both.)  python tools/ab_priority.py"""
import importlib, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench
pkg = importlib.import_module("visual-question-answering-vqa-system_amd")
M = pkg.load_dropin()
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a", flush=True)
data = bench.synth_batch(512, torch.device("cuda", 0), 1234)
for rep in range(2):
    for tag, prio in (("main = null stream (normal)", None), ("main = high-priority stream", -1), ("main = pool stream (normal)", 0)):
        st = torch.cuda.current_stream() if prio is None else torch.cuda.Stream(priority=prio)
        with torch.cuda.stream(st):
            model = M.VQAModel(compute_dtype="bf16", seed=1234).to("cuda").train()
            tr = pkg.trainer.HipTrainer(model)
            for _ in range(8):
                tr.step(*data)
            st.synchronize()
            t0 = time.perf_counter()
            for _ in range(30):
                tr.step(*data)
            st.synchronize()
            print(f"{tag:34s} {(time.perf_counter() - t0) / 30 * 1e3:7.3f} ms/step", flush=True)
        del tr, model
        torch.cuda.synchronize(); torch.cuda.empty_cache()
