"""Host enqueue time vs GPU time of a train step, plain / forced reducer (one-rank RCCL group) / forced without the communication
stream.  Usage: python tools/host_time.py [B]"""
import importlib, os, sys, time
import torch
import torch.distributed as dist
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
pkg = importlib.import_module("visual-question-answering-vqa-system_amd")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
M = pkg.load_dropin()
data = bench.synth_batch(B, torch.device("cuda", 0), 1234)
USE_STREAM = os.environ.get("HT_STREAM", "0") == "1"
ctx_stream = torch.cuda.Stream() if USE_STREAM else torch.cuda.current_stream()
print("compute stream:", "pool (non-blocking)" if USE_STREAM else "legacy null stream", flush=True)
torch.cuda.set_stream(ctx_stream)
for tag, kw in (("plain", dict()), ("forced", dict(force_reducer=True)), ("forced, no comm stream", dict(force_reducer=True, overlap=False)), ("plain", dict())):
    model = M.VQAModel(compute_dtype="bf16", seed=1234).to("cuda").train()
    tr = pkg.trainer.HipTrainer(model, **kw)
    for _ in range(8):
        tr.step(*data)
    torch.cuda.synchronize()
    host = []
    t0 = time.perf_counter()
    for _ in range(30):
        h0 = time.perf_counter()
        tr.step(*data)
        host.append(time.perf_counter() - h0)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / 30
    host.sort()
    print(f"{tag:26s} step {el*1e3:7.3f} ms   host enqueue median {host[15]*1e3:7.3f} ms  min {host[0]*1e3:7.3f}  max {host[-1]*1e3:7.3f}", flush=True)
    del tr, model
    torch.cuda.empty_cache()
dist.destroy_process_group()
