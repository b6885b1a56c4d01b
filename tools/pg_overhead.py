"""What does an initialised process group cost a (plain, single-rank) train step?  Each configuration runs in a fresh child process.
Usage: python tools/pg_overhead.py            (parent)   |   python tools/pg_overhead.py child <mode>"""
import importlib, os, subprocess, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

MODES = (sys.argv[1].split(",") if len(sys.argv) > 1 and sys.argv[1] != "child" else ["none", "gloo", "nccl", "nccl_destroyed", "nccl_nomonitor", "nccl_lazy", "nccl_used_once", "none"])


def child(mode):
    import torch
    import torch.distributed as dist
    import bench
    pkg = importlib.import_module("visual-question-answering-vqa-system_amd")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29545")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    mode = mode.split("@")[0]
    if mode == "gloo":
        dist.init_process_group("gloo", rank=0, world_size=1)
    elif mode.startswith("nccl") and mode != "nccl_lazy":
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    elif mode == "nccl_lazy":
        dist.init_process_group("nccl", rank=0, world_size=1)          # no device_id: the communicator is created at the first collective
    if mode == "nccl_used_once":
        t = torch.ones(1024, device=dev); dist.all_reduce(t); torch.cuda.synchronize()
    if mode == "nccl_destroyed":
        dist.destroy_process_group()
    M = pkg.load_dropin()
    data = bench.synth_batch(512, dev, 1234)
    model = M.VQAModel(compute_dtype="bf16", seed=1234).to("cuda").train()
    tr = pkg.trainer.HipTrainer(model)
    for _ in range(10):
        tr.step(*data)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        tr.step(*data)
    torch.cuda.synchronize()
    print(f"RESULT {mode:16s} {(time.perf_counter() - t0) / 40 * 1e3:7.3f} ms/step", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "child":
        child(sys.argv[2])
    else:
        for m in MODES:
            env = dict(os.environ)
            for kv in m.split("@")[1:]:                      # mode@VAR=value@VAR=value: extra environment
                k, v = kv.split("=", 1); env[k] = v
            if m == "nccl_nomonitor":
                env.update(TORCH_NCCL_ENABLE_MONITORING="0", TORCH_NCCL_ASYNC_ERROR_HANDLING="0", TORCH_NCCL_DUMP_ON_TIMEOUT="0")
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", m], env=env, capture_output=True, text=True, timeout=300)
            lines = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
            print(lines[0] if lines else f"RESULT {m} FAILED rc={r.returncode} {r.stderr[-300:]}", flush=True)
