"""A/B of the engine's schedule switches on one box: python tools/ab_fusions.py  (B=512 bf16, 30 timed steps each, repeated twice)."""
import importlib, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench
pkg = importlib.import_module("visual-question-answering-vqa-system_amd")
M = pkg.load_dropin()
data = bench.synth_batch(512, torch.device("cuda", 0), 1234)
CASES = [("default", {}), ("no act mask in the dgrad epilogue", dict(fuse_act_dgrad=False)), ("no chain trim (column sums / text add on the chain)", dict(chain_trim=False)), ("neither", dict(chain_trim=False, fuse_act_dgrad=False)), ("backward operands packed on the main stream", dict(pack_off_main=False)), ("handed bn2 reduce in the conv1 dgrad epilogue", dict(fuse_hand_reduce=True)), ("cross-attention q / kv paths not hoisted", dict(hoist_cross=False)), ("no c64p epilogue variant (igemm 128x64)", dict(use_c64p_epi=False)), ("no bn1 reduce in the dgrad epilogue", dict(fuse_bn1_reduce=False)), ("conv8p not for stage 2 (igemm there)", dict(conv8p_n_multiple=256, conv8p_bwd_n_multiple=256)), ("no conv8p (igemm 128x128)", dict(use_conv8p=False)), ("conv8p forward only", dict(use_conv8p_bwd=False)), ("no fuse_bn_conv (a1 stored)", dict(fuse_bn_conv=False)), ("no bn_finalize fusion", dict(fuse_bn_finalize=False)), ("no se_pool", dict(fuse_se_pool=False)),
         ("no se_bnred", dict(fuse_se_bnred=False)), ("none of the three", dict(fuse_se_pool=False, fuse_se_bnred=False, fuse_bn_finalize=False))]
NC = int(os.environ.get("AB_CASES", "4"))
if os.environ.get("AB_ONLY"):          # e.g. AB_ONLY=0,3: the cases with these indices
    CASES = [CASES[int(i)] for i in os.environ["AB_ONLY"].split(",")]
    NC = len(CASES)
for rep in range(int(os.environ.get("AB_REPS", "2"))):
    for tag, kw in CASES[:NC]:
        model = M.VQAModel(compute_dtype="bf16", seed=1234).to("cuda").train()
        tr = pkg.trainer.HipTrainer(model)
        for k, v in kw.items():
            setattr(tr.engine, k, v)
        for _ in range(8):
            tr.step(*data)
        torch.cuda.synchronize()
        st0 = torch.cuda.memory_stats()
        t0 = time.perf_counter()
        host = []
        for _ in range(int(os.environ.get('AB_STEPS', '30'))):
            h0 = time.perf_counter()
            tr.step(*data)
            host.append(time.perf_counter() - h0)
        torch.cuda.synchronize()
        st1 = torch.cuda.memory_stats()
        host.sort()
        print(f"{tag:28s} {(time.perf_counter() - t0) / 30 * 1e3:7.3f} ms/step   host median {host[len(host)//2]*1e3:6.2f} max {host[-1]*1e3:6.2f}  "
              f"device allocs +{st1['num_device_alloc'] - st0['num_device_alloc']} frees +{st1['num_device_free'] - st0['num_device_free']} "
              f"retries +{st1['num_alloc_retries'] - st0['num_alloc_retries']} reserved {st1['reserved_bytes.all.current'] / 2**30:.1f} GiB", flush=True)
        del tr, model
        torch.cuda.empty_cache()
