"""Where a barrier interval of conv8p goes: s_memtime stamps at three points of each of the four phases (ablation build, VQA_C8P_DBG = 32),
summed over the K loop of workgroup 0, per wave.  Read the shares (a stamp costs ~40 cycles and drains the LDS reads in flight).
    python tools/conv8p_stamps.py"""
import ctypes, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.build_ablation as A
A.build(); A.use()
pkg = importlib.import_module("visual-question-answering-vqa-system_amd")
K, L = pkg.kernels, pkg._lib
B, T, dev = 512, torch.bfloat16, "cuda"
lib = ctypes.CDLL(L.LIB_PATH)
for name, C, H in (("s4 512->512 7 (224x256 tile, 72 K tiles)", 512, 7), ("s3 256->256 14 (224x256 tile, 36 K tiles)", 256, 14), ("s2 128->128 28 (448x128 tile, 18 K tiles)", 128, 28)):
    x = torch.randn(B * H * H, C, device=dev).to(T)
    w = (torch.randn(C, 9 * C, device=dev) * 0.03).to(T)
    nkt = 9 * C // 64
    os.environ["VQA_C8P_DBG"] = "32"
    for _ in range(3):
        K.conv8p(x, w, B, H, H, C, C, transposed=1)
    torch.cuda.synchronize()
    buf = (ctypes.c_uint * 160)()
    assert lib.vqa_conv8p_stamps(buf) == 0
    os.environ["VQA_C8P_DBG"] = "0"
    print(f"--- {name}: cycles per K tile and wave (load segment + mid barrier + fragment wait | MFMA segment | end barrier) per phase")
    for wv in range(8):
        v = [buf[wv * 20 + i] / nkt for i in range(12)]
        tot = sum(v)
        print(f"wave {wv} (group {wv >> 2}): " + "  ".join(f"P{p + 1} {v[3*p]:5.0f} | {v[3*p+1]:4.0f} | {v[3*p+2]:4.0f}" for p in range(4)) + f"   total {tot:6.0f}")

    # second pass: stamps around the counted vmcnt wait of each load segment only (VQA_C8P_DBG = 64): cycles from the previous stamp (= the end of the
    # previous phase's wait ... i.e. a whole phase) | the wait itself
    os.environ["VQA_C8P_DBG"] = "64"
    for _ in range(3):
        K.conv8p(x, w, B, H, H, C, C, transposed=1)
    torch.cuda.synchronize()
    assert lib.vqa_conv8p_stamps(buf) == 0
    os.environ["VQA_C8P_DBG"] = "0"
    for wv in (0, 4):
        v = [buf[wv * 20 + i] / nkt for i in range(12, 20)]
        print(f"wave {wv}: counted vmcnt wait per phase (cycles incl. ~40 of the stamp): " + "  ".join(f"P{p + 1} {v[4 + p]:5.0f} (rest of the K tile's quarter {v[p]:5.0f})" for p in range(4)))

    # third pass: raw s_memtime inside P1's load segment (VQA_C8P_DBG = 128; no waits of their own): 4 fragment reads issued | 2 DMA pieces issued | counted
    # vmcnt wait | mid barrier (waiting for the partner's MFMA segment) + the stamp's lgkmcnt(0)
    os.environ["VQA_C8P_DBG"] = "128"
    for _ in range(3):
        K.conv8p(x, w, B, H, H, C, C, transposed=1)
    torch.cuda.synchronize()
    assert lib.vqa_conv8p_stamps(buf) == 0
    os.environ["VQA_C8P_DBG"] = "0"
    for wv in (0, 1, 4, 5):
        v = [buf[wv * 20 + i] / nkt for i in range(4)]
        print(f"wave {wv}: P1 load segment: 4 ds_read_b128 issued {v[0]:5.0f} | 2 LDS-DMA pieces issued {v[1]:5.0f} | vmcnt wait {v[2]:5.0f} | barrier + fragment wait {v[3]:5.0f} cycles")
