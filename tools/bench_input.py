"""Input-pipeline kernels: achieved HBM GB/s of vqa_image_normalize at the benchmark geometry and the rate of
the device token packer.  Not part of the bench.py contract; numbers are quoted in DESIGN.md."""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("visual-question-answering-vqa-system_amd")
P, T = pkg.load_dropin_preprocess(), pkg.load_dropin_tokenizer()
dev = "cuda"
norm = P.DeviceImageNormalizer()
for B in (64, 512):
    img = torch.randint(0, 256, (B, 224, 224, 3), dtype=torch.uint8, device=dev)
    flip = (torch.rand(B) < 0.5).to(dev)
    for _ in range(3):
        norm(img, flip)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        norm(img, flip)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e-3
    nbytes = img.numel() * (1 + 4)
    print(f"image_normalize B={B}: {t*1e6:8.1f} us  {nbytes/t/1e12:5.2f} TB/s (u8 in + f32 out = {nbytes/1e6:.0f} MB)  {B/t/1e6:.2f} M images/s")
tok = T.Tokenizer(max_length=20, vocab_size=10000)
words = [f"w{i}" for i in range(5000)]
import random
random.seed(1)
qs = [" ".join(random.choices(words, k=random.randint(3, 12))) + "?" for _ in range(512)]
tok.build_vocab(qs, min_freq=1)
for _ in range(3):
    tok.batch_encode_device(qs, dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    ids, mask = tok.batch_encode_device(qs, dev)
torch.cuda.synchronize()
t1 = (time.perf_counter() - t0) / 20
t0 = time.perf_counter()
for _ in range(20):
    a, b = tok.batch_encode(qs)
    ids2 = torch.tensor(a).to(dev); mask2 = torch.tensor(b).to(dev)
torch.cuda.synchronize()
t2 = (time.perf_counter() - t0) / 20
print(f"tokenise 512 questions -> device tensors: device packer {t1*1e3:.2f} ms, host lists + torch.tensor {t2*1e3:.2f} ms; equal: {torch.equal(ids, ids2) and torch.equal(mask, mask2)}")
