"""GPU: the device half of the input pipeline through the C ABI.
  vqa_pack_tokens      == the REAL reference Tokenizer's batch_encode (golden fixture), bit-exact (index work)
  vqa_image_normalize  == ToTensor + Normalize restated with torch (oracle.input_oracle; torchvision is absent in the build
                          container, so this half is pinned to the restatement only), bit-exact, at 224x224 and with flips
and the drop-in surfaces built on them (Tokenizer.batch_encode_device, DeviceImageNormalizer, gpu_collate_fn) feeding the model."""
import json
import os

import numpy as np
import pytest
import torch

from _pkg import pkg
from oracle import input_oracle as IO
from oracle import vqa_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "input_pipeline.npz"))


@pytest.mark.parametrize("tag,L", [("l20", 20), ("l8", 8)])
def test_device_token_packing_matches_reference_tokenizer(gold, tag, L):
    T = pkg().load_dropin_tokenizer()
    qs = json.loads(str(gold["questions"]))
    tok = T.Tokenizer(max_length=L, vocab_size=40)
    tok.word2idx = json.loads(str(gold[f"vocab_{tag}"]))
    ids, mask = tok.batch_encode_device(qs, DEV)
    assert ids.dtype == torch.int64 and mask.dtype == torch.int64 and ids.shape == (len(qs), L)
    assert np.array_equal(ids.cpu().numpy(), gold[f"ids_{tag}"]) and np.array_equal(mask.cpu().numpy(), gold[f"mask_{tag}"])
    ids, mask = tok.batch_encode_device(qs, DEV, add_special_tokens=False)
    assert np.array_equal(ids.cpu().numpy(), gold[f"ids_ns_{tag}"]) and np.array_equal(mask.cpu().numpy(), gold[f"mask_ns_{tag}"])


def test_device_token_packing_large_ragged_batch_matches_oracle():
    T = pkg().load_dropin_tokenizer()
    rng = np.random.default_rng(5)
    words = [f"w{i}" for i in range(300)]
    qs = [" ".join(rng.choice(words, size=int(n))) for n in rng.integers(0, 40, size=4096)]
    tok = T.Tokenizer(max_length=20, vocab_size=260)
    tok.build_vocab(qs, min_freq=1)
    ids, mask = tok.batch_encode_device(qs, DEV)
    rid, rmask = IO.batch_encode(qs, tok.word2idx, 20)
    assert np.array_equal(ids.cpu().numpy(), rid) and np.array_equal(mask.cpu().numpy(), rmask)
    assert (ids == 1).any()                                       # some words fell outside the 260-entry vocabulary -> UNK


def test_image_normalize_bit_exact(gold):
    P = pkg().load_dropin_preprocess()
    norm = P.DeviceImageNormalizer()
    img = torch.from_numpy(gold["img_u8"]).to(DEV)
    assert np.array_equal(norm(img).cpu().numpy(), gold["img_norm"])
    assert np.array_equal(norm(img, torch.from_numpy(gold["img_flip"])).cpu().numpy(), gold["img_norm_flip"])
    # benchmark geometry, random flips: every value, bit for bit
    g = torch.Generator().manual_seed(7)
    big = torch.randint(0, 256, (32, 224, 224, 3), generator=g, dtype=torch.uint8)
    flip = torch.rand(32, generator=g) < 0.5
    out = norm(big.to(DEV), flip)
    assert out.shape == (32, 3, 224, 224) and torch.equal(out.cpu(), IO.to_tensor_normalize(big, flip))
    with pytest.raises(RuntimeError):
        norm(big)                                                  # host tensor: no CPU fallback
    with pytest.raises(RuntimeError):
        norm(big.to(DEV).float())


def test_gpu_collate_feeds_the_model():
    P, T = pkg().load_dropin_preprocess(), pkg().load_dropin_tokenizer()
    cfg = O.full_config(vocab_size=100, num_answers=10, embed_dim=32)
    m = pkg().load_dropin().VQAModel(**cfg, compute_dtype="fp32")
    m.load_state_dict(O.init_state_dict(cfg, 3))
    m = m.to(DEV).eval()
    tok = T.Tokenizer(max_length=20, vocab_size=100)
    qs = ["what color is the cat", "how many dogs", "is it raining?", "where is the red ball"]
    tok.build_vocab(qs, min_freq=1)
    g = torch.Generator().manual_seed(1)
    items = []
    for i, q in enumerate(qs):
        ids, mask = tok.encode(q)
        items.append((torch.randint(0, 256, (64, 64, 3), generator=g, dtype=torch.uint8).numpy(), ids, mask, i))
    batch = P.gpu_collate_fn(items)
    assert set(batch) == {"images", "token_ids", "attention_mask", "answers"}          # vqa_collate_fn's keys (data/preprocess.py:305-315)
    assert batch["images"].dtype == torch.float32 and batch["images"].shape == (4, 3, 64, 64) and batch["images"].is_cuda
    assert batch["token_ids"].dtype == torch.int64 and batch["answers"].tolist() == [0, 1, 2, 3]
    ids_d, mask_d = tok.batch_encode_device(qs, DEV)
    assert torch.equal(ids_d, batch["token_ids"]) and torch.equal(mask_d, batch["attention_mask"])
    with torch.no_grad():
        logits, _ = m(batch["images"], batch["token_ids"], batch["attention_mask"])
        ref, _ = O.vqa_forward(IO.to_tensor_normalize(torch.stack([torch.as_tensor(it[0]) for it in items])), batch["token_ids"].cpu(),
                               batch["attention_mask"].cpu(), O.init_state_dict(cfg, 3), cfg, training=False)
    assert (logits.cpu() - ref).abs().max().item() < 1e-3
