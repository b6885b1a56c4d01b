"""GPU: the device half of the input pipeline through the C ABI.
  vqa_pack_tokens      == the REAL reference Tokenizer's batch_encode (golden fixture), bit-exact (index work)
  vqa_image_normalize  == ToTensor + Normalize restated with torch (oracle.input_oracle), bit-exact, at 224x224 and with flips
  vqa_image_resize     == PIL.Image.resize(BILINEAR) + ToTensor + Normalize, pinned by outputs of the REAL PIL (resize_pil.npz)
  vqa_image_color_jitter == PIL ImageEnhance blends + HSV hue shift (what ColorJitter runs on PIL images), pinned by the REAL PIL
                          (jitter_pil.npz) and, over all 2^24 colours, by the oracle that tests/test_input_cpu.py pins to PIL
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


# ---- Resize on the GPU: vqa_image_resize == PIL.Image.resize(BILINEAR), bit for bit ----
@pytest.fixture(scope="module")
def rgold(golden_dir):
    return np.load(os.path.join(golden_dir, "resize_pil.npz"))


@pytest.mark.parametrize("case", IO.RESIZE_CASES, ids=[c[0] for c in IO.RESIZE_CASES])
def test_device_resize_bit_exact_against_pil(rgold, case):
    """data/preprocess.py:70,90,118 / api/inference.py:140-170: Resize((S, S)) [+ RandomCrop window + flip] + ToTensor + Normalize.
    uint8 output == what the real PIL produced (golden), float output == ToTensor + Normalize of it, every value bit for bit."""
    tag, H, W, S, crop, (cy, cx), flip, seed, nb = case
    P = pkg().load_dropin_preprocess()
    img = IO.pattern_image(H, W, seed, nb)
    rz = P.DeviceImageResizer(size=S, crop=crop)
    fl = torch.tensor([bool(flip)]) if flip else None
    out, u8 = rz([img], crop_yx=[(cy, cx)] if crop else None, flip=fl, return_u8=True)
    torch.cuda.synchronize()
    assert np.array_equal(u8[0].cpu().numpy(), rgold[f"{tag}_u8"])
    ref = IO.to_tensor_normalize(torch.from_numpy(rgold[f"{tag}_u8"])[None])
    assert torch.equal(out.cpu(), ref)
    rows = out[0].cpu()[:, [0, out.shape[2] // 2, out.shape[2] - 1], :].numpy()
    assert np.array_equal(rows, rgold[f"{tag}_norm_rows"])


def test_device_resize_ragged_batch_matches_oracle_and_feeds_the_model():
    """A ragged batch (70 images of different sizes: three launch groups of 32, mixed up- / down-scaling, images that keep one
    axis) in ONE call against the numpy restatement of Pillow (itself pinned to PIL by the golden cases), then through
    gpu_collate_fn(resizer=...) into the model."""
    P = pkg().load_dropin_preprocess()
    rng = np.random.default_rng(3)
    sizes = [(int(rng.integers(20, 400)), int(rng.integers(20, 400))) for _ in range(66)] + [(224, 100), (90, 224), (224, 224), (1, 7)]
    imgs = [IO.pattern_image(h, w, 100 + i, 8 if i % 3 == 0 else 5) for i, (h, w) in enumerate(sizes)]
    flip = torch.tensor([i % 4 == 1 for i in range(len(imgs))])
    rz = P.DeviceImageResizer(size=224)
    out, u8 = rz(imgs, flip=flip, return_u8=True)
    torch.cuda.synchronize()
    assert out.shape == (70, 3, 224, 224) and u8.shape == (70, 224, 224, 3)
    got = u8.cpu().numpy()
    for i, im in enumerate(imgs):
        ref = IO.pil_resize_bilinear(im, 224, 224)
        if flip[i]:
            ref = ref[:, ::-1]
        assert np.array_equal(got[i], ref), (i, sizes[i])
    assert torch.equal(out.cpu(), IO.to_tensor_normalize(torch.from_numpy(got)))
    # augmented pipeline: Resize(256) -> RandomCrop(224) -> flip; origins drawn by the collate function, every value checked
    rz2 = P.DeviceImageResizer(size=256, crop=224)
    yx = [(int(rng.integers(0, 33)), int(rng.integers(0, 33))) for _ in imgs[:9]]
    out2, u82 = rz2(imgs[:9], crop_yx=yx, flip=flip[:9], return_u8=True)
    for i in range(9):
        ref = IO.pil_resize_bilinear(imgs[i], 256, 256)[yx[i][0]: yx[i][0] + 224, yx[i][1]: yx[i][1] + 224]
        if flip[i]:
            ref = ref[:, ::-1]
        assert np.array_equal(u82[i].cpu().numpy(), ref), i
    with pytest.raises(RuntimeError):
        rz2(imgs[:2])                                               # a crop needs its origins
    with pytest.raises(RuntimeError):
        rz([imgs[0].astype(np.float32)])
    with pytest.raises(RuntimeError):
        rz(imgs[:2], flip=torch.tensor([True]))                     # one flag per image
    # the collate route: decoded images of any size -> model-ready batch
    T = pkg().load_dropin_tokenizer()
    tok = T.Tokenizer(max_length=20, vocab_size=100)
    qs = ["what color is the cat", "how many dogs", "is it raining?"]
    tok.build_vocab(qs, min_freq=1)
    items = [(imgs[i],) + tuple(tok.encode(q)) + (i,) for i, q in enumerate(qs)]
    batch = P.gpu_collate_fn(items, resizer=rz)
    assert batch["images"].shape == (3, 3, 224, 224) and batch["images"].is_cuda
    ref3 = IO.to_tensor_normalize(torch.from_numpy(np.stack([IO.pil_resize_bilinear(imgs[i], 224, 224) for i in range(3)])))
    assert torch.equal(batch["images"].cpu(), ref3)


# ---- transforms.ColorJitter (data/preprocess.py:77-82) ---------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def jgold(golden_dir):
    return np.load(os.path.join(golden_dir, "jitter_pil.npz"))


def _factors(b, c, s, h):
    return torch.tensor([[float("nan") if v is None else v for v in (b, c, s, h)]], dtype=torch.float64)


@pytest.mark.parametrize("case", IO.JITTER_CASES, ids=[c[0] for c in IO.JITTER_CASES])
def test_device_color_jitter_bit_exact_against_pil(jgold, case):
    """uint8 output == what the real PIL's ImageEnhance / HSV round trip produced (golden), float output == ToTensor + Normalize of it."""
    tag, H, W, seed, nb, order, b, c, s, h = case
    P = pkg().load_dropin_preprocess()
    img = torch.from_numpy(IO.pattern_image(H, W, seed, nb))[None].to(DEV)
    cj = P.DeviceColorJitter(0.2, 0.2, 0.2, 0.1)
    out, u8 = cj(img, order=torch.tensor([order]), factors=_factors(b, c, s, h), return_u8=True)
    torch.cuda.synchronize()
    assert np.array_equal(u8[0].cpu().numpy(), jgold[f"{tag}_u8"])
    assert torch.equal(out.cpu(), IO.to_tensor_normalize(torch.from_numpy(jgold[f"{tag}_u8"])[None]))


@pytest.mark.parametrize("case", IO.PIPELINE_CASES, ids=[c[0] for c in IO.PIPELINE_CASES])
def test_device_training_transform_bit_exact_against_pil(jgold, case):
    """get_train_transforms(use_augmentation=True) (data/preprocess.py:66-84) in ONE resizer call against the same chain run by the real
    PIL: Resize(256) -> crop window -> flip -> ColorJitter (uint8, bit for bit) -> ToTensor -> Normalize."""
    tag, H, W, seed, nb, S, crop, (cy, cx), flip, order, b, c, s, h = case
    P = pkg().load_dropin_preprocess()
    rz = P.DeviceImageResizer(size=S, crop=crop, jitter=P.DeviceColorJitter(0.2, 0.2, 0.2, 0.1))
    out, u8 = rz([IO.pattern_image(H, W, seed, nb)], crop_yx=[(cy, cx)], flip=torch.tensor([bool(flip)]), return_u8=True,
                 jitter_params=(torch.tensor([order]), _factors(b, c, s, h)))
    torch.cuda.synchronize()
    assert np.array_equal(u8[0].cpu().numpy(), jgold[f"{tag}_u8"])
    assert torch.equal(out.cpu(), IO.to_tensor_normalize(torch.from_numpy(jgold[f"{tag}_u8"])[None]))


def test_device_hue_round_trip_on_every_colour():
    """RGB -> HSV -> (+delta) -> RGB for all 2^24 colours (a 4096 x 4096 image) at three shifts, and the saturation / brightness blends
    on the same image, against the numpy restatement that tests/test_input_cpu.py pins to the real PIL over the same full domain."""
    P = pkg().load_dropin_preprocess()
    allc = np.stack(np.meshgrid(np.arange(256), np.arange(256), np.arange(256), indexing="ij"), -1).reshape(1, 4096, 4096, 3).astype(np.uint8)
    x = torch.from_numpy(allc).to(DEV)
    cj = P.DeviceColorJitter(0.2, 0.2, 0.2, 0.5)
    for hue in (0.0, 0.1, -0.37):
        _, u8 = cj(x, order=torch.tensor([[3, 0, 1, 2]]), factors=_factors(None, None, None, hue), return_u8=True)
        assert np.array_equal(u8[0].cpu().numpy(), IO.color_jitter(allc[0], (3,), hue=hue)), hue
    for sat, br in ((0.8, 1.2), (1.37, 0.61)):
        _, u8 = cj(x, order=torch.tensor([[2, 0, 3, 1]]), factors=_factors(br, None, sat, None), return_u8=True)
        assert np.array_equal(u8[0].cpu().numpy(), IO.color_jitter(allc[0], (2, 0), brightness=br, saturation=sat)), (sat, br)


def test_device_color_jitter_batch_random_draws_and_training_pipeline():
    """A batch with per-image permutations and factors drawn by the class itself (every image against the oracle), then the whole
    training transform of data/preprocess.py:66-87 -- Resize(256) -> RandomCrop(224) -> flip -> ColorJitter -> ToTensor -> Normalize --
    in one resizer call on a ragged batch, and through gpu_collate_fn."""
    P = pkg().load_dropin_preprocess()
    g = torch.Generator().manual_seed(11)
    B, H, W = 37, 96, 120
    imgs = np.stack([IO.pattern_image(H, W, 300 + i, 8 if i % 2 else 5) for i in range(B)])
    cj = P.DeviceColorJitter(brightness=0.2, contrast=0.2, saturation=0.2, hue=0.1)
    order, factors = cj.draw(B, g)
    assert order.shape == (B, 4) and sorted(order[0].tolist()) == [0, 1, 2, 3] and len({tuple(o.tolist()) for o in order}) > 5
    assert float(factors[:, :3].min()) >= 0.8 and float(factors[:, :3].max()) <= 1.2 and float(factors[:, 3].abs().max()) <= 0.1
    out, u8 = cj(torch.from_numpy(imgs).to(DEV), order=order, factors=factors, return_u8=True)
    torch.cuda.synchronize()
    got = u8.cpu().numpy()
    for i in range(B):
        f = [float(v) for v in factors[i]]
        ref = IO.color_jitter(imgs[i], order[i].tolist(), *[np.float32(v) if k < 3 else v for k, v in enumerate(f)])
        assert np.array_equal(got[i], ref), (i, order[i].tolist(), f)
    assert torch.equal(out.cpu(), IO.to_tensor_normalize(torch.from_numpy(got)))
    # adjustments switched off: an identity that still normalises; contrast only: needs the image mean
    off = P.DeviceColorJitter()
    assert off.brightness is None and off.hue is None
    o2, u2 = off(torch.from_numpy(imgs[:3]).to(DEV), return_u8=True)
    assert np.array_equal(u2.cpu().numpy(), imgs[:3])
    with pytest.raises(ValueError):
        P.DeviceColorJitter(hue=0.6)
    with pytest.raises(ValueError):
        cj(torch.from_numpy(imgs[:1]).to(DEV), order=order[:1], factors=torch.tensor([[1.0, 1.0, 1.0, 0.7]]))
    with pytest.raises(RuntimeError):
        cj(torch.from_numpy(imgs[:2]).to(DEV), order=order[:1], factors=factors[:1])
    with pytest.raises(RuntimeError):
        cj(torch.from_numpy(imgs[:1]))                                  # GPU only
    # the whole augmented pipeline on a ragged batch
    rng = np.random.default_rng(9)
    raw = [IO.pattern_image(int(rng.integers(40, 300)), int(rng.integers(40, 300)), 500 + i, 5) for i in range(6)]
    yx = [(int(rng.integers(0, 33)), int(rng.integers(0, 33))) for _ in raw]
    flip = torch.tensor([i % 2 == 0 for i in range(6)])
    rz = P.DeviceImageResizer(size=256, crop=224, jitter=cj)
    order, factors = cj.draw(6, g)
    out, u8 = rz(raw, crop_yx=yx, flip=flip, return_u8=True, jitter_params=(order, factors))
    for i in range(6):
        ref = IO.pil_resize_bilinear(raw[i], 256, 256)[yx[i][0]: yx[i][0] + 224, yx[i][1]: yx[i][1] + 224]
        if flip[i]:
            ref = ref[:, ::-1]
        f = [float(v) for v in factors[i]]
        ref = IO.color_jitter(np.ascontiguousarray(ref), order[i].tolist(), *f)
        assert np.array_equal(u8[i].cpu().numpy(), ref), i
    assert torch.equal(out.cpu(), IO.to_tensor_normalize(u8.cpu()))
    items = [(raw[i], torch.zeros(20, dtype=torch.long), torch.ones(20, dtype=torch.long), i) for i in range(4)]
    batch = P.gpu_collate_fn(items, resizer=rz, flip_p=0.5, generator=torch.Generator().manual_seed(3))
    assert batch["images"].shape == (4, 3, 224, 224) and torch.isfinite(batch["images"]).all()
