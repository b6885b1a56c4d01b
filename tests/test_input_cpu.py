"""CPU: the input-side oracle (oracle/input_oracle.py) and the host half of the drop-in tokenizer against the fixture written
by the REAL reference `utils.tokenizer.Tokenizer` (tests/golden/input_pipeline.npz, make_golden.py gen_input): ids, masks,
decode, with and without special tokens, at max_length 20 and 8 (truncation with END forced onto the last slot), including
empty / punctuation-only questions, apostrophes, underscores, unicode words, tabs and newlines, unknown words."""
import json
import os

import numpy as np
import pytest

from _pkg import pkg
from oracle import input_oracle as IO


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "input_pipeline.npz"))


@pytest.mark.parametrize("tag,L", [("l20", 20), ("l8", 8)])
def test_oracle_encode_matches_reference_tokenizer(gold, tag, L):
    qs = json.loads(str(gold["questions"]))
    vocab = json.loads(str(gold[f"vocab_{tag}"]))
    ids, mask = IO.batch_encode(qs, vocab, L)
    assert np.array_equal(ids, gold[f"ids_{tag}"]) and np.array_equal(mask, gold[f"mask_{tag}"])
    ids, mask = IO.batch_encode(qs, vocab, L, add_special_tokens=False)
    assert np.array_equal(ids, gold[f"ids_ns_{tag}"]) and np.array_equal(mask, gold[f"mask_ns_{tag}"])


@pytest.mark.parametrize("tag,L", [("l20", 20), ("l8", 8)])
def test_dropin_tokenizer_host_half_matches_reference(gold, tag, L, tmp_path):
    T = pkg().load_dropin_tokenizer()
    qs = json.loads(str(gold["questions"]))
    tok = T.Tokenizer(max_length=L, vocab_size=40)
    tok.build_vocab(qs[:14], min_freq=1)                       # same corpus / limits as the fixture: the vocabulary itself must agree
    assert tok.word2idx == json.loads(str(gold[f"vocab_{tag}"]))
    ids, mask = tok.batch_encode(qs)
    assert np.array_equal(np.array(ids), gold[f"ids_{tag}"]) and np.array_equal(np.array(mask), gold[f"mask_{tag}"])
    ids, mask = tok.batch_encode(qs, add_special_tokens=False)
    assert np.array_equal(np.array(ids), gold[f"ids_ns_{tag}"]) and np.array_equal(np.array(mask), gold[f"mask_ns_{tag}"])
    assert [tok.decode(r) for r in gold[f"ids_{tag}"]] == json.loads(str(gold[f"decoded_{tag}"]))
    # vocabulary file format round trip (utils/tokenizer.py:274-310)
    path = str(tmp_path / "vocab.json")
    tok.save(path)
    t2 = T.Tokenizer()
    t2.load(path)
    assert t2.word2idx == tok.word2idx and t2.max_length == L and t2.encode(qs[5]) == tok.encode(qs[5])
    # no silent CPU path for the device packer
    with pytest.raises(RuntimeError):
        tok.batch_encode_device(qs, device="cpu")


def test_image_oracle_matches_fixture_and_formula(gold):
    import torch
    img = torch.from_numpy(gold["img_u8"])
    out = IO.to_tensor_normalize(img)
    assert np.array_equal(out.numpy(), gold["img_norm"])
    assert np.array_equal(IO.to_tensor_normalize(img, gold["img_flip"]).numpy(), gold["img_norm_flip"])
    # spot values: (u/255 - mean) / std
    assert abs(float(out[0, 0, 0, 0]) - (0.0 - 0.485) / 0.229) < 1e-6
    assert abs(float(out[0, 1, 0, 0]) - (1.0 - 0.456) / 0.224) < 1e-6


# ---- image half: Resize (PIL BILINEAR) + ToTensor + Normalize, pinned by the REAL PIL (tests/golden/resize_pil.npz) ----
import hashlib  # noqa: E402

import torch  # noqa: E402


@pytest.fixture(scope="module")
def rgold(golden_dir):
    return np.load(os.path.join(golden_dir, "resize_pil.npz"))


def _expected_from_oracle(case):
    tag, H, W, S, crop, (cy, cx), flip, seed, nb = case
    img = IO.pattern_image(H, W, seed, nb)
    out = IO.pil_resize_bilinear(img, S, S)
    if crop:
        out = out[cy: cy + crop, cx: cx + crop]
    if flip:
        out = out[:, ::-1]
    return img, np.ascontiguousarray(out)


@pytest.mark.parametrize("case", IO.RESIZE_CASES, ids=[c[0] for c in IO.RESIZE_CASES])
def test_oracle_resize_is_bit_exact_against_pil(rgold, case):
    """oracle.input_oracle.pil_resize_bilinear (numpy restatement of Pillow's Resample.c) == the uint8 image the real
    PIL.Image.resize(BILINEAR) produced (+ crop / flip as indexing); ToTensor + Normalize restated == the rows stored from the
    real np.array(pil) -> torch pipeline.  Down-, up-scaling, one axis only, identity, 1x1, extreme aspect ratio, pure noise."""
    tag = case[0]
    img, got = _expected_from_oracle(case)
    assert hashlib.sha256(img.tobytes()).hexdigest() == str(rgold[f"{tag}_sha"])      # the rebuilt input IS the fixture's input
    assert np.array_equal(got, rgold[f"{tag}_u8"])
    t = IO.to_tensor_normalize(torch.from_numpy(got)[None])[0]
    rows = t[:, [0, t.shape[1] // 2, t.shape[1] - 1], :].numpy()
    assert np.array_equal(rows, rgold[f"{tag}_norm_rows"])


def test_pil_coefficients_known_answers():
    """precompute_coeffs / normalize_coeffs_8bpc known answers: 2x down-scale = the 4-tap triangle {1,3,3,1}/8 in 22-bit fixed
    point, clipped windows at the borders renormalise to sum 1.0, identity scale degenerates to a single unit tap."""
    xmin, xn, kk = IO.pil_bilinear_coeffs(8, 4)
    one = 1 << IO.PRECISION_BITS
    assert kk.shape[1] == 5 and list(xmin) == [0, 1, 3, 5] and list(xn) == [3, 4, 4, 3]
    assert list(kk[1, :4]) == [one // 8, 3 * one // 8, 3 * one // 8, one // 8]
    assert abs(int(kk[0].sum()) - one) <= 2 and abs(int(kk[3].sum()) - one) <= 2
    xmin, xn, kk = IO.pil_bilinear_coeffs(5, 5)
    assert all(int(kk[i, : xn[i]].max()) == one and int(kk[i].sum()) == one for i in range(5))


# ---- transforms.ColorJitter = PIL ImageEnhance + HSV conversion (data/preprocess.py:77-82): oracle pinned by the real PIL ----------------
@pytest.fixture(scope="module")
def jgold(golden_dir):
    return np.load(os.path.join(golden_dir, "jitter_pil.npz"))


@pytest.mark.parametrize("case", IO.JITTER_CASES, ids=[c[0] for c in IO.JITTER_CASES])
def test_oracle_color_jitter_is_bit_exact_against_pil(jgold, case):
    tag, H, W, seed, nb, order, b, c, s, h = case
    got = IO.color_jitter(IO.pattern_image(H, W, seed, nb), order, b, c, s, h)
    assert got.dtype == np.uint8 and np.array_equal(got, jgold[f"{tag}_u8"])


@pytest.mark.parametrize("case", IO.PIPELINE_CASES, ids=[c[0] for c in IO.PIPELINE_CASES])
def test_oracle_training_transform_matches_pil_end_to_end(jgold, case):
    """Resize(256) -> RandomCrop(224) window -> flip -> ColorJitter, every step run by the real PIL (golden) vs the chained restatements."""
    tag, H, W, seed, nb, S, crop, (cy, cx), flip, order, b, c, s, h = case
    x = IO.pil_resize_bilinear(IO.pattern_image(H, W, seed, nb), S, S)[cy: cy + crop, cx: cx + crop]
    if flip:
        x = x[:, ::-1]
    assert np.array_equal(IO.color_jitter(np.ascontiguousarray(x), order, b, c, s, h), jgold[f"{tag}_u8"])


def _all_colours():
    return np.stack(np.meshgrid(np.arange(256), np.arange(256), np.arange(256), indexing="ij"), -1).reshape(4096, 4096, 3).astype(np.uint8)


def test_oracle_colour_conversions_match_pil_on_every_input():
    """RGB -> L, RGB -> HSV and HSV -> RGB of the installed PIL over all 2^24 triples (an exhaustive pin: the domain is finite), and
    Image.blend over every (degenerate, image) byte pair for factors inside, outside and at the ends of [0, 1]."""
    Image = pytest.importorskip("PIL.Image")
    allc = _all_colours()
    assert np.array_equal(IO.pil_rgb_to_l(allc), np.asarray(Image.fromarray(allc, "RGB").convert("L")))
    assert np.array_equal(IO.pil_rgb_to_hsv(allc), np.asarray(Image.fromarray(allc, "RGB").convert("HSV")))
    assert np.array_equal(IO.pil_hsv_to_rgb(allc), np.asarray(Image.fromarray(allc, "HSV").convert("RGB")))
    a, b = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    ia, ib = Image.fromarray(a, "L"), Image.fromarray(b, "L")
    rng = np.random.default_rng(5)
    for f in [0.0, 1.0, 0.8, 1.2, 0.5, 1.9999, 2.5, -0.3, 1e-3] + list(rng.uniform(0.0, 2.0, 40)):
        assert np.array_equal(IO.pil_blend(a, b, f), np.asarray(Image.blend(ia, ib, f))), f


def test_hue_delta_wraps_like_a_uint8_addition():
    assert IO.hue_delta(0.1) == 25 and IO.hue_delta(-0.1) == 231 and IO.hue_delta(0.5) == 127 and IO.hue_delta(-0.5) == 129
    assert IO.hue_delta(0.0) == 0 and IO.hue_delta(-0.001) == 0
    with pytest.raises(ValueError):
        IO.color_jitter(np.zeros((2, 2, 3), np.uint8), (3,), hue=0.6)
