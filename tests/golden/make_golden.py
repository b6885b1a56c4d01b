"""Generate tests/golden/*.npz by running the REAL reference (imported from /root/reference).

Run in the build container only (the reference never travels to the GPU box):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden.py

Weights come from oracle.vqa_oracle.init_state_dict(cfg, seed) (this repo's own deterministic
generator) and are loaded into the reference VQAModel with load_state_dict(strict=True); inputs
come from oracle.vqa_oracle.synthetic_batch.  Only tensors (inputs' seeds, outputs, gradient
summaries) are stored -- no reference source.  tests/test_oracle_golden.py regenerates the same
weights/inputs from the seeds and checks the CPU oracle against the stored reference outputs.
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from models.vqa_model import VQAModel  # noqa: E402  (the reference)
from oracle import vqa_oracle as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(8)


def checksum(sd):
    keys = sorted(sd)
    return np.array([[float(sd[k].double().sum()), float(sd[k].double().abs().sum())] for k in keys])


def fixed_mask(B, L, lens):
    lens = torch.tensor(lens)
    return (torch.arange(L)[None, :] < lens[:, None]).long()


def build(cfg, seed, jitter):
    sd = O.init_state_dict(cfg, seed, jitter=jitter)
    model = VQAModel(**cfg)
    missing = model.load_state_dict(sd, strict=True)
    assert list(model.state_dict().keys()) == [n for n, _, _ in O.param_shapes(cfg)], "state_dict order/layout"
    return sd, model


def grads_summary(model, names):
    gd = dict(model.named_parameters())
    norms = np.array([float(gd[n].grad.double().norm()) for n in names])
    heads = np.stack([np.pad(gd[n].grad.flatten()[:64].numpy(), (0, max(0, 64 - gd[n].numel()))) for n in names])
    return norms, heads


def gen_full_eval():
    cfg = O.full_config()
    sd, model = build(cfg, seed=1, jitter=True)
    model.eval()
    images, ids, _, _ = O.synthetic_batch(4, seed=11)
    mask = fixed_mask(4, 20, [20, 15, 7, 5])
    with torch.no_grad():
        logits, aux = model(images, ids, mask, return_aux=True)
    top2 = logits.topk(2, dim=-1).values
    np.savez_compressed(
        os.path.join(OUT, "full_eval.npz"), weight_checksum=checksum(sd), logits=logits.numpy(),
        margin=(top2[:, 0] - top2[:, 1]).numpy(), fused=aux["fused"].numpy(),
        text_pooled=aux["text_pooled"].numpy(), attended_pooled=aux["attended_pooled"].numpy(),
        text_features=aux["text_features"].numpy(), image_projected=aux["image_projected"].numpy(),
        cross_w0=aux["cross_attention_weights"][0].numpy(), cross_w1=aux["cross_attention_weights"][1].numpy(),
        image_features=aux["image_features"].numpy())
    print("full_eval margins", (top2[:, 0] - top2[:, 1]).tolist())
    # all-padding row -> NaN logits for that sample (text_encoder.py:244 -inf masking)
    mask2 = mask.clone()
    mask2[2] = 0
    with torch.no_grad():
        l2, _ = model(images, ids, mask2)
    np.savez_compressed(os.path.join(OUT, "full_eval_allpad.npz"), logits=l2.numpy())
    print("allpad nan rows", torch.isnan(l2).any(dim=1).tolist())


def gen_full_train(tag, cfg, seed, B, image_size=224, seq_len=20, vocab=1000):
    sd, model = build(cfg, seed=seed, jitter=True)
    names = O.parameter_names(cfg)
    model.train()
    images, ids, mask, answers = O.synthetic_batch(B, seed=seed + 100, image_size=image_size, seq_len=seq_len,
                                                   vocab=vocab, num_answers=cfg["num_answers"])
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=0.01, betas=(0.9, 0.999))
    opt.zero_grad()
    logits, aux = model(images, ids, mask, return_aux=True)
    loss = torch.nn.CrossEntropyLoss()(logits, answers)
    loss.backward()
    norms, heads = grads_summary(model, names)
    gnorm = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    opt.step()
    delta = np.array([float((p.detach() - before[n]).double().norm()) for n, p in model.named_parameters()])
    st = model.state_dict()
    bn_keys = [k for k in st if "running_" in k]
    np.savez_compressed(
        os.path.join(OUT, f"{tag}.npz"), weight_checksum=checksum(sd), logits=logits.detach().numpy(),
        loss=loss.item(), gnorm=float(gnorm), grad_norms=norms, grad_heads=heads, step_delta_norms=delta,
        fused=aux["fused"].detach().numpy(), image_features=aux["image_features"].detach().numpy(),
        bn_running=np.concatenate([st[k].numpy() for k in bn_keys]),
        nbt=int(st["image_encoder.stem.1.num_batches_tracked"]))
    print(tag, "loss", loss.item(), "gnorm", float(gnorm))


def gen_overfit():
    """reproduce_issue.py:16-76 behaviour: the small model overfits one batch (acc > 0.9 after 50 steps)."""
    cfg = O.full_config(vocab_size=100, num_answers=10, embed_dim=32)
    sd, model = build(cfg, seed=42, jitter=False)
    g = torch.Generator().manual_seed(42)
    images = torch.randn(4, 3, 224, 224, generator=g)
    ids = torch.randint(0, 100, (4, 10), generator=g)
    mask = torch.ones(4, 10)
    targets = torch.tensor([1] * 4)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    model.train()
    losses = []
    for _ in range(50):
        opt.zero_grad()
        logits, _ = model(images, ids, mask)
        loss = torch.nn.functional.cross_entropy(logits, targets)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    acc = (logits.argmax(-1) == targets).float().mean().item()
    np.savez_compressed(os.path.join(OUT, "overfit.npz"), losses=np.array(losses), acc=acc)
    print("overfit acc", acc, "final loss", losses[-1])


def gen_metrics():
    """utils/metrics.py VQAAccuracy on seeded logits: several batches, well-separated values (no ties) plus one batch of
    duplicated maxima away from the target (so tie ORDER, which torch.topk leaves unspecified, cannot matter)."""
    from utils.metrics import VQAAccuracy  # the reference
    g = torch.Generator().manual_seed(11)
    acc = VQAAccuracy()
    per_batch = []
    batches = []
    for B, C in ((64, 1000), (7, 10), (33, 1000), (5, 6)):
        logits = torch.randn(B, C, generator=g)
        targets = torch.randint(0, C, (B,), generator=g)
        # make ~1/3 of the rows top-1 hits and ~1/3 rank 1..4
        for b in range(B):
            if b % 3 == 0:
                logits[b, targets[b]] = logits[b].max() + 1.0
            elif b % 3 == 1:
                v, _ = logits[b].sort(descending=True)
                logits[b, targets[b]] = (v[2] + v[3]) / 2 if C > 4 else v[1] - 1e-3
        acc.update(logits, targets)
        per_batch.append([acc.correct, acc.correct_top5, acc.total])
        batches.append((logits.numpy(), targets.numpy()))
    out = {"running": np.array(per_batch, dtype=np.int64)}
    for i, (l, t) in enumerate(batches):
        out[f"logits{i}"] = l; out[f"targets{i}"] = t
    m = acc.compute()
    out["accuracy"] = np.array([m["accuracy"], m["accuracy_top5"]])
    np.savez_compressed(os.path.join(OUT, "metrics.npz"), **out)
    print("metrics", per_batch, m["accuracy"], m["accuracy_top5"])


QUESTIONS = [
    "What color is the cat?", "How many people are there?", "Is this a beach?", "What is the man doing?",
    "What's in the background?", "what   is  the  WOMAN holding -- an umbrella, or a (red) bag?!", "", "?!...",
    "Is the dog's tail up", "Don't the two giraffes look like they're eating leaves from the very tall tree behind the fence",
    "one two three four five six seven eight nine ten eleven twelve thirteen fourteen fifteen sixteen seventeen eighteen",
    "a b c d e f g h i j k l m n o p q r", "a b c d e f g h i j k l m n o p q r s", "a b c d e f g h i j k l m n o p q r s t u v",
    "Is it snowing_outside at 5pm?", "caf\u00e9 na\u00efve \u00fcber stra\u00dfe", "What\tcolor\nis   the\r\nkite", "zebra xylophone quux",
]


def gen_input():
    """utils/tokenizer.py Tokenizer (the real reference class) on a fixed question list: vocabulary, ids, masks for two
    (max_length, add_special_tokens) settings.  Image half: torchvision is absent here, so ToTensor + Normalize are restated
    with torch only (oracle.input_oracle.to_tensor_normalize) -- the fixture pins the kernel to that restatement, not to torchvision."""
    import json
    from utils.tokenizer import Tokenizer  # the reference
    from oracle import input_oracle as IO
    out = {}
    for tag, max_len in (("l20", 20), ("l8", 8)):
        tok = Tokenizer(max_length=max_len, vocab_size=40)
        tok.build_vocab(QUESTIONS[:14], min_freq=1)
        ids, mask = tok.batch_encode(QUESTIONS)
        ids_ns, mask_ns = tok.batch_encode(QUESTIONS, add_special_tokens=False)
        out[f"ids_{tag}"] = np.array(ids, dtype=np.int64); out[f"mask_{tag}"] = np.array(mask, dtype=np.int64)
        out[f"ids_ns_{tag}"] = np.array(ids_ns, dtype=np.int64); out[f"mask_ns_{tag}"] = np.array(mask_ns, dtype=np.int64)
        out[f"vocab_{tag}"] = np.array(json.dumps(tok.word2idx, ensure_ascii=True))
        out[f"decoded_{tag}"] = np.array(json.dumps([tok.decode(r) for r in ids], ensure_ascii=True))
    out["questions"] = np.array(json.dumps(QUESTIONS, ensure_ascii=True))
    g = torch.Generator().manual_seed(2024)
    img = torch.randint(0, 256, (3, 16, 24, 3), generator=g, dtype=torch.uint8)
    img[0, 0, 0] = torch.tensor([0, 255, 128], dtype=torch.uint8)
    flip = torch.tensor([0, 1, 0], dtype=torch.uint8)
    out["img_u8"] = img.numpy()
    out["img_flip"] = flip.numpy()
    out["img_norm"] = IO.to_tensor_normalize(img).numpy()
    out["img_norm_flip"] = IO.to_tensor_normalize(img, flip).numpy()
    np.savez_compressed(os.path.join(OUT, "input_pipeline.npz"), **out)
    print("input_pipeline", out["ids_l20"].shape, int(out["mask_l20"].sum()), "vocab", len(json.loads(str(out["vocab_l20"]))))


def gen_resize():
    """transforms.Resize((S, S)) on a PIL image is PIL.Image.resize((S, S), BILINEAR) (torchvision's functional resize calls exactly
    that for PIL inputs; torchvision itself is absent here, PIL 12.2.0 is present): data/preprocess.py:70,90,118, api/inference.py:
    140-170.  Outputs of the REAL PIL on integer-pattern images (oracle.input_oracle.pattern_image: rebuilt, not stored), the
    RandomCrop window / horizontal flip of the augmented pipeline as plain indexing of PIL's output, and ToTensor + Normalize of the
    actual PIL image via np.array (torchvision's ToTensor: from_numpy(np.array(pic)) -> permute -> float / 255; Normalize: sub_ mean, div_ std)."""
    import hashlib
    from PIL import Image
    from oracle import input_oracle as IO
    out = {"pil_version": np.array(__import__("PIL").__version__)}
    mean = torch.as_tensor(IO.IMAGENET_MEAN, dtype=torch.float32)[:, None, None]
    std = torch.as_tensor(IO.IMAGENET_STD, dtype=torch.float32)[:, None, None]
    for tag, H, W, S, crop, (cy, cx), flip, seed, nb in IO.RESIZE_CASES:
        img = IO.pattern_image(H, W, seed, nb)
        pil = Image.fromarray(img, mode="RGB").resize((S, S), Image.BILINEAR)
        if crop:
            pil = pil.crop((cx, cy, cx + crop, cy + crop))                       # RandomCrop: F.crop -> img.crop((left, top, right, bottom))
        if flip:
            pil = pil.transpose(Image.FLIP_LEFT_RIGHT)                           # RandomHorizontalFlip: F.hflip
        arr = np.array(pil, copy=True)                                           # ToTensor starts here
        t = torch.from_numpy(arr).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
        t = t.sub_(mean).div_(std)                                               # Normalize
        out[f"{tag}_u8"] = arr
        out[f"{tag}_sha"] = np.array(hashlib.sha256(img.tobytes()).hexdigest())
        # the float tensor is a pure function of arr; keep a thin slice of it (rows 0, 111, last) instead of 600 KB per case
        out[f"{tag}_norm_rows"] = t[:, [0, t.shape[1] // 2, t.shape[1] - 1], :].numpy()
    np.savez_compressed(os.path.join(OUT, "resize_pil.npz"), **out)
    print("resize_pil", len(IO.RESIZE_CASES), "cases", os.path.getsize(os.path.join(OUT, "resize_pil.npz")), "bytes")


def pil_color_jitter(img, order, brightness, contrast, saturation, hue):
    """torchvision's ColorJitter.forward on a PIL image, spelled with the PIL calls its functional_pil ops make (torchvision is absent)."""
    from PIL import Image, ImageEnhance
    im = Image.fromarray(img, mode="RGB")
    for fn in order:
        if fn == 0 and brightness is not None:
            im = ImageEnhance.Brightness(im).enhance(brightness)                  # F.adjust_brightness
        elif fn == 1 and contrast is not None:
            im = ImageEnhance.Contrast(im).enhance(contrast)                      # F.adjust_contrast
        elif fn == 2 and saturation is not None:
            im = ImageEnhance.Color(im).enhance(saturation)                       # F.adjust_saturation
        elif fn == 3 and hue is not None:                                         # F.adjust_hue
            h, s_, v = im.convert("HSV").split()
            np_h = np.array(h, dtype=np.uint8)
            with np.errstate(over="ignore", invalid="ignore"):
                np_h += np.array(hue * 255).astype(np.uint8)                      # uint8 addition wraps across the hue circle
            im = Image.merge("HSV", (Image.fromarray(np_h, "L"), s_, v)).convert("RGB")
    return np.array(im, copy=True)


def gen_jitter():
    """transforms.ColorJitter (data/preprocess.py:77-82) on PIL images = PIL's ImageEnhance / HSV conversion: outputs of the REAL PIL
    for fixed permutations and factors on rebuilt integer-pattern images (oracle.input_oracle.JITTER_CASES)."""
    import warnings
    from oracle import input_oracle as IO
    out = {"pil_version": np.array(__import__("PIL").__version__)}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for tag, H, W, seed, nb, order, b, c, s_, h in IO.JITTER_CASES:
            out[f"{tag}_u8"] = pil_color_jitter(IO.pattern_image(H, W, seed, nb), order, b, c, s_, h)
        from PIL import Image
        for tag, H, W, seed, nb, S, crop, (cy, cx), flip, order, b, c, s_, h in IO.PIPELINE_CASES:     # the whole training transform
            pil = Image.fromarray(IO.pattern_image(H, W, seed, nb), mode="RGB").resize((S, S), Image.BILINEAR)
            pil = pil.crop((cx, cy, cx + crop, cy + crop))
            if flip:
                pil = pil.transpose(Image.FLIP_LEFT_RIGHT)
            out[f"{tag}_u8"] = pil_color_jitter(np.array(pil, copy=True), order, b, c, s_, h)
    np.savez_compressed(os.path.join(OUT, "jitter_pil.npz"), **out)
    print("jitter_pil", len(IO.JITTER_CASES), "cases", os.path.getsize(os.path.join(OUT, "jitter_pil.npz")), "bytes")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "jitter":
        gen_jitter(); sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "resize":
        gen_resize(); sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "metrics":
        gen_metrics(); sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "input":
        gen_input(); sys.exit(0)
    gen_full_eval()
    gen_full_train("full_train", O.full_config(dropout=0.0, answer_dropout=0.0), seed=2, B=4)
    gen_full_train("small_train", O.full_config(dropout=0.0, answer_dropout=0.0, vocab_size=100, num_answers=10,
                                                embed_dim=32), seed=3, B=2, image_size=64, seq_len=10, vocab=100)
    gen_overfit()
    gen_metrics()
    gen_input()
    gen_resize()
    gen_jitter()
