"""GPU: the N>1 train step.  Two ranks share the one GPU of the test box and talk over gloo (RCCL needs one GPU per
rank); this exercises HipTrainer + GradBucketReducer end to end: bucket all-reduce issued from engine.backward's
segment callbacks on a side stream, 1/world folded into the AdamW kernel, replicas stay identical."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from _pkg import pkg
    from oracle import vqa_oracle as O
    P = pkg()
    torch.cuda.set_device(0)
    cfg = O.full_config(dropout=0.0, answer_dropout=0.0, embed_dim=64, vocab_size=200, num_answers=40)
    sd = O.init_state_dict(cfg, 5, jitter=True)
    M = P.load_dropin()

    def fresh():
        m = M.VQAModel(**cfg, compute_dtype="fp32")
        m.load_state_dict(sd)
        return m.to("cuda").train()

    images, ids, mask, answers = O.synthetic_batch(4, seed=300, image_size=64, seq_len=12, vocab=200, num_answers=40)
    sl = slice(2 * rank, 2 * rank + 2)                      # this rank's shard of the global batch
    shard = [t[sl].cuda() for t in (images, ids, mask, answers)]
    model = fresh()
    tr = P.trainer.HipTrainer(model, lr=1e-3)
    assert tr.world == 2
    before = model._flat.detach().clone()
    # the data-parallel backward must not join the weight-gradient stream into the data-gradient stream per stage (only the
    # reducer's communication stream waits for both): exactly ONE join, at the end of backward
    eng0 = tr.engine
    joins, orig_join = [], eng0._join_off_path
    eng0._join_off_path = lambda: (joins.append(1), orig_join())[1]
    tr.step(*shard)
    torch.cuda.synchronize()
    g_sum = tr.G.clone()
    n_joins = len(joins)
    issued_all = len(tr.buckets)
    # clip + AdamW on the REDUCED gradient: g = G_sum / world, global norm over g, first AdamW step (m = v = 0)
    gm = g_sum.double() / world
    norm = gm.norm()
    gc = gm * min(1.0, 1.0 / (float(norm) + 1e-6))
    expect = before.double() * (1 - 1e-3 * 0.01) - 1e-3 * gc / (gc.abs() + 1e-8)
    upd_err = (model._flat.detach().double() - expect).abs().max().item()
    norm_err = abs(float(tr.grad_norm()) - float(norm)) / float(norm)
    # reference: this rank's own gradient without any reduction, summed over ranks by a plain all_reduce
    m2 = fresh()
    eng = m2._ensure_engine()
    G2 = torch.zeros_like(m2._flat)
    logits, _, tape = eng.forward(shard[0], shard[1], shard[2].float(), True, False, need_tape=True)
    dl = torch.empty_like(logits)
    loss = torch.zeros(1, device="cuda")
    P._lib.call("vqa_cross_entropy", 0, logits.data_ptr(), shard[3].data_ptr(), loss.data_ptr(), dl.data_ptr(), None, 2, 40, 1.0, None, None)
    eng.backward(tape, dl, G2)
    torch.cuda.synchronize()
    ref = G2.cpu()
    dist.all_reduce(ref)
    err = (g_sum.cpu() - ref).abs().max().item() / ref.abs().max().item()
    # replicas must hold identical parameters after the step
    mine = model._flat.detach().cpu()
    other = mine.clone()
    dist.broadcast(other, src=0)
    same = torch.equal(mine, other)
    moved = not torch.equal(mine, m2._flat.detach().cpu())
    q.put((rank, err, same, moved, n_joins, upd_err, norm_err))
    dist.destroy_process_group()


def test_two_rank_train_step_over_gloo_on_one_gpu():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, same, moved, n_joins, upd_err, norm_err in res:
        assert err < 1e-5, (rank, err)        # fp32 sums of the same two gradients
        assert same and moved
        assert n_joins == 1, n_joins          # un-serialised: no per-stage join of the weight-gradient stream
        assert upd_err < 2e-6 and norm_err < 1e-5, (upd_err, norm_err)      # 1/world, clip and AdamW act on the reduced gradient
