"""CPU rehearsal of the N>1 path: two gloo ranks run the bucketed gradient all-reduce of trainer.GradBucketReducer over
the real bucket table, in backward-completion order, and must end with the mean-able sum on every rank."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from _pkg import sub


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import vqa_oracle as O
    LY, TR = sub("layout"), sub("trainer")
    ent = LY.build_entries(O.full_config(embed_dim=32, vocab_size=100, num_answers=10))
    n = LY.flat_size(ent)
    g = torch.Generator().manual_seed(100 + rank)
    G = torch.randn(n, generator=g)
    mine = G.clone()
    red = TR.GradBucketReducer(G, LY.bucket_ranges(ent))
    order = []
    for name, _, _ in red.buckets:            # the engine calls on_segment in exactly this order
        red.on_segment(name)
        order.append(name)
    scale = red.finish()
    others = [torch.randn(n, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
    expect = sum(others)
    ok = torch.allclose(G, expect, atol=1e-6) and abs(scale - 1.0 / world) < 1e-12 and not torch.equal(G, mine)
    q.put((rank, bool(ok), order))
    dist.destroy_process_group()


def test_bucketed_allreduce_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert res[0][2][0] == "answer_head" and res[0][2][-1] == "image_encoder.stem"


def _worker_force(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import vqa_oracle as O
    LY, TR = sub("layout"), sub("trainer")
    ent = LY.build_entries(O.full_config(embed_dim=32, vocab_size=100, num_answers=10))
    n = LY.flat_size(ent)
    G = torch.randn(n, generator=torch.Generator().manual_seed(7))
    mine = G.clone()
    idle = TR.GradBucketReducer(G, LY.bucket_ranges(ent))                  # one rank, not forced: inactive, nothing is issued
    idle.on_segment("answer_head")
    red = TR.GradBucketReducer(G, LY.bucket_ranges(ent), force=True)       # one rank, forced: every bucket goes through all_reduce
    bad = torch.tensor([3], dtype=torch.int32)
    red.reduce_aux(bad)
    for name, _, _ in red.buckets:
        red.on_segment(name)
    issued = list(red.issued)
    scale = red.finish()
    q.put((rank, (not idle.active) and idle.bytes_reduced == 0 and red.active and red.bytes_reduced == 4 * n + 4
           and torch.equal(G, mine) and int(bad) == 3 and scale == 1.0 and len(issued) == len(red.groups) == 4, issued))
    dist.destroy_process_group()


def test_forced_reducer_runs_every_bucket_in_a_world_of_one():
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_force, args=(0, 1, port, q))
    p.start()
    rank, ok, issued = q.get(timeout=120)
    p.join(timeout=60)
    assert p.exitcode == 0 and ok
    # adjacent segments travel as one message: head + fusion + text encoder first, stage 2 + stage 1 + stem last
    assert issued[0] == "answer_head+fusion+text_encoder" and issued[1] == "image_encoder.stage4"
    assert issued[-1] == "image_encoder.stage2+image_encoder.stage1+image_encoder.stem"


def test_forced_reducer_needs_a_process_group():
    import pytest
    LY, TR = sub("layout"), sub("trainer")
    from oracle import vqa_oracle as O
    ent = LY.build_entries(O.full_config(embed_dim=32, vocab_size=100, num_answers=10))
    with pytest.raises(RuntimeError):
        TR.GradBucketReducer(torch.zeros(LY.flat_size(ent)), LY.bucket_ranges(ent), force=True)
