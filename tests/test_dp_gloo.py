"""CPU, world_size 2 and 4, gloo: the data-parallel exchange step (bucketed sum all-reduce of the gradient arena +
1/world scaling) gives every rank the average gradient, bucket by bucket, in the order the staged backward hands
buckets over (layers N-1 .. 1, the embedding bucket, layer 0 -- trainer.staged_backward); plus the host logic of
fit()'s batch sharding."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import quadruplet_sentence_transformer_amd  # noqa: F401


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from quadruplet_sentence_transformer_amd.config import PRESETS, build_layout
    from quadruplet_sentence_transformer_amd.trainer import allreduce_ranges, gradient_buckets
    cfg = PRESETS["tiny-bert"]
    _, total = build_layout(cfg)
    g = torch.Generator().manual_seed(100 + rank)
    grads = torch.randn(total, generator=g)
    local = grads.clone()
    works = []
    bk = gradient_buckets(cfg)
    N = cfg.num_layers
    order = list(range(N - 1)) + [N, N - 1]              # staged_backward: layers N-1..1, embeddings, layer 0
    assert sorted(order) == list(range(N + 1))
    from quadruplet_sentence_transformer_amd.trainer import staged_reduce_order
    assert staged_reduce_order(bk, total, True) == [bk[k] for k in order] and staged_reduce_order(bk, total, False) == [(0, total)]
    if rank == world - 1:
        # a rank with an empty shard (fit(): fewer rows in the last batch than ranks) issues the whole sequence at once
        works += allreduce_ranges(grads, staged_reduce_order(bk, total, True), None, async_op=True)
    else:
        for k in order:
            works += allreduce_ranges(grads, [bk[k]], None, async_op=True)
    for w in works:
        w.wait()
    grads *= 1.0 / world                                  # qst_clip_adamw_step's grad_scale
    others = [torch.randn(total, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
    expect = sum(others) / world
    ok = torch.allclose(grads, expect, rtol=1e-6, atol=1e-6) and torch.equal(local, others[rank])
    # every rank must hold bit-identical averaged gradients (replicas stay in lock-step)
    gathered = [torch.zeros_like(grads) for _ in range(world)]
    dist.all_gather(gathered, grads)
    same = all(torch.equal(gathered[0], t) for t in gathered)
    ret[rank] = bool(ok and same)
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_buckets_tile_the_arena():
    from quadruplet_sentence_transformer_amd.config import PRESETS, build_layout
    from quadruplet_sentence_transformer_amd.trainer import gradient_buckets
    for name in ("tiny-bert", "tiny-mpnet", "all-MiniLM-L6-v2", "all-mpnet-base-v2", "bert-base-uncased"):
        cfg = PRESETS[name]
        segs, total = build_layout(cfg)
        bk = gradient_buckets(cfg)
        assert len(bk) == cfg.num_layers + 1
        spans = sorted(bk)
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))          # contiguous, disjoint, complete
        for k, (b, e) in enumerate(bk[:-1]):                                 # bucket k = layer N-1-k, whole segments
            names = {s.name.split(".")[1] for s in segs if b <= s.offset < e}
            assert names == {str(cfg.num_layers - 1 - k)}
        assert all(not s.name.startswith("layer.") for s in segs if s.offset < bk[-1][1])


def test_fit_batch_sharding_rows():
    from quadruplet_sentence_transformer_amd.sentence_transformer import _shard_batch
    B = 10
    feats = [{"input_ids": torch.arange(B * 3).view(B, 3) + 100 * c, "attention_mask": torch.ones(B, 3, dtype=torch.int64)}
             for c in range(4)]
    labels = torch.arange(B)
    seen = []
    for r in range(4):
        f, l, n_total, n_mine = _shard_batch(feats, labels, r, 4)
        assert n_total == B and n_mine == l.numel() == len(range(r, B, 4))
        for c in range(4):
            assert torch.equal(f[c]["input_ids"], feats[c]["input_ids"][r::4])
        seen += l.tolist()
    assert sorted(seen) == list(range(B))                                   # every row on exactly one rank
    f, l, _, n = _shard_batch(feats, labels[:2], 3, 4)                      # more ranks than rows: an empty shard
    assert n == 0 and l.numel() == 0


@pytest.mark.parametrize("world", [2, 4])
def test_bucketed_allreduce(world):
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world))
