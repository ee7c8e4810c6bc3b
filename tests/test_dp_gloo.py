"""CPU, world_size 2, gloo: the data-parallel exchange step (bucketed sum all-reduce of the gradient arena +
1/world scaling) gives every rank the average gradient, bucket by bucket, in backward-completion order."""
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
    for b in gradient_buckets(cfg):                      # same order as the overlapped step
        works += allreduce_ranges(grads, [b], None, async_op=True)
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


def test_bucketed_allreduce_world2():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world))
