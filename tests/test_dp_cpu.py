"""Data-parallel gradient averaging (macaw_llm_amd.dp.GradSync) over 2 processes with the gloo
backend on CPU: hooks fire from backward, large tensors go out per tensor, small ones coalesced,
result = mean over ranks; parameters without gradients are skipped."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from macaw_llm_amd.dp import GradSync, shard_batch
        torch.manual_seed(0)                        # identical replicas
        big = torch.nn.Parameter(torch.randn(64, 48))
        small = torch.nn.Parameter(torch.randn(7))
        small2 = torch.nn.Parameter(torch.randn(3, 5))
        unused = torch.nn.Parameter(torch.randn(4))
        frozen = torch.nn.Parameter(torch.randn(4), requires_grad=False)
        params = [big, small, small2, unused, frozen]
        sync = GradSync(params, small_threshold=100)
        lo, hi = shard_batch(8, rank, world)
        x = torch.arange(8, dtype=torch.float32)[lo:hi]      # this rank's shard of the batch
        loss = (big.sum() * x.sum()) + (small * (rank + 1)).sum() + (small2 ** 2).sum() * (rank + 2)
        loss.backward()
        sync.finish()
        exp_big = torch.full_like(big, torch.arange(8.).sum().item() / world)
        exp_small = torch.full_like(small, (1 + 2) / 2)
        exp_small2 = 2 * small2.detach() * ((2 + 3) / 2)
        ok = (torch.allclose(big.grad, exp_big) and torch.allclose(small.grad, exp_small)
              and torch.allclose(small2.grad, exp_small2) and unused.grad is None and frozen.grad is None)
        # a second step must work too (handles / coalescing state reset)
        for p in params:
            p.grad = None
        ((big * (rank + 1)).sum() + small.sum()).backward()
        sync.finish()
        ok = ok and torch.allclose(big.grad, torch.full_like(big, 1.5)) and torch.allclose(small.grad, torch.ones(7))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_gradsync_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def test_shard_batch():
    from macaw_llm_amd.dp import shard_batch
    assert [shard_batch(256, r, 8) for r in (0, 7)] == [(0, 32), (224, 256)]
    with pytest.raises(ValueError):
        shard_batch(10, 0, 4)


def _worker_overlapped(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from macaw_llm_amd.train import OverlappedStep

        class SGD:  # stands in for FusedAdamW (whose kernel needs the GPU)
            step_count = 0

            def step_param(self, p):
                p.data -= 0.5 * p.grad

        torch.manual_seed(0)
        big = torch.nn.Parameter(torch.randn(40, 30))
        small = torch.nn.Parameter(torch.randn(5))
        w0, s0 = big.detach().clone(), small.detach().clone()
        rt = OverlappedStep([big, small], SGD(), small_threshold=100)
        ok = True
        for it in range(2):
            rt.begin()
            ((big * (rank + 1)).sum() + (small * (2 * rank + 1)).sum()).backward()
            rt.finish()
            w0 = w0 - 0.5 * 1.5          # mean over ranks of (rank + 1)
            s0 = s0 - 0.5 * 2.0          # mean over ranks of (2 rank + 1)
            ok = ok and torch.allclose(big.data, w0) and torch.allclose(small.data, s0)
        ok = ok and rt.opt.step_count == 2
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_overlapped_step_gloo_world2():
    """train.OverlappedStep: replicas stay identical and equal to SGD on the rank-mean gradient"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_overlapped, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]
