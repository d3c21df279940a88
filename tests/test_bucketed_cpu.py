"""macaw_llm_amd.bucketed.BucketedStep over gloo on CPU, world 2 and world 8: flat buckets,
reduce-scatter -> owner updates its slice -> in-place all-gather, parameters whose sizes are not
multiples of 8 * world, fused runs kept adjacent, parameters without gradient, gradient
accumulation, global-norm clipping.  The optimizer is a stand-in with FusedAdamW's interface
(the real kernel needs the GPU: tests/test_train_gpu.py)."""
import math
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


class ShardSGD:
    """FusedAdamW's interface (step_count, lr, step_shard) with plain SGD arithmetic"""

    def __init__(self, lr=0.5):
        self.step_count = 0
        self.lr = lr
        self.calls = 0

    def step_shard(self, key, w, g, grad_scale=1.0):
        self.calls += 1
        idx, lo, n = key
        assert w.numel() == g.numel() == n
        w -= self.lr * grad_scale * g


class FusedFn(torch.autograd.Function):
    """two parameters that are row slices of ONE buffer, gradients returned as row slices of one
    buffer too (the fused q|k|v layout of the engine)"""

    @staticmethod
    def forward(ctx, a, b, s):
        ctx.s = s
        return (a.sum() + 2 * b.sum()) * s

    @staticmethod
    def backward(ctx, dy):
        gfull = torch.empty(48, 16)
        gfull[:16] = ctx.s
        gfull[16:] = 2 * ctx.s
        gfull *= (torch.arange(768.).view(48, 16) / 768 + 1) * dy
        return gfull[:16], gfull[16:], None


def _make_params():
    torch.manual_seed(0)
    big = torch.nn.Parameter(torch.randn(40, 32))
    odd = torch.nn.Parameter(torch.randn(41, 31))       # 1271 elements: not a multiple of 8 * world
    small = torch.nn.Parameter(torch.randn(5))
    fused = torch.randn(48, 16)
    # registered in the order (fb, fa): adjacency must be found by address, not by order
    fa, fb = torch.nn.Parameter(torch.empty(0)), torch.nn.Parameter(torch.empty(0))
    fa.data, fb.data = fused[:16], fused[16:]
    unused = torch.nn.Parameter(torch.randn(9, 7))      # never receives a gradient
    lonely = torch.nn.Parameter(torch.randn(600))       # own bucket, never receives a gradient
    return big, odd, small, fa, fb, unused, lonely


def _loss(ps, rank, micro=0, drop_small=False):
    big, odd, small, fa, fb, unused, lonely = ps
    scale = torch.arange(1280.).view(40, 32) / 1280
    m = micro + 1
    out = (FusedFn.apply(fa, fb, float(rank + 1) * m) + (big * scale * (rank + 1) * m).sum()
           + (odd * (rank + 3)).sum() * m)
    if not drop_small:        # drop_small: this rank's batch lacks the "modality" that feeds `small`
        out = out + (small * (2 * rank + 1)).sum() * m
    return out


def _expected_grads(world, micros, small_ranks=None):
    """rank-mean gradients summed over the micro-batches (small_ranks: the ranks whose batch feeds
    `small`; the others contribute zeros to its mean)"""
    mr = sum(r + 1 for r in range(world)) / world
    mo = sum(r + 3 for r in range(world)) / world
    ms = sum(2 * r + 1 for r in (range(world) if small_ranks is None else small_ranks)) / world
    mm = sum(m + 1 for m in range(micros))
    scale = torch.arange(1280.).view(40, 32) / 1280
    fscale = torch.cat([torch.ones(16, 16), 2 * torch.ones(32, 16)]) * (torch.arange(768.).view(48, 16) / 768 + 1)
    return dict(big=mr * mm * scale, odd=torch.full((41, 31), mo * mm), small=torch.full((5,), ms * mm),
                fused=mr * mm * fscale)


def _worker(rank, world, port, q, mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from macaw_llm_amd.bucketed import BucketedStep
        ps = _make_params()
        big, odd, small, fa, fb, unused, lonely = ps
        ref = dict(big=big.detach().clone(), odd=odd.detach().clone(), small=small.detach().clone(),
                   fused=torch.cat([fa.detach(), fb.detach()]).clone(), unused=unused.detach().clone(),
                   lonely=lonely.detach().clone())
        micros = 2 if mode in ("accumulate", "accumulate_mean") else 1
        clip = 3.0 if mode == "clip" else None
        opt = ShardSGD()
        rt = BucketedStep([big, odd, small, fb, fa, unused, lonely], opt, bucket_bytes=6000,
                          accumulate_steps=micros, max_grad_norm=clip, zero1=(mode != "allreduce"),
                          average_accumulated=(mode == "accumulate_mean"))
        ok = len(rt.buckets) >= 3
        # fused run kept gap-free and in address order inside its bucket
        ok = ok and fb.data.data_ptr() == fa.data.data_ptr() + fa.numel() * 4
        for b in rt.buckets:
            ok = ok and b.n % (8 * world) == 0
        for it in range(3 if mode == "ragged" else 2):
            # "ragged": from the second step on, the odd ranks' batches do not feed `small` -- its bucket
            # completes during the backward on some ranks and only in finish() on the others; the
            # collectives must still be issued in ONE order everywhere (rank-invariant schedule)
            ragged = mode == "ragged" and it >= 1
            eg = _expected_grads(world, micros, [r for r in range(world) if r % 2 == 0] if ragged else None)
            for m in range(micros):
                rt.begin()
                _loss(ps, rank, m, drop_small=ragged and rank % 2 == 1).backward()
                rt.finish()
            sc = 1.0 / micros if mode == "accumulate_mean" else 1.0
            if clip is not None:
                nrm = math.sqrt(sum(float((g ** 2).sum()) for g in eg.values()))
                sc = sc * min(1.0, clip / (nrm + 1e-6))
                ok = ok and abs(float(rt.grad_norm) - nrm) <= 1e-4 * nrm
            for k in ("big", "odd", "small", "fused"):
                ref[k] = ref[k] - 0.5 * sc * eg[k]
            got = dict(big=big.data, odd=odd.data, small=small.data, fused=torch.cat([fa.data, fb.data]))
            for k in got:
                if not torch.allclose(got[k], ref[k], atol=1e-5, rtol=1e-5):
                    ok = False
                    print("DEBUG", mode, world, rank, it, k, (got[k] - ref[k]).abs().max().item(), flush=True)
            # parameters without gradient: untouched (zero gradient in a used bucket; unused bucket skipped)
            ok = ok and torch.equal(unused.data, ref["unused"]) and torch.equal(lonely.data, ref["lonely"])
        ok = ok and opt.step_count == (3 if mode == "ragged" else 2)
        ok = ok and rt._order is not None and sorted(rt._order) == list(range(len(rt.buckets)))
        # replicas identical (the all-gather really distributed the other ranks' slices)
        flat = torch.cat([b.w for b in rt.buckets])
        others = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(others, flat)
        ok = ok and all(torch.equal(o, others[0]) for o in others)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["plain", "accumulate", "accumulate_mean", "clip", "ragged", "allreduce"])
@pytest.mark.parametrize("world", [2, 8])
def test_bucketed_step_gloo(world, mode):
    if world == 8 and mode not in ("plain", "ragged") and (os.cpu_count() or 1) < 4:
        pytest.skip("too few cores for 8 ranks x 3 modes")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(res) == [(r, True) for r in range(world)]


def test_single_process_matches_plain_sgd_and_schedule():
    from macaw_llm_amd.bucketed import BucketedStep, cosine_with_warmup
    ps = _make_params()
    big, odd, small, fa, fb, unused, lonely = ps
    w0 = big.detach().clone()
    opt = ShardSGD()
    rt = BucketedStep(list(ps), opt, bucket_bytes=6000)
    assert not rt.collective
    rt.begin()
    _loss(ps, 0).backward()
    rt.finish()
    scale = torch.arange(1280.).view(40, 32) / 1280
    assert torch.allclose(big.data, w0 - 0.5 * scale)
    assert big.grad is not None and torch.allclose(big.grad, scale)     # p.grad views the bucket
    # HF cosine schedule with 3 % warm-up (train.sh)
    import transformers
    lin = torch.nn.Linear(2, 2)
    o = torch.optim.SGD(lin.parameters(), lr=3e-5)
    sch = transformers.get_cosine_schedule_with_warmup(o, num_warmup_steps=math.ceil(0.03 * 200), num_training_steps=200)
    for step in range(200):
        assert abs(sch.get_last_lr()[0] - cosine_with_warmup(step, 200, 0.03, 3e-5)) < 1e-12
        o.step()
        sch.step()
    rt.set_lr(1e-4)
    assert opt.lr == 1e-4


def test_comm_report_contract_single_rank_gloo_group():
    """bench.py's `comm` object = BucketedStep.comm_report(): armed for one step it lists every
    bucket's reduce-scatter and all-gather in launch order with their bytes; durations are None on a backend that
    does not time its work objects (gloo), the device-side tail only exists on a GPU."""
    import socket
    import torch.distributed as dist
    from macaw_llm_amd.bucketed import BucketedStep
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        ps = _make_params()
        rt = BucketedStep(list(ps), ShardSGD(), bucket_bytes=6000, force_collectives=True)
        assert rt.collective and rt.comm_report() is None          # not armed
        for armed in (False, True):                                # (the first step freezes the bucket order)
            rt.profile_comm(armed)
            rt.begin()
            _loss(ps, 0).backward()
            rt.finish()
        rep = rt.comm_report()
        nb = len(rt.buckets)
        assert rep["world"] == 1 and rep["collective"] == "reduce_scatter+all_gather" and rep["buckets"] == nb
        assert len(rep["rs_ms"]) == nb and len(rep["ag_ms"]) == nb and sorted(rep["bucket_order"]) == list(range(nb))
        total = sum(b.g.numel() * b.g.element_size() for b in rt.buckets)
        assert rep["rs_bytes"] == total and rep["ag_bytes"] == total
        assert rep["tail_after_backward_ms"] is None and rep["rs_total_ms"] is None      # CPU tensors, gloo
        assert "not timed" in rep["timing"]
        rt.profile_comm(False)
        assert rt.comm_report() is None
        rt.remove()
    finally:
        dist.destroy_process_group()


def test_committed_collective_path_line_carries_comm():
    """profiles/r04_bench_cfg3_1rank_rccl.json: the bench line of the collective path through a 1-rank RCCL group
    on the GPU -- per-bucket device-side durations from the process group, the un-overlapped tail"""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = json.loads(open(os.path.join(root, "profiles", "r04_bench_cfg3_1rank_rccl.json")).read().strip().splitlines()[-1])
    c = d["comm"]
    assert c["collective"] == "reduce_scatter+all_gather" and c["buckets"] == len(c["rs_ms"]) == len(c["ag_ms"]) > 10
    assert all(x is not None and x > 0 for x in c["rs_ms"]) and c["rs_total_ms"] > 0
    assert c["rs_bytes"] == c["ag_bytes"] > 13e9                   # 6.7e9 trainable bf16 parameters + padding
    assert 0 < c["tail_after_backward_ms"] < d["ms_per_step"]


# ---------------------------------------------------------------- dynamic fp16 loss scale ---
def test_dynamic_loss_scaler_follows_deepspeed_update_rule():
    """deepspeed/runtime/fp16/loss_scaler.py DynamicLossScaler.update_scale, restated: with hysteresis 2 the
    FIRST overflow only uses up the hysteresis, the second halves the scale; `window` clean steps after the last
    overflow double it and refill the hysteresis; the floor is min_scale.  Defaults = the reference's
    configs/deepspeed_config.json:14-21."""
    from macaw_llm_amd.bucketed import DynamicLossScaler
    s = DynamicLossScaler()
    assert (s.scale, s.window, s.hysteresis, s.min_scale) == (2.0 ** 16, 1000, 2, 1.0)
    s = DynamicLossScaler(init_scale=16.0, window=3, hysteresis=2, min_scale=2.0)

    def ref_update(st, overflow):          # line-by-line restatement on a plain dict
        if overflow:
            if st["delayed_shift"] == 1 or st["cur_hysteresis"] == 1:
                st["cur_scale"] = max(st["cur_scale"] / 2.0, st["min_scale"])
            else:
                st["cur_hysteresis"] -= 1
            st["last_overflow_iter"] = st["cur_iter"]
        else:
            if (st["cur_iter"] - st["last_overflow_iter"]) % st["scale_window"] == 0:
                st["cur_hysteresis"] = st["delayed_shift"]
                st["cur_scale"] *= 2.0
        st["cur_iter"] += 1

    st = dict(cur_scale=16.0, cur_iter=0, last_overflow_iter=-1, scale_window=3, min_scale=2.0, delayed_shift=2,
              cur_hysteresis=2)
    seq = [1, 1, 0, 0, 0, 0, 1, 0, 0, 0, 1, 1, 1, 1, 1, 1, 0, 0, 0]
    for ov in seq:
        s.update(bool(ov))
        ref_update(st, bool(ov))
        assert s.scale == st["cur_scale"] and s.cur_hysteresis == st["cur_hysteresis"], (ov, s.scale, st)
    assert s.skipped == sum(seq) and s.scale >= 2.0


def _scaler_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from macaw_llm_amd.bucketed import BucketedStep, DynamicLossScaler
        ps = _make_params()
        big, odd, small, fa, fb, unused, lonely = ps
        ref = dict(big=big.detach().clone(), odd=odd.detach().clone(), small=small.detach().clone(),
                   fused=torch.cat([fa.detach(), fb.detach()]).clone())
        opt = ShardSGD()
        sc = DynamicLossScaler(init_scale=8.0, window=2, hysteresis=2, min_scale=1.0)
        rt = BucketedStep([big, odd, small, fb, fa, unused, lonely], opt, bucket_bytes=6000, loss_scaler=sc)
        eg = _expected_grads(world, 1)
        ok = True
        # step:      0      1         2         3      4      5
        # overflow:  no     rank 1    rank 0    no     no     no      (ONE rank overflowing skips the step on ALL)
        plan = [None, 1, 0, None, None, None]
        want_scale = [8.0, 8.0, 8.0, 4.0, 4.0, 8.0, 8.0]     # the scale each step runs with (last: after step 5)
        taken = 0
        for it, bad in enumerate(plan):
            ok = ok and rt.loss_scale == want_scale[it]
            rt.begin()
            loss = rt.scale_loss(_loss(ps, rank))
            if bad is not None and bad % world == rank:
                loss = loss * float("inf")
            loss.backward()
            rt.finish()
            skipped = bad is not None
            ok = ok and rt.last_step_skipped == skipped
            if not skipped:
                taken += 1
                for k in ("big", "odd", "small", "fused"):
                    ref[k] = ref[k] - 0.5 * eg[k]              # unscaled inside the update: plain SGD on the mean
            got = dict(big=big.data, odd=odd.data, small=small.data, fused=torch.cat([fa.data, fb.data]))
            for k in got:
                if not torch.allclose(got[k], ref[k], atol=1e-5, rtol=1e-5):
                    ok = False
                    print("DEBUG scaler", world, rank, it, k, (got[k] - ref[k]).abs().max().item(), flush=True)
        ok = ok and rt.loss_scale == want_scale[-1] and opt.step_count == taken == 4 and sc.skipped == 2
        flat = torch.cat([b.w for b in rt.buckets])
        others = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(others, flat)
        ok = ok and all(torch.equal(o, others[0]) for o in others) and bool(torch.isfinite(flat).all())
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2])
def test_dynamic_loss_scale_skips_overflowing_steps_on_every_rank(world):
    """a non-finite gradient on ONE rank: no rank updates, Adam's step counter does not advance, the scale
    backs off after the hysteresis and grows back after `window` clean steps; clean steps are exactly the
    unscaled update (the scale is divided out inside the optimizer's grad_scale)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_scaler_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(res) == [(r, True) for r in range(world)]


def test_stage_digests_are_recorded_per_step_and_bucket_and_are_reproducible():
    """BucketedStep.trace_digests (the diagnosis of a world-2 bit mismatch, tests/test_train_gpu.py): one entry per
    step and bucket with the reduced-gradient and updated-bucket digests; two identical runs produce identical
    traces, a changed gradient changes exactly the stages downstream of it."""
    from macaw_llm_amd.bucketed import BucketedStep

    def run(scale):
        ps = _make_params()
        rt = BucketedStep(ps, ShardSGD(), bucket_bytes=4096, overlap=False, direct_grads=False)
        rt.trace_digests()
        for _ in range(2):
            rt.begin()
            (_loss(ps, 0) * scale).backward()
            rt.finish()
        names = rt.bucket_of([(f"p{i}", p) for i, p in enumerate(ps)])
        rt.remove()
        return rt.digests, names

    a, names = run(1.0)
    b, _ = run(1.0)
    c, _ = run(2.0)
    assert len(a) == 2 and len(a[0]) >= 3 and all(e["reduced"] and e["updated"] for e in a[0])
    assert a == b
    assert sorted(names) == [f"p{i}" for i in range(7)] and names["p3"] == names["p4"]      # the fused pair shares a bucket
    changed = [(e["bucket"], e["reduced"] != f["reduced"], e["updated"] != f["updated"]) for e, f in zip(a[0], c[0])]
    assert any(r and u for _, r, u in changed)                  # buckets with gradients: both stages moved
    assert all(r == u for _, r, u in changed)                   # a bucket without gradients moved in neither
