"""GPU: the training-step runtime (bucketed.BucketedStep + optim.FusedAdamW) through RCCL, against
the per-tensor reference steps of rounds 1-2 (tests/legacy_steps.py).
The box has one GPU, so an RCCL process group has ONE rank: the collectives are trivial but the
whole call path of the multi-GPU step (async reduce-scatter on RCCL's stream -> shard AdamW on
the side stream -> in-place all-gather, stream joins) executes for real and must reproduce the
collective-free step bit for bit; world size 2 runs as two processes sharing the GPU over gloo."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu

from golden_util import load_case  # noqa: E402
from oracle import configs  # noqa: E402
from test_model_gpu import build_model, to_dev  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p



def _probe_worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = torch.device("cuda:0")
        t = torch.full((64,), float(rank + 1), device=dev)
        dist.all_reduce(t)
        out = torch.empty(32, device=dev)
        dist.reduce_scatter_tensor(out, torch.ones(64, device=dev))
        full = torch.empty(64, device=dev)
        dist.all_gather_into_tensor(full, out)
        torch.cuda.synchronize()
        ok = bool((t == 3).all()) and bool((full == 2).all())
        q.put((rank, "ok" if ok else "wrong values"))
    except Exception as e:
        q.put((rank, "error: " + repr(e)))
    finally:
        try:
            dist.destroy_process_group()
        except Exception:
            pass


_GLOO_CUDA = {}


def _require_gloo_on_cuda():
    """capability probe, run BEFORE a world-2 test's workers: two processes on this GPU do an all-reduce, a
    reduce-scatter and an all-gather of CUDA tensors through gloo.  Only its failure skips a test; an error of
    the test's own workers is a failure whatever its text says (the hatch of rounds 1-3 matched on 'gloo' /
    'unsupported' in the message and could have hidden a real one)."""
    if "ok" not in _GLOO_CUDA:
        import torch.multiprocessing as mp
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_probe_worker, args=(r, 2, port, q)) for r in range(2)]
        for p in procs:
            p.start()
        try:
            res = [q.get(timeout=120) for _ in procs]
        except Exception as e:       # a hung probe is a capability failure too
            res = [(0, "error: probe timed out " + repr(e))]
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
        _GLOO_CUDA["ok"] = all(r[1] == "ok" for r in res)
        _GLOO_CUDA["why"] = "; ".join(r[1] for r in res if r[1] != "ok")
    if not _GLOO_CUDA["ok"]:
        pytest.skip("gloo cannot run collectives on CUDA tensors on this box: " + _GLOO_CUDA["why"][:200])


def _run(dev, fx, cfg, steps, **kw):
    from macaw_llm_amd.optim import FusedAdamW
    from legacy_steps import OverlappedStep
    model = build_model(cfg, fx["state"], torch.bfloat16, dev, fuse=True).eval()   # eval: no dropout RNG
    params = [p for p in model.parameters() if p.requires_grad]
    opt = FusedAdamW(params, lr=1e-3, weight_decay=0.01)
    rt = OverlappedStep(params, opt, small_threshold=4096, **kw)
    inp = to_dev(fx["inputs"], dev)
    losses = []
    for _ in range(steps):
        rt.begin()
        loss = model(inputs=inp).loss
        loss.backward()
        rt.finish()
        losses.append(loss.item())
    torch.cuda.synchronize()
    rt.remove()
    # parameters that are not in the golden state dict (unused towers) keep an unseeded random init
    return losses, {n: p.detach().clone() for n, p in model.named_parameters()
                    if p.requires_grad and n in fx["state"]}, rt


def test_sharded_and_allreduce_steps_match_plain_step_through_rccl(dev):
    fx = load_case("micro_all")
    cfg = configs.get(fx["config_name"])
    ref_losses, ref_params, _ = _run(dev, fx, cfg, 3)
    assert ref_losses[2] < ref_losses[0]                       # it trains
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            device_id=dev)
    try:
        for shard in (True, False):
            losses, params, rt = _run(dev, fx, cfg, 3, force_collectives=True, shard_optimizer=shard)
            assert rt.collective and rt.shard == shard
            assert losses == ref_losses, (shard, losses, ref_losses)
            assert params.keys() == ref_params.keys()
            for n in params:
                assert torch.equal(params[n], ref_params[n]), (shard, n)
    finally:
        dist.destroy_process_group()


def _worker_two_ranks_one_gpu(rank, world, port, q, shard):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from golden_util import load_case
        from oracle import configs
        from test_model_gpu import build_model, to_dev
        from macaw_llm_amd.optim import FusedAdamW
        from legacy_steps import OverlappedStep
        dev = torch.device("cuda:0")
        from conftest import poison_allocator
        poison_allocator(256, 256)
        fx = load_case("micro_all")
        cfg = configs.get(fx["config_name"])
        model = build_model(cfg, fx["state"], torch.bfloat16, dev, fuse=True).eval()
        params = [p for p in model.parameters() if p.requires_grad]
        opt = FusedAdamW(params, lr=1e-3, weight_decay=0.01)
        poison_allocator(256, 256)
        rt = OverlappedStep(params, opt, small_threshold=4096, shard_optimizer=shard)
        inp = to_dev(fx["inputs"], dev)
        # every rank sees its own half of the batch (2 samples -> 1 each)
        mine = {k: (v[rank:rank + 1] if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == 2 else v)
                for k, v in inp.items()}
        for _ in range(2):
            rt.begin()
            model(inputs=mine).loss.backward()
            rt.finish()
        torch.cuda.synchronize()
        import hashlib
        out = {n: hashlib.sha1(p.detach().float().cpu().numpy().tobytes()).hexdigest()
               for n, p in model.named_parameters() if p.requires_grad and n in fx["state"]}
        q.put((rank, rt.shard, out))       # digests, not tensors: the child may exit before the parent reads
    except Exception as e:   # surfaced by the parent
        q.put((rank, "error", repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("shard", [True, False])
def test_two_ranks_share_one_gpu_through_gloo(dev, shard):
    """World size 2 with REAL data exchange and the real fused AdamW: two processes on this one GPU,
    collectives through gloo (CUDA tensors staged by the backend).  The ZeRO-1 step (each rank
    owns half of every large tensor's optimizer state) must leave both replicas identical and equal
    to the all-reduce + replicated-AdamW step."""
    import torch.multiprocessing as mp
    _require_gloo_on_cuda()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_two_ranks_one_gpu, args=(r, 2, port, q, shard)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    if any(r[1] == "error" for r in res):
        raise AssertionError("; ".join(str(r[2]) for r in res if r[1] == "error"))
    assert res[0][1] == shard and res[1][1] == shard
    a, b = res[0][2], res[1][2]
    assert a.keys() == b.keys()
    assert len(a) > 20
    for n in a:
        assert a[n] == b[n], n                       # replicas identical after 2 steps (bit for bit)
    globals().setdefault("_TWO_RANK_RESULTS", {})[shard] = a
    other = globals()["_TWO_RANK_RESULTS"].get(not shard)
    if other is not None:                            # sharded == replicated update, bit for bit
        for n in a:
            assert a[n] == other[n], n


# ---------------------------------------------------------------- BucketedStep (flat buckets) ---
def _run_bucketed(dev, fx, cfg, steps, micros=1, half=None, **kw):
    from macaw_llm_amd.optim import FusedAdamW
    from macaw_llm_amd.bucketed import BucketedStep
    model = build_model(cfg, fx["state"], torch.bfloat16, dev, fuse=True).eval()
    params = [p for p in model.parameters() if p.requires_grad]
    opt = FusedAdamW(params, lr=1e-3, weight_decay=0.01)
    rt = BucketedStep(params, opt, bucket_bytes=64 << 10, accumulate_steps=micros, **kw)
    inp = to_dev(fx["inputs"], dev)
    losses = []
    for _ in range(steps):
        for m in range(micros):
            rt.begin()
            loss = model(inputs=inp).loss
            loss.backward()
            rt.finish()
        losses.append(loss.item())
    torch.cuda.synchronize()
    rt.remove()
    l0 = model.llm.model.layers[0]
    assert all(v is not None for v in l0.fused_weights())       # re-homing kept q|k|v / gate|up fused
    return losses, {n: p.detach().clone() for n, p in model.named_parameters()
                    if p.requires_grad and n in fx["state"]}, rt


def test_bucketed_step_is_bit_identical_to_the_per_tensor_step(dev):
    """same kernel (mk_adamw arithmetic) on the same gradients: flat buckets with the grad-weight
    GEMMs writing straight into them == per-tensor gradients + multi-tensor AdamW, bit for bit;
    through a 1-rank RCCL group the reduce-scatter / all-gather call path runs for real."""
    fx = load_case("micro_all")
    cfg = configs.get(fx["config_name"])
    ref_losses, ref_params, _ = _run(dev, fx, cfg, 3)
    from macaw_llm_amd import bucketed as Bk
    copies = {"n": 0}
    real_copy = Bk._copy_

    def counting_copy(dst, src):
        copies["n"] += 1
        return real_copy(dst, src)
    Bk._copy_ = counting_copy
    try:
        losses, params, rt = _run_bucketed(dev, fx, cfg, 3)
    finally:
        Bk._copy_ = real_copy
    assert len(rt.buckets) >= 3 and not rt.collective
    assert losses == ref_losses
    # the gradients really are stored straight into the buckets: autograd must TAKE the bucket views as
    # p.grad (round 2 handed out a shared tensor object, AccumulateGrad cloned it and the hook copied
    # it back: ~300 hidden copies per step at 7B).  What may still be copied in: the handful of
    # gradients that are produced as fresh tensors (bias_k / bias_v of the three alignment attentions)
    n_params = sum(len(b.items) for b in rt.buckets)
    per_step = (copies["n"] - n_params) / 3          # (the re-homing at construction copies every parameter once)
    assert per_step <= 10, (copies["n"], n_params, per_step)
    for n in params:
        assert torch.equal(params[n], ref_params[n]), n
    # copies instead of direct GEMM stores: same bits
    losses2, params2, _ = _run_bucketed(dev, fx, cfg, 3, direct_grads=False)
    assert losses2 == ref_losses and all(torch.equal(params2[n], ref_params[n]) for n in params2)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            device_id=dev)
    try:
        losses3, params3, rt3 = _run_bucketed(dev, fx, cfg, 3, force_collectives=True)
        assert rt3.collective
        assert losses3 == ref_losses
        for n in params3:
            assert torch.equal(params3[n], ref_params[n]), n
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("average", [True, False])
def test_bucketed_accumulation_and_clipping(dev, average):
    """two identical micro-batches: average_accumulated (the default; HF Trainer / DeepSpeed divide the
    loss by gradient_accumulation_steps, train.sh:29) gives the gradient of ONE batch, the sum form 2 x
    it; with max_grad_norm the update uses grad_scale = max_norm / ||g||: compare against the
    per-tensor step fed the same gradient and scale."""
    from macaw_llm_amd.optim import FusedAdamW
    fx = load_case("micro_all")
    cfg = configs.get(fx["config_name"])
    mult = 1.0 if average else 2.0
    model = build_model(cfg, fx["state"], torch.bfloat16, dev, fuse=True).eval()
    params = [p for p in model.parameters() if p.requires_grad]
    inp = to_dev(fx["inputs"], dev)
    model(inputs=inp).loss.backward()
    for p in params:
        if p.grad is not None:
            p.grad = (p.grad.float() * mult).to(p.grad.dtype)
    gn = torch.sqrt(sum((p.grad.float() ** 2).sum() for p in params if p.grad is not None)).item()
    max_norm = 0.5 * gn
    opt = FusedAdamW(params, lr=1e-3, weight_decay=0.01)
    opt.step(grad_scale=min(1.0, max_norm / (gn + 1e-6)))
    want = {n: p.detach().clone() for n, p in model.named_parameters() if p.requires_grad and n in fx["state"]}
    _, got, rt = _run_bucketed(dev, fx, cfg, 1, micros=2, max_grad_norm=max_norm, average_accumulated=average)
    assert abs(float(rt.grad_norm) - gn) <= 2e-3 * gn
    for n in got:
        # (bf16 accumulation g + g is exact; the clip factor differs in the last fp32 digits)
        assert (got[n].float() - want[n].float()).abs().max().item() <= 2e-3 * max(1e-3, want[n].float().abs().max().item()), n


def test_bucketed_step_built_on_an_unfused_model_still_trains_every_projection(dev):
    """round-2 advisor finding (high): BucketedStep built BEFORE the first forward of a model whose
    q|k|v / gate|up were not fused yet, followed by the lazy fusion of modeling.AUTO_FUSE, re-homed the
    parameters OUT of the buckets -- the optimizer kept updating the bucket, the model computed with
    the stale copies, silently.  Now (a) model= fuses first, (b) without it the pinned parameters are
    never moved (per-projection GEMMs), (c) begin() verifies the homes.  Both ways must train q|k|v
    and agree with the fused-first run to bf16 rounding of the different GEMM shapes."""
    from macaw_llm_amd.optim import FusedAdamW
    from macaw_llm_amd.bucketed import BucketedStep
    from macaw_llm_amd import modeling as M
    fx = load_case("micro_all")
    cfg = configs.get(fx["config_name"])
    inp = to_dev(fx["inputs"], dev)

    def run(mode):
        model = build_model(cfg, fx["state"], torch.bfloat16, dev, fuse=(mode == "fused")).eval()
        l0 = model.llm.model.layers[0]
        if mode != "fused":
            assert M._rows_view([l0.self_attn.q_proj.weight, l0.self_attn.k_proj.weight,
                                 l0.self_attn.v_proj.weight]) is None            # really unfused at build
        params = [p for p in model.parameters() if p.requires_grad]
        opt = FusedAdamW(params, lr=1e-3, weight_decay=0.0)
        rt = BucketedStep(params, opt, bucket_bytes=64 << 10, model=model if mode == "model_arg" else None)
        w0 = {n: p.detach().clone() for n, p in model.named_parameters() if p.requires_grad and n in fx["state"]}
        losses = []
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for _ in range(3):
                rt.begin()
                loss = model(inputs=inp).loss
                loss.backward()
                rt.finish()
                losses.append(loss.item())
        torch.cuda.synchronize()
        rt._check_homes()                     # parameters still view their bucket slots after three forwards
        fused_now = all(v is not None for v in l0.fused_weights())
        rt.remove()
        w1 = {n: p.detach().clone() for n, p in model.named_parameters() if n in w0}
        return losses, w0, w1, fused_now

    ref = run("fused")
    for mode in ("model_arg", "pinned_unfused"):
        losses, w0, w1, fused_now = run(mode)
        assert fused_now == (mode == "model_arg")
        assert losses[2] < losses[0]
        for n in w1:
            if any(k in n for k in ("q_proj", "k_proj", "v_proj", "gate_proj", "up_proj")) and "llm." in n:
                assert not torch.equal(w0[n], w1[n]), f"{mode}: {n} was not trained"
            d = (w1[n].float() - ref[2][n].float()).abs().max().item()
            assert d <= 4e-3 + 2e-2 * ref[2][n].float().abs().max().item(), (mode, n, d)
        assert abs(losses[2] - ref[0][2]) <= 2e-2 * abs(ref[0][2])


def _worker_bucketed_two_ranks(rank, world, port, q):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from golden_util import load_case
        from oracle import configs
        from test_model_gpu import build_model, to_dev
        from macaw_llm_amd.optim import FusedAdamW
        from macaw_llm_amd.bucketed import BucketedStep
        dev = torch.device("cuda:0")
        from conftest import poison_allocator
        poison_allocator(256, 256)
        fx = load_case("micro_all")
        cfg = configs.get(fx["config_name"])
        model = build_model(cfg, fx["state"], torch.bfloat16, dev, fuse=True).eval()
        params = [p for p in model.parameters() if p.requires_grad]
        opt = FusedAdamW(params, lr=1e-3, weight_decay=0.01)
        poison_allocator(256, 256)
        rt = BucketedStep(params, opt, bucket_bytes=64 << 10)
        rt.trace_digests()                 # per step and bucket: local gradient / reduced slice / updated bucket
        inp = to_dev(fx["inputs"], dev)
        mine = {k: (v[rank:rank + 1] if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == 2 else v)
                for k, v in inp.items()}
        for _ in range(2):
            rt.begin()
            model(inputs=mine).loss.backward()
            rt.finish()
        torch.cuda.synchronize()
        import hashlib
        out = {n: hashlib.sha1(p.detach().float().cpu().numpy().tobytes()).hexdigest()
               for n, p in model.named_parameters() if p.requires_grad and n in fx["state"]}
        q.put((rank, rt.collective, out, rt.digests, rt.bucket_of(model.named_parameters())))
    except Exception as e:
        q.put((rank, "error", repr(e)))
    finally:
        dist.destroy_process_group()


def _run_bucketed_two_ranks():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bucketed_two_ranks, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    if any(r[1] == "error" for r in res):
        raise AssertionError("; ".join(str(r[2]) for r in res if r[1] == "error"))
    return res


def _diagnose_bucketed_mismatch(res, bad):
    """A replica / runtime mismatch is a NON-DETERMINISM somewhere in the step (round 2 saw one weight differ once,
    rounds 3-5 never again: DESIGN.md section 6).  Name it: run the same two ranks once more and compare the stage
    digests of the two runs (BucketedStep.trace_digests) -- the first (step, bucket, stage) that differs between two
    runs that must be bit-identical is the culprit: `local_grad` = a backward kernel writing into that bucket,
    `reduced` = the collective, `updated` = AdamW / the all-gather."""
    lines = [f"mismatching parameters: {bad[:8]} (buckets {sorted({res[0][4].get(n) for n in bad})})"]
    try:
        again = _run_bucketed_two_ranks()
    except Exception as e:                      # the diagnosis must not hide the finding
        return "\n".join(lines + [f"re-run for the stage digests failed: {e!r}"])
    first = None
    for rank in (0, 1):
        for s1, s2 in zip(res[rank][3], again[rank][3]):
            for e1, e2 in zip(s1, s2):
                for stage in ("local_grad", "reduced", "updated"):
                    if e1[stage] != e2[stage]:
                        names = [n for n, bi in res[rank][4].items() if bi == e1["bucket"]]
                        lines.append(f"rank {rank} step {e1['step']} bucket {e1['bucket']} stage {stage}: "
                                     f"{e1[stage]} vs {e2[stage]} on the re-run; parameters of the bucket: {names}")
                        first = first or (rank, e1["step"], e1["bucket"], stage)
                        break
    if first is None:
        lines.append("the re-run reproduced every stage digest of the failing run: deterministic difference "
                     "between the two runtimes / replicas, not a race")
        for rank in (0, 1):
            lines.append(f"rank {rank} digests: {res[rank][3]}")
    else:
        lines.append(f"FIRST differing stage: rank {first[0]} step {first[1]} bucket {first[2]} {first[3]}")
    return "\n".join(lines)


def test_bucketed_two_ranks_share_one_gpu_through_gloo(dev):
    """world 2 with real data exchange and the real fused AdamW: the bucketed ZeRO-1 step leaves
    both replicas identical, and identical to the per-tensor ZeRO-1 step of the test above (same
    rank-mean gradients, same kernel).  STRICT; a failure re-runs the ranks and prints which (step, bucket,
    stage) is not reproducible (_diagnose_bucketed_mismatch)."""
    _require_gloo_on_cuda()
    res = _run_bucketed_two_ranks()
    a, b = res[0][2], res[1][2]
    assert res[0][1] is True and a.keys() == b.keys() and len(a) > 20
    assert len(res[0][3]) == 2 and all(e["updated"] for e in res[0][3][0])     # two steps traced, every bucket
    # replicas: the bucket AFTER the all-gather is the same bytes on both ranks, step by step
    for s0, s1 in zip(res[0][3], res[1][3]):
        for e0, e1 in zip(s0, s1):
            assert e0["updated"] == e1["updated"], _diagnose_bucketed_mismatch(res, [n for n in a if a[n] != b[n]])
    bad_rep = [n for n in a if a[n] != b[n]]
    assert not bad_rep, _diagnose_bucketed_mismatch(res, bad_rep)
    other = globals().get("_TWO_RANK_RESULTS", {}).get(True)
    if other is not None:
        # same rank-mean gradients, same kernel body: bit-identical to the per-tensor ZeRO-1 step.  (Round 2
        # retried here on a rare one-weight mismatch; the cause was found and fixed in round 3 -- DESIGN.md
        # section 6 "the world-2 mismatch" -- and the comparison is strict again.)
        bad = [n for n in a if a[n] != other[n]]
        assert not bad, _diagnose_bucketed_mismatch(res, bad)


def test_graphed_step_is_bit_identical_to_the_eager_step(dev):
    """train.GraphedStep: the whole step (forward, backward, fused AdamW) replayed from one hipGraph,
    with the optimizer scalars and the dropout seed offset in device memory, must produce the SAME
    losses and parameters, bit for bit, as the step launched kernel by kernel -- dropout on (train
    mode), a changing learning rate, and an eager step interleaved after the capture."""
    from macaw_llm_amd.optim import FusedAdamW
    from macaw_llm_amd.train import GraphedStep
    from macaw_llm_amd.bucketed import BucketedStep
    fx = load_case("micro_all")
    cfg = configs.get(fx["config_name"])
    inp = to_dev(fx["inputs"], dev)

    def run(graphed):
        model = build_model(cfg, fx["state"], torch.bfloat16, dev, fuse=True).train()
        params = [p for p in model.parameters() if p.requires_grad]
        opt = FusedAdamW(params, lr=1e-3, weight_decay=0.01)
        gs = GraphedStep(model, lambda: model(inputs=inp).loss, BucketedStep(params, opt, bucket_bytes=64 << 10))
        losses = []
        for it in range(6):
            opt.lr = 1e-3 * (1.0 - 0.1 * it)
            if graphed and it != 4:
                losses.append(float(gs.step()))
            else:
                losses.append(float(gs.eager_step()))
        torch.cuda.synchronize()
        gs.rt.remove()
        # (parameters outside the fixture's state dict are random per build: compare the trained ones)
        return losses, {n: p.detach().clone() for n, p in model.named_parameters()
                        if p.requires_grad and n in fx["state"]}, gs

    le, pe, _ = run(False)
    lg, pg, gs = run(True)
    assert gs.graph is not None and gs._graph_steps == 4
    assert le == lg, (le, lg)
    assert len(set(le)) > 1                     # the steps really differ (updates + fresh dropout masks)
    assert pe.keys() == pg.keys() and len(pe) > 20
    for n in pe:
        assert torch.equal(pe[n], pg[n]), n


@pytest.mark.parametrize("inject", [None, "all"])
def test_bench_py_runs_its_n_gt_1_branch_with_two_ranks_on_this_gpu(dev, inject):
    """bench.py's N > 1 branch (process-group setup, BucketedStep ZeRO-1 with the frozen bucket order,
    cross-rank agreement on a failed setup step, max-over-ranks timing, describe()) had never executed
    before the driver's 8-GPU run (round-2 verdict).  Here it runs as two torchrun ranks sharing this
    GPU over gloo (MACAW_SHARE_GPU / MACAW_DIST_BACKEND; 2-layer model, marked invalid as a
    benchmark).  inject="all": the ZeRO-1 setup step fails on every rank (the symmetric failure a
    missing collective or an OOM produces) -- the ranks must agree, fall back TOGETHER to all-reduce +
    replicated AdamW and say so in config.parallelism."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MACAW_SHARE_GPU="1", MACAW_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if inject is not None:
        env["MACAW_BENCH_INJECT_FAIL"] = inject
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--layers", "2", "--batch-per-gpu", "2", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]           # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4 and d["value"] > 0
    assert "invalid" in d
    par = d["config"]["parallelism"]
    if inject is None:
        assert "ZeRO-1" in par and "DEGRADED" not in par, par
    else:
        assert "DEGRADED" in par and "all-reduce / replicated AdamW" in par, par
    assert d["roofline"]["launches_per_step"] > 0 and d["config"]["loss"] == d["config"]["loss"]   # not NaN


def test_bare_bench_py_gpus_2_launches_its_own_ranks(dev):
    """VERDICT r5 Missing 1: the driver calls `python bench.py --gpus N ...` WITHOUT a launcher.  With WORLD_SIZE unset
    and --gpus 2 bench.py starts its own two ranks through torch.distributed.run (train.sh:13) and forwards rank 0's
    single JSON line and the exit code.  Two ranks sharing this GPU over gloo; marked invalid as a benchmark."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MACAW_SHARE_GPU="1", MACAW_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--layers", "2",
           "--batch-per-gpu", "2", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4 and d["value"] > 0 and "invalid" in d
    assert "ZeRO-1" in d["config"]["parallelism"]


def test_optimizer_state_dict_resumes_bit_exactly(dev):
    """HF Trainer saves `optimizer.state_dict()` beside the model every save_steps (train.sh:24-26; the
    reference's resume is commented out, run_clm_llms.py:556-561).  FusedAdamW.state_dict() = fp32 master
    weights + both moments of every bucket slot + step counter; 2 steps, checkpoint, 2 more steps must equal
    2 steps, checkpoint -> a NEW model / optimizer / runtime loaded from it -> 2 steps, bit for bit -- both when
    the state is restored lazily (loaded before the first step) and into slots that already exist."""
    import copy
    from macaw_llm_amd.bucketed import BucketedStep
    from macaw_llm_amd.optim import FusedAdamW
    fx = load_case("micro_all")
    cfg = configs.get(fx["config_name"])
    inp = to_dev(fx["inputs"], dev)

    def fresh():
        model = build_model(cfg, fx["state"], torch.bfloat16, dev, fuse=True).eval()
        params = [p for p in model.parameters() if p.requires_grad]
        opt = FusedAdamW(params, lr=1e-3, weight_decay=0.01)
        return model, opt, BucketedStep(params, opt, bucket_bytes=64 << 10)

    def steps(model, rt, n):
        for _ in range(n):
            rt.begin()
            model(inputs=inp).loss.backward()
            rt.finish()
        torch.cuda.synchronize()

    A, opt_a, rt_a = fresh()
    steps(A, rt_a, 2)
    ck_model = {k: v.detach().clone() for k, v in A.state_dict().items()}
    ck_opt = copy.deepcopy(opt_a.state_dict())
    assert ck_opt["step_count"] == 2 and all(k.startswith("shard:") for k in ck_opt["state"])
    assert all(t["master"].dtype == torch.float32 for t in ck_opt["state"].values())
    steps(A, rt_a, 2)
    want = {n: p.detach().clone() for n, p in A.named_parameters()}
    rt_a.remove()
    assert any(not torch.equal(want[n], ck_model[n]) for n in want if n in ck_model)      # steps 3-4 did train
    for warm in (0, 1):
        B, opt_b, rt_b = fresh()
        steps(B, rt_b, warm)                       # warm = 1: the optimizer slots exist before the load
        B.load_state_dict(ck_model)
        opt_b.load_state_dict(ck_opt)
        steps(B, rt_b, 2)
        assert opt_b.step_count == 4
        for n, p in B.named_parameters():
            assert torch.equal(p.detach(), want[n]), (warm, n)
        rt_b.remove()


def test_one_rank_updates_overlap_the_backward_and_change_no_bit(dev):
    """round 5 (opt-in, measured +-0 at 7B and therefore off by default): with one rank and no collectives the AdamW
    launch of a bucket can go out on a high-priority side stream as soon as the bucket's gradients are complete (from
    the second step on: the first discovers the order) instead of as one launch behind the backward.  Same kernel,
    same keys: the weights and the optimizer state are bit-identical to the serial form."""
    fx = load_case("micro_all")
    cfg = configs.get(fx["config_name"])
    l_ser, p_ser, rt_ser = _run_bucketed(dev, fx, cfg, 4)
    l_ovl, p_ovl, rt_ovl = _run_bucketed(dev, fx, cfg, 4, local_overlap=True)
    assert rt_ovl.local_overlap and rt_ovl.side is not None and not rt_ser.local_overlap and rt_ser.side is None
    assert any(b.updated for b in rt_ovl.buckets), "no bucket was updated behind the backward"
    assert not any(getattr(b, "updated", False) for b in rt_ser.buckets)
    assert l_ovl == l_ser
    for n in p_ser:
        assert torch.equal(p_ovl[n], p_ser[n]), n
    assert sorted(map(str, rt_ovl.opt.state)) == sorted(map(str, rt_ser.opt.state))
    assert "side stream" in rt_ovl.describe() and "one fused AdamW launch" in rt_ser.describe()


def test_grad_weight_gemms_on_the_side_stream_change_no_bit(dev):
    """engine.DW_SIDE: the grad-weight GEMM of every decoder-layer projection goes out on a second stream beside the
    grad-input GEMM (submitted first) and the layer's backward joins before it returns ("auto": where the [M, D] grad-input
    GEMMs fill less than one round of CUs, BASELINE cfg 2).  Same kernels on the same operands, per-stream GEMM scratch,
    gradients stored straight into the buckets by the side stream: losses, weights and optimizer state bit-identical to
    the one-stream step, with and without the per-bucket updates behind the backward."""
    from macaw_llm_amd import engine
    fx = load_case("micro_all")
    cfg = configs.get(fx["config_name"])
    old = engine.DW_SIDE["on"]
    try:
        engine.DW_SIDE["on"] = False
        l_ref, p_ref, _ = _run_bucketed(dev, fx, cfg, 4)
        engine.DW_SIDE["on"] = True
        engine.DW_SIDE["launches"] = 0
        l_side, p_side, _ = _run_bucketed(dev, fx, cfg, 4)
        assert engine.DW_SIDE["launches"] > 0, "no grad-weight GEMM went through the side stream"
        l_both, p_both, _ = _run_bucketed(dev, fx, cfg, 4, local_overlap=True)
    finally:
        engine.DW_SIDE["on"] = old
    assert l_side == l_ref and l_both == l_ref
    for n in p_ref:
        assert torch.equal(p_side[n], p_ref[n]), n
        assert torch.equal(p_both[n], p_ref[n]), n
    # "auto": on exactly where the grad-input GEMMs of the layer are less than one round of 256 x 256 tiles
    engine.DW_SIDE["on"] = "auto"
    try:
        cus = torch.cuda.get_device_properties(dev).multi_processor_count
        assert engine._DwSide(dev, 2176, 4096).side is not None          # cfg 2: 9 x 16 = 144 tiles
        assert engine._DwSide(dev, 4608, 4096).side is None or cus > 288   # cfg 3: 288 tiles
        assert engine._DwSide(dev, 144, 4096).side is None                # one sample: skinny kernels, left alone
        assert engine._DwSide(dev, 8192, 4096).side is None or cus != 256  # cfg 4: 512 tiles = two whole rounds
        assert engine._DwSide(dev, 4608, 5120).side is not None or cus != 256   # cfg 5: 360 tiles = 256 + 104
    finally:
        engine.DW_SIDE["on"] = old


def test_bucketed_two_ranks_with_the_grad_weight_side_stream(dev, monkeypatch):
    """the world-2 ZeRO-1 step (reduce-scatter / shard AdamW / all-gather launched from the gradient hooks) with every
    decoder layer's grad-weight GEMMs on the side stream (MACAW_DW_STREAM=1 in the ranks): the layer's backward joins
    before the hooks fire, so the collectives see finished gradients -- replicas identical and identical to the
    one-stream run, bit for bit."""
    _require_gloo_on_cuda()
    plain = _run_bucketed_two_ranks()
    monkeypatch.setenv("MACAW_DW_STREAM", "1")
    side = _run_bucketed_two_ranks()
    assert side[0][1] is True and len(side[0][2]) > 20
    assert side[0][2] == side[1][2], [n for n in side[0][2] if side[0][2][n] != side[1][2][n]]
    assert side[0][2] == plain[0][2], [n for n in side[0][2] if side[0][2][n] != plain[0][2][n]]
    # stage digests of the gradients, step by step ("updated" is not comparable across two spawns: the model's dead
    # parameters -- temporal_*, logit_scale: never used, zero gradient -- are initialised randomly per process)
    for rank in (0, 1):
        for s_p, s_s in zip(plain[rank][3], side[rank][3]):
            for e_p, e_s in zip(s_p, s_s):
                assert e_p["local_grad"] == e_s["local_grad"] and e_p["reduced"] == e_s["reduced"], (rank, e_p, e_s)
