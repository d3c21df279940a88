"""CPU tests of the round-5 advisor findings: the optimizer layout in all-reduce mode is rank-independent
(HF's stock rank-0 optimizer.pt resumes on every rank; 2 gloo ranks), the high-priority collective group is cached per
LIVE default process group, and a training_step that raises drops its accumulation window."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _layout_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from macaw_llm_amd.bucketed import BucketedStep
        from macaw_llm_amd.optim import FusedAdamW
        res = {}
        for zero1 in (False, True):
            torch.manual_seed(0)
            ps = [torch.nn.Parameter(torch.randn(16, 16)) for _ in range(3)]
            opt = FusedAdamW(ps, lr=1e-3)
            rt = BucketedStep(ps, opt, bucket_bytes=4096, zero1=zero1)
            opt.attach_runtime(rt)
            lay = rt.layout()
            # what HF's stock save/load does when the mixin does not shard (macaw_zero1=False): rank 0 writes
            # optimizer.state_dict(), EVERY rank loads that file into a fresh optimizer + runtime
            box = [opt.state_dict() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            ps2 = [torch.nn.Parameter(p.detach().clone()) for p in ps]
            opt2 = FusedAdamW(ps2, lr=1e-3)
            opt2.load_state_dict(box[0])                         # before the runtime exists: stashed (HF's order)
            rt2 = BucketedStep(ps2, opt2, bucket_bytes=4096, zero1=zero1)
            try:
                opt2.attach_runtime(rt2)
                err = None
            except ValueError as e:
                err = str(e)
            res[zero1] = (lay["rank"], lay["world"], lay["zero1"], err)
            rt.remove()
            rt2.remove()
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_rank0_optimizer_file_resumes_on_every_rank_in_all_reduce_mode():
    """with macaw_zero1=False and N > 1 the mixin uses HF's stock save (rank 0 only) and every
    rank reloads that file; layout() said rank = r in all-reduce mode although the state is replicated and the shard
    keys rank-independent -> ValueError on ranks >= 1.  Under ZeRO-1 the rank still matters and rank 1 must refuse
    rank 0's shards."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_layout_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in range(world):
        lr, lw, lz, err = res[rank][False]
        assert (lr, lw, lz) == (0, 2, False) and err is None, res          # all-reduce: rank-independent, resumes
        lr, lw, lz, err = res[rank][True]
        assert (lr, lw, lz) == (rank, 2, True), res
        assert (err is None) == (rank == 0), res                          # ZeRO-1: rank 1 refuses rank 0's shards
        if rank == 1:
            assert "layout" in err


def test_overlap_group_cache_is_keyed_by_the_live_default_group(monkeypatch):
    """_OVERLAP_GROUP was keyed by world size only; after destroy_process_group() / re-init in the
    same process the stale group was handed back.  And only a missing OPTION (TypeError / AttributeError) may fall
    back to the default group: a failing new_group() must surface."""
    from macaw_llm_amd import bucketed as bk
    made = []
    worlds = [object(), object()]
    cur = {"w": worlds[0]}
    monkeypatch.setattr(dist.distributed_c10d, "_get_default_group", lambda: cur["w"])

    class _PG:
        @staticmethod
        def Options(is_high_priority_stream=False):
            return ("opts", is_high_priority_stream)

    monkeypatch.setattr(dist, "ProcessGroupNCCL", _PG, raising=False)
    monkeypatch.setattr(dist, "new_group", lambda **kw: made.append(kw) or f"group{len(made)}")
    monkeypatch.delenv("MACAW_COMM_NORMAL_PRIORITY", raising=False)
    bk._OVERLAP_GROUP.clear()
    g1 = bk.overlap_group()
    assert g1[0] == "group1" and made[0]["pg_options"] == ("opts", True) and bk.overlap_group() is g1 and len(made) == 1
    cur["w"] = worlds[1]                                   # the process re-initialised torch.distributed
    g2 = bk.overlap_group()
    assert g2[0] == "group2" and len(made) == 2
    # an older torch without the option: default group, said so
    cur["w"] = object()

    class _Old:
        @staticmethod
        def Options(**kw):
            raise TypeError("unexpected keyword is_high_priority_stream")

    monkeypatch.setattr(dist, "ProcessGroupNCCL", _Old, raising=False)
    g3 = bk.overlap_group()
    assert g3[0] is None and "no high-priority option" in g3[1]
    # new_group itself failing is NOT swallowed
    cur["w"] = object()
    monkeypatch.setattr(dist, "ProcessGroupNCCL", _PG, raising=False)

    def boom(**kw):
        raise RuntimeError("rendezvous failed")

    monkeypatch.setattr(dist, "new_group", boom)
    with pytest.raises(RuntimeError, match="rendezvous"):
        bk.overlap_group()
    bk._OVERLAP_GROUP.clear()


def test_training_step_that_raises_drops_its_window():
    """hf.MacawTrainerMixin.training_step had no try/except: a forward that raised at micro-step
    k > 0 left _micro = k and the direct-gradient destinations installed; the next begin() accumulated onto stale
    bucket contents."""
    from macaw_llm_amd.hf import MacawTrainerMixin
    log = []

    class Rt:
        _micro, accumulate_steps = 1, 2

        def begin(self):
            log.append("begin")

        def abort(self):
            log.append("abort")
            self._micro = 0

        def finish(self):
            log.append("finish")

        def scale_loss(self, loss):
            return loss

    import contextlib

    class T(MacawTrainerMixin):
        def macaw_runtime(self):
            return self._rt

        def _prepare_inputs(self, x):
            return x

        def compute_loss_context_manager(self):
            return contextlib.nullcontext()

        def compute_loss(self, model, inputs):
            raise RuntimeError("bad batch")

    t = T()
    t._rt = Rt()
    with pytest.raises(RuntimeError, match="bad batch"):
        t.training_step(torch.nn.Linear(1, 1), {})
    assert log == ["begin", "abort"] and t._rt._micro == 0
